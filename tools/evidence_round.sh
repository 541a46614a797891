#!/bin/bash
# The round's evidence in one place (run through gpurun from the repo root, one stage per call: the stages are minutes each):
#   bash tools/evidence_round.sh bench|zoo|models|tools     -> gpurun_out/ev_*; copy what is to be judged into profiles/rN_*
set -u
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
case "${1:-bench}" in
bench)   # the bench.py lines (default workload, C4, sum, the DDP step graphed / eager / without the forced exchange) and the
         # rocprofv3 passes of the default command (kernel stats, HBM traffic, SQ counters)
  python bench.py > $O/ev_bench_line.json 2> $O/ev_bench_line.err
  python bench.py --feat 4 --no-cpu-baseline > $O/ev_bench_line_C4.json 2>> $O/ev_bench_line.err
  python bench.py --reduce sum --no-cpu-baseline > $O/ev_bench_line_sum.json 2>> $O/ev_bench_line.err
  python bench.py --mode ddp-step --steps 10 --warmup 3 > $O/ev_bench_line_ddp_step.json 2> $O/ev_ddp.err
  python bench.py --mode ddp-step --steps 10 --warmup 3 --no-graph > $O/ev_bench_line_ddp_step_eager.json 2>> $O/ev_ddp.err
  CLOUDCT_SYNCBN_FORCE=0 python bench.py --mode ddp-step --steps 10 --warmup 3 > $O/ev_bench_line_ddp_step_noexchange.json 2>> $O/ev_ddp.err
  bash tools/profile_round.sh > $O/ev_profile_round.log 2>&1
  bash tools/pmc_sq.sh > $O/ev_pmc_sq.log 2>&1
  python3 tools/pmc_sq.py > $O/ev_bench_sq_counters.txt 2>&1
  bash tools/dev/bench_seeds.sh > $O/ev_bench_seeds.txt 2>&1
  ;;
zoo)     # the zoo head shapes: HIP-event sweep, rocprofv3 counters, the Splat(max) backward's point segments
  python tools/zoo_sweep.py > $O/ev_zoo_sweep.txt 2>&1
  bash tools/zoo_prof.sh > $O/ev_zoo_prof.log 2>&1
  cp $O/zoo_prof_report.txt $O/ev_zoo_counters.txt
  python tools/dev/nseg_sweep.py > $O/ev_nseg_sweep.txt 2>&1
  ;;
models)  # the three training steps: times, kernel families against their bounds
  for m in segmenter classifier inpainter; do python tools/${m}_step_bench.py 2>&1 | grep "training step"; done > $O/ev_model_steps.txt
  for m in segmenter classifier inpainter; do bash tools/family_roofline.sh $m > /dev/null 2>&1; done
  cat $O/family_roofline_segmenter.txt $O/family_roofline_classifier.txt $O/family_roofline_inpainter.txt > $O/ev_model_breakdown.txt
  bash tools/dev/emd_update_ab.sh > $O/ev_emd_ab.txt 2>&1
  ;;
tools)   # every bench tool quoted in DESIGN.md
  bash tools/dev/tools_snapshot.sh > $O/ev_tools_output.txt 2>&1
  ;;
esac
ls -la $O | grep " ev_" | head -40
