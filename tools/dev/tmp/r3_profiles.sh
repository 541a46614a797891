set -u
cd /root/repo
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
bash tools/pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1
python3 tools/pmc_sq.py > gpurun_out/bench_sq_counters.txt 2>&1
python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
python3 bench.py --feat 4 --no-cpu-baseline > gpurun_out/bench_line_C4.json 2>/dev/null
python3 bench.py --reduce sum --no-cpu-baseline > gpurun_out/bench_line_sum.json 2>/dev/null
python3 bench.py --mode ddp-step --no-cpu-baseline > gpurun_out/bench_line_ddp_step.json 2>/dev/null
bash tools/mfma_util.sh > gpurun_out/mfma_util.log 2>&1
bash tools/model_prof.sh > gpurun_out/model_prof.log 2>&1
bash tools/run_all_benches.sh > gpurun_out/tools_output.txt 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/mfma_pmc_* gpurun_out/mfma_stats_*
ls gpurun_out
tail -3 gpurun_out/bench_line.json
