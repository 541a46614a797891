# per batch (CLOUDCT_EMD_SINGLE_UPDATE=1), same box
run() { python tools/dev/emd_inpainter_bidders.py 2>&1 | grep "50 iterations\|oracle"; python tools/loss_bench.py 2>&1 | grep -i emd; }
echo "== several workgroups, alone up to 4096 entries"; run
echo "== one workgroup per batch"; CLOUDCT_EMD_SINGLE_UPDATE=1 run
echo "== inpainter step, several workgroups / one"; python tools/inpainter_step_bench.py 2>&1 | tail -1 | grep -o "training step.*"; CLOUDCT_EMD_SINGLE_UPDATE=1 python tools/inpainter_step_bench.py 2>&1 | tail -1 | grep -o "training step.*"
