#!/bin/bash
# round 5: the staged (coalesced) record scatter of the binned table gradient: parity, then the NeRF / PaletteNeRF training steps under rocprofv3, staged on / off
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q -k "grid or binned or train or converge or crops" > $O/pytest_bins.log 2>&1; echo "rc $?" >> $O/pytest_bins.log
for v in staged plain; do
  [ $v = plain ] && export PNR_NO_SCATTER_STAGED=1
  for m in nerf palette; do
    cd /tmp; rm -rf /tmp/prof_tp
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model $m --steps 20 --warmup 5 > $O/train_${m}_$v.log 2>&1
    db=$(find /tmp/prof_tp -name '*.db' | head -1)
    python3 $R/profiles/summarize.py $db > $O/train_${m}_$v.txt
  done
done
