#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q -k "grid or binned or train or converge" > $O/pytest_bins.log 2>&1; echo "rc $?" >> $O/pytest_bins.log
for m in nerf palette; do
  cd /tmp; rm -rf /tmp/prof_tp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model $m --steps 20 --warmup 5 > $O/train_${m}_v3.log 2>&1
  db=$(find /tmp/prof_tp -name '*.db' | head -1)
  python3 $R/profiles/summarize.py $db > $O/train_${m}_v3.txt
done
cd $R
timeout 900 python profiles/train_palette.py --dead-rows > $O/train_palette_5k.txt 2>&1
