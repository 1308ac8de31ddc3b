#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-m}
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_frames.py -q -m gpu -x -k "mlp or train or grad" > $O/pytest_mlp_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_mlp_${TAG}.log
cd /tmp
for m in palette nerf; do
  rm -rf /tmp/prof_ts_$m
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts_$m -o p -- python3 $R/profiles/train_step_bench.py --model $m --steps 20 --warmup 5 > $O/ts_${m}_${TAG}.log 2>&1
  db=$(find /tmp/prof_ts_$m -name '*.db' | head -1)
  python3 $R/profiles/summarize.py $db | head -14 > $O/ts_${m}_${TAG}.txt
done
