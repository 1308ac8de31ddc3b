#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sigma_geo or train or grad or converge or sh" > $O/pytest_seam.log 2>&1; echo "rc $?" >> $O/pytest_seam.log
cd /tmp; rm -rf /tmp/prof_tp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf_seam.log 2>&1
db=$(find /tmp/prof_tp -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db > $O/train_nerf_seam.txt
