#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_frames.py -m gpu -x -q -k "march or reference or train" > $O/pytest_opmarch.log 2>&1; echo "rc $?" >> $O/pytest_opmarch.log
cd /tmp; rm -rf /tmp/prof_tp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf_main.log 2>&1
db=$(find /tmp/prof_tp -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db > $O/train_nerf_main.txt
