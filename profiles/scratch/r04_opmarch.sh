#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "march or reference or train" > $O/pytest_opmarch.log 2>&1; echo "rc $?" >> $O/pytest_opmarch.log
timeout 300 python profiles/reference_kernels.py --json > $O/reference_kernels2.log 2>&1
timeout 300 python profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf2.log 2>&1
PNR_NO_TRAIN_COOP=1 timeout 300 python profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf2_nocoop.log 2>&1
