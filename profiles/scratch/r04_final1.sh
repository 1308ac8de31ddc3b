#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; echo "rc $?" >> $O/pytest_gpu_all.log
bash profiles/r04_profile.sh
cd $R
timeout 300 python profiles/train_step_bench.py --model palette --steps 50 --warmup 5 > $O/train_palette_h.log 2>&1
timeout 300 python profiles/train_step_bench.py --model nerf --steps 50 --warmup 5 > $O/train_nerf_h.log 2>&1
