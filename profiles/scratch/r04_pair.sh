#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pair or train or palette or grad or converge" > $O/pytest_pair.log 2>&1; echo "rc $?" >> $O/pytest_pair.log
cd /tmp; rm -rf /tmp/prof_tp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5 > $O/train_palette_pair.log 2>&1
db=$(find /tmp/prof_tp -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db > $O/train_palette_pair.txt
cd $R
timeout 300 python profiles/train_step_bench.py --model palette --steps 50 --warmup 5 > $O/train_palette_h.log 2>&1
