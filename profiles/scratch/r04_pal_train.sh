#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp; rm -rf /tmp/prof_tp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5 > $O/train_palette_now.log 2>&1
db=$(find /tmp/prof_tp -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db > $O/train_palette_now.txt
