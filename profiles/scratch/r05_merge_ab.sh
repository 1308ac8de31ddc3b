#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
for v in merge nomerge; do
  [ $v = nomerge ] && export PNR_NO_CELL_MERGE=1
  cd /tmp; rm -rf /tmp/prof_tp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf_$v.log 2>&1
  db=$(find /tmp/prof_tp -name '*.db' | head -1)
  python3 $R/profiles/summarize.py $db > $O/train_nerf_$v.txt
done
