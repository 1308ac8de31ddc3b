#!/bin/bash
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r05
cd /tmp; rm -rf /tmp/prof_y
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_y -o p -- python3 $R/bench.py --workload lego --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $R/gpurun_out/r05/y.log 2>&1
db=$(find /tmp/prof_y -name '*.db' | head -1)
python3 $R/profiles/frame_trace_db.py $db 10 > $R/gpurun_out/r05/frame_trace_lego.txt
