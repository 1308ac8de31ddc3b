#!/bin/bash
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r05
cd /tmp; rm -rf /tmp/prof_x
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_x -o p -- python3 $R/bench.py --workload lego --no-cpu-baseline --no-extras --no-traffic > $R/gpurun_out/r05/x.log 2>&1
db=$(find /tmp/prof_x -name '*.db' | head -1)
python3 $R/profiles/frame_launch_avgs.py $db > $R/gpurun_out/r05/frame_launch_avgs.txt
grep -o '"avg_launch_ms": [0-9.]*' $R/gpurun_out/r05/x.log | head -1 >> $R/gpurun_out/r05/frame_launch_avgs.txt
