#!/bin/bash
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r05
cd /tmp
name=bench_lego
rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 $R/bench.py --workload lego --no-cpu-baseline --no-extras --no-traffic > $R/gpurun_out/r05/$name.log 2>&1
db=$(find /tmp/prof_$name -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload lego --no-cpu-baseline --no-extras --no-traffic   (round 5; bench.py's default steps / warmup)"; python3 $R/profiles/summarize.py $db;
  grep -o '"avg_launch_ms": [0-9.]*' $R/gpurun_out/r05/$name.log | head -1 | sed 's/^/# the same run, bench.py line (HIP events carried by the working lookup launches of the first timed step): /';
  grep -o '"frac": [0-9.]*' $R/gpurun_out/r05/$name.log | head -1 | sed 's/^/# the same run, bench.py line roofline /';
  python3 $R/profiles/frame_launch_avgs.py $db 5 20 | grep "^#"; } > $R/gpurun_out/r05/$name.txt
for wl in lego_palette garden; do
  name=bench_$wl; rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $R/gpurun_out/r05/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic   (round 5)"; python3 $R/profiles/summarize.py $db; } > $R/gpurun_out/r05/$name.txt
done
