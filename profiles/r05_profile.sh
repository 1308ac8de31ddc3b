#!/bin/bash
# Round-5 rocprofv3 kernel-trace summaries (run on the GPU box from the repo root): bench workloads (native and drop-in/compat), configs[3] training steps,
# the stand-alone lookup op at 2^20 ray-coherent samples, the occupancy sweep.  Writes gpurun_out/r05/<name>.txt; the ones kept are copied to profiles/r05_<name>.txt.
R=$PWD; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r05
cd /tmp
prof() {  # name, program args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r05/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 5)"; python3 $R/profiles/summarize.py $db;
    grep -o '"roofline": {"bound": "hbm", "kernel": "[^"]*", "achieved": [0-9.]*, "peak": [0-9.]*, "unit": "GB/s", "frac": [0-9.]*' $R/gpurun_out/r05/$name.log | head -1 | sed 's/^/# the same run, bench.py line: /';
    grep -o '"avg_launch_ms": [0-9.]*' $R/gpurun_out/r05/$name.log | head -1 | sed 's/^/# the same run, bench.py line (HIP events carried by the working lookup launches of the first timed step): /'; } > $R/gpurun_out/r05/$name.txt
}
prof bench_lego $R/bench.py --workload lego --no-cpu-baseline --no-extras --no-traffic     # (bench.py's default steps / warmup: the driver's command without the legs behind the headline)
prof bench_lego_palette $R/bench.py --workload lego_palette --steps 15 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_garden $R/bench.py --workload garden --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_lego_compat $R/bench.py --workload lego --mode compat --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_lego_palette_compat $R/bench.py --workload lego_palette --mode compat --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-traffic
prof grid_op_once $R/profiles/grid_op_bench.py --once
prof train_step_palette $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5
prof train_step_nerf $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5
prof occupancy_sweep $R/profiles/extra_state_bench.py
