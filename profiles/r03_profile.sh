#!/bin/bash
# Round-3 rocprofv3 kernel-trace summaries (run on the GPU box from the repo root): bench workloads, configs[3] training steps, the occupancy sweep.
# Writes gpurun_out/r03/<name>.txt; copy the ones to keep into profiles/r03_<name>.txt.
R=$PWD; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03
cd /tmp
prof() {  # name, program args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r03/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 3)"; python3 $R/profiles/summarize.py $db; } > $R/gpurun_out/r03/$name.txt
}
prof bench_lego $R/bench.py --workload lego --steps 15 --warmup 3 --no-cpu-baseline --no-extras
prof bench_lego_palette $R/bench.py --workload lego_palette --steps 15 --warmup 3 --no-cpu-baseline --no-extras
prof bench_garden $R/bench.py --workload garden --steps 10 --warmup 3 --no-cpu-baseline --no-extras
prof train_step_palette $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5
prof train_step_nerf $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5
prof occupancy_sweep $R/profiles/extra_state_bench.py
