#!/bin/bash
# Round-6 rocprofv3 kernel-trace summaries (run on the GPU box from the repo root): bench workloads (native, the reference's -O mode, drop-in / compat with and
# without dropin.fuse_field), configs[3] training steps, one garden shard against the whole frame, the N > 1 default line over a one-rank communicator.
# Writes gpurun_out/r06/<name>.txt; the ones kept are copied to profiles/r06_<name>.txt.
R=$PWD; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
cd /tmp
prof() {  # name, program args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r06/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 6)"; python3 $R/profiles/summarize.py $db;
    grep -o '"roofline": {"bound": "[a-z0-9]*", "kernel": "[^"]*", "achieved": [0-9.]*, "peak": [0-9.]*, "unit": "GB/s", "frac": [0-9.]*' $R/gpurun_out/r06/$name.log | head -1 | sed 's/^/# the same run, bench.py line: /';
    grep -o '"avg_launch_ms": [0-9.]*' $R/gpurun_out/r06/$name.log | head -1 | sed 's/^/# the same run, bench.py line (HIP events carried by the working lookup launches of the first timed step): /';
    grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/r06/$name.log | head -1 | sed 's/^/# the same run, bench.py line: /'; } > $R/gpurun_out/r06/$name.txt
}
prof bench_lego $R/bench.py --workload lego --no-cpu-baseline --no-extras --no-traffic     # (bench.py's default steps / warmup: the driver's command without the legs behind the headline)
prof bench_lego_fp16 $R/bench.py --workload lego --fp16 --no-cpu-baseline --no-extras --no-traffic     # the reference's -O mode: k_frame_grid_h1 (fp16 table, at::Half accumulator)
prof bench_lego_palette $R/bench.py --workload lego_palette --steps 15 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_garden $R/bench.py --workload garden --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_lego_compat $R/bench.py --workload lego --mode compat --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
prof bench_lego_palette_compat $R/bench.py --workload lego_palette --mode compat --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-traffic
prof dropin_fuse_field $R/profiles/dropin_legs.py
prof train_step_palette $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5
prof train_step_nerf $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5
for s in 8 1; do
  rm -rf /tmp/prof_shard$s
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_shard$s -o p -- python3 $R/profiles/shard_profile.py $s > $R/gpurun_out/r06/shard$s.log 2>&1
  db=$(find /tmp/prof_shard$s -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 profiles/shard_profile.py $s   (round 6)"; grep "^shards" $R/gpurun_out/r06/shard$s.log; python3 $R/profiles/summarize.py $db | head -16; python3 $R/profiles/frame_gaps.py $db; python3 $R/profiles/frame_boundary.py $db; } > $R/gpurun_out/r06/shard$s.txt
done
cd $R
python3 profiles/flex_bench.py > gpurun_out/r06/flex_bench.txt 2>&1
PNR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29553 python3 bench.py --dist-default --no-cpu-baseline > gpurun_out/r06/bench_dist_default.json 2> gpurun_out/r06/bench_dist_default.err
python3 bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
