#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
for cap in 0 1024 1250 640; do
  export PNR_MARCH_BLOCKS1=$cap
  for wl in lego garden; do
    timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cap $cap $wl', round(d['ms_per_step'], 3), 'ms', d['step_ms']['median'])"
  done
done; done > $O/cap1.log 2>&1
