#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-t}
for spin in 1 0; do
  if [ $spin = 0 ]; then export PNR_NO_SPIN_WAIT=1; else unset PNR_NO_SPIN_WAIT; fi
  echo "== spin $spin"
  python profiles/frame_host_time.py --workload garden --shards 8; python profiles/frame_host_time.py --workload garden; python profiles/frame_host_time.py --workload lego; python profiles/frame_host_time.py --workload lego_palette
  for rep in 1 2; do
  for wl in lego lego_palette garden; do
    timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'], 3), 'ms', d['step_ms'])"
  done; done
  python profiles/shard_profile.py 8
  python profiles/shard_profile.py 1
done > $O/spin_${TAG}.log 2>&1
