#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-t}
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_full_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_full_${TAG}.log
{ python profiles/frame_host_time.py --workload garden --shards 8; python profiles/frame_host_time.py --workload garden; python profiles/frame_host_time.py --workload lego; python profiles/frame_host_time.py --workload lego_palette; } > $O/host_time_${TAG}.txt 2>&1
{ python profiles/frame_looks.py --workload lego; python profiles/frame_looks.py --workload garden --frames 30; python profiles/frame_looks.py --workload lego_palette --frames 30; } > $O/looks_${TAG}.txt 2>&1
for wl in lego lego_palette garden; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'], 3), 'ms', d['step_ms'])"
done > $O/bench_${TAG}.log 2>&1
python profiles/shard_profile.py 8 > $O/shard8_plain_${TAG}.log 2>&1
python profiles/shard_profile.py 1 >> $O/shard8_plain_${TAG}.log 2>&1
