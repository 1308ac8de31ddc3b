#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-t}
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_full_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_full_${TAG}.log
TAG=$TAG bash profiles/scratch/r05_gaps.sh
bash profiles/r05_shard.sh
for wl in lego lego_palette garden; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'], 3), 'ms', d['step_ms'])"
done > $O/bench_${TAG}.log 2>&1
