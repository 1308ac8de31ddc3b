#!/bin/bash
# A/B: frame lookup with the row index formed per kind of level (product) against the general form for every level
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; print('$1', '$3', round(d['ms_per_step'], 3), 'ms', 'lookup', round(r['avg_launch_ms'] * 1e3, 2), 'us frac', round(r['frac'], 3), 'alone', round(r.get('lookup_alone', {}).get('avg_launch_ms', 0) * 1e3, 2), round(r.get('lookup_alone', {}).get('frac', 0), 3))"
}
for round in 1 2 3; do
  for v in "product:" "nokind:$R/palettenerf_amd/libpnr_hip_nokind.so"; do
    run ${v%%:*} "${v#*:}" lego
    run ${v%%:*} "${v#*:}" garden
  done
done > $O/ab_kind.log 2>&1
timeout 900 python -m pytest tests/test_gpu_frames.py -x -q -m gpu > $O/pytest_frames.log 2>&1; echo "rc $?" >> $O/pytest_frames.log
