#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-n}
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_full_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_full_${TAG}.log
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e9, 3), 'G/s', d['step_ms']['median'])"
}
for round in 1 2 3; do
  for v in "new:" "prev:$R/palettenerf_amd/libpnr_hip_prevfield.so"; do run ${v%%:*} "${v#*:}" lego; done
done > $O/ab_${TAG}.log 2>&1
