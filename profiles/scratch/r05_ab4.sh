#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-k}
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 4), 'ms', d['step_ms']['median'], 'looks', d['config'].get('host_looks_per_frame'))"
}
for round in 1 2 3; do
  for v in $VARIANTS; do
    lib=$R/palettenerf_amd/libpnr_hip_$v.so; [ $v = product ] && lib=
    run $v "$lib" lego
    run $v "$lib" lego_palette
  done
done > $O/ab_${TAG}.log 2>&1
