#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-k}
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q -m gpu -k "palette or nerf or field" > $O/pytest_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_${TAG}.log
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', d['step_ms']['median'])"
}
for round in 1 2 3; do
  for v in $VARIANTS; do
    lib=$R/palettenerf_amd/libpnr_hip_$v.so; [ $v = product ] && lib=
    run $v "$lib" garden
    run $v "$lib" lego_palette
    run $v "$lib" lego
  done
done > $O/ab_${TAG}.log 2>&1
