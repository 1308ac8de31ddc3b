#!/bin/bash
# round 5: the slab-free wide tail at 16 / 12 waves against the round-4 kernel (HEAD's palette_field.hip): parity, then A/B in frames and stand-alone
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-e}
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q -m gpu -k "palette" > $O/pytest_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_${TAG}.log
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e9, 3), 'G/s', d['step_ms']['median'])"
}
for round in 1 2 3; do
  for v in "w16:" "w12:$R/palettenerf_amd/libpnr_hip_w12.so" "head:$R/palettenerf_amd/libpnr_hip_head.so"; do
    run ${v%%:*} "${v#*:}" garden
    run ${v%%:*} "${v#*:}" lego_palette
  done
done > $O/ab_${TAG}.log 2>&1
for v in product w12 head; do
  lib=$R/palettenerf_amd/libpnr_hip_$v.so; [ $v = product ] && lib=$R/palettenerf_amd/libpnr_hip.so
  echo "== $v"; PNR_LIB_PATH=$lib timeout 200 python profiles/field_kernel_bench.py --rows 1089480 335180 --prec f16x3 2>&1 | grep "palette field"
done > $O/alone_${TAG}.txt 2>&1
