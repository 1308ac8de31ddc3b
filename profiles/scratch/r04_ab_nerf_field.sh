#!/bin/bash
# A/B: NeRF field with each k-block's activations split in front of its own products (product) against all four splits up front
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload lego --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; print('$1', round(d['ms_per_step'], 3), 'ms', 'field alone', round(r['mfma']['avg_launch_ms'] * 1e3, 2), 'us', round(r['mfma']['frac'], 4))"
}
for round in 1 2 3; do
  for v in "product:" "splitfirst:$R/palettenerf_amd/libpnr_hip_splitfirst.so"; do run ${v%%:*} "${v#*:}"; done
done > $O/ab_nerf_field.log 2>&1
PNR_LIB_PATH= timeout 300 python profiles/field_kernel_bench.py --rows 365482 1089480 --prec f16x3 >> $O/ab_nerf_field.log 2>&1
PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_splitfirst.so timeout 300 python profiles/field_kernel_bench.py --rows 365482 1089480 --prec f16x3 >> $O/ab_nerf_field.log 2>&1
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q -m gpu -k "nerf or field or frame" > $O/pytest_field.log 2>&1; echo "rc $?" >> $O/pytest_field.log
