#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py -x -q -m gpu -k "heads or fuse_field or palette_field" > $O/pytest_heads.log 2>&1; echo "rc $?" >> $O/pytest_heads.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l)
        for k, v in d['extra']['dropin'].items(): print(k, round(v.get('ms_per_step', 0), 2), 'ms', v.get('launches_per_frame'), 'launches', round(v.get('kernel_ms_per_frame', 0), 2), 'ms kernels', round(v.get('kernel_ms_hip_per_frame', 0), 2), 'ours', v.get('error'))" > $O/dropin_legs.log 2>&1
