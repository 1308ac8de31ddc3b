#!/bin/bash
# round 5: the PaletteNeRF field with its aux_map rows in LDS -- parity tests, phase timing, A/B against the old form (same sources, knobs off)
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "palette" > $O/pytest_pal2.log 2>&1; echo "rc $?" >> $O/pytest_pal2.log
for wl in garden lego_palette; do
  PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_paltiming.so timeout 300 python profiles/pal_timing.py --workload $wl > $O/pal_timing2_$wl.txt 2>&1
done
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e9, 3), 'G/s', d['step_ms']['median'])"
}
for round in 1 2 3; do
  for v in "new:" "old:$R/palettenerf_amd/libpnr_hip_oldacc.so"; do
    run ${v%%:*} "${v#*:}" garden
    run ${v%%:*} "${v#*:}" lego_palette
  done
done > $O/ab_pal2.log 2>&1
