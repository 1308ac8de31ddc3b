#!/bin/bash
# A/B: PaletteNeRF field with a block's weight fragments requested ahead of its activation split (product) against the compiler-placed reads
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
run() { PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e9, 3), 'G/s')"
}
for round in 1 2 3; do
  for v in "product:" "nowahead:$R/palettenerf_amd/libpnr_hip_nowahead.so"; do
    run ${v%%:*} "${v#*:}" garden
    run ${v%%:*} "${v#*:}" lego_palette
  done
done > $O/ab_palette4.log 2>&1
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q -m gpu -k "palette" > $O/pytest_palette.log 2>&1; echo "rc $?" >> $O/pytest_palette.log
