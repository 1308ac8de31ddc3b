#!/bin/bash
# A/B of the PaletteNeRF field's wave-major tile schedule (product) against the workgroup-major one, garden / lego_palette frames and the 8-shard emulation
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
run() { # label libpath workload extra
  PNR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic $4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); e = d.get('extra', {}).get('shard_emulation'); print('$1', '$3', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e9, 3), 'G/s', [round(x, 3) for x in e['ms']] if e else '')"
}
for round in 1 2 3; do
  for v in "product:" "wgmajor:$R/palettenerf_amd/libpnr_hip_wgmajor.so"; do
    run ${v%%:*} "${v#*:}" garden "--shard-emulation 8"
    run ${v%%:*} "${v#*:}" lego_palette
  done
done > $O/ab_palette3.log 2>&1
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "palette or garden" > $O/pytest_palette.log 2>&1; echo "rc $?" >> $O/pytest_palette.log
