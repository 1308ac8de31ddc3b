#!/bin/bash
# NOTE: the variant this script measured lost and its code was removed ("Experiments that lost"); kept as the record of how it was measured.
# palette-field A/B on one box: the committed kernel (git stash of the working copy is not available on the box: BASE=<file> holds the old source) against the working copy
run() { python bench.py --steps 20 --warmup 4 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['step_ms']['median'],3))"; }
R=$PWD; export TMPDIR=/tmp
prof() {
  cd /tmp; rm -rf /tmp/ph_x
  rocprofv3 --kernel-trace --stats -d /tmp/ph_x -o p -- python3 $R/bench.py --workload garden --steps 8 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python3 profiles/summarize.py $(find /tmp/ph_x -name '*.db' | head -1) | head -8 | tail -4 | cut -c1-110
}
if [ -n "$CHECK" ]; then timeout 1200 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q -m gpu -k "palette" 2>&1 | tail -2; fi
echo "== new  garden: $(run --workload garden) | $(run --workload garden)   palette800: $(run --workload lego_palette)"
prof
if [ -f "$BASE" ]; then
  cp palettenerf_amd/csrc/palette_field.hip /tmp/pf_new.hip; cp $BASE palettenerf_amd/csrc/palette_field.hip
  python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
  echo "== base garden: $(run --workload garden) | $(run --workload garden)   palette800: $(run --workload lego_palette)"
  prof
  cp /tmp/pf_new.hip palettenerf_amd/csrc/palette_field.hip
fi
