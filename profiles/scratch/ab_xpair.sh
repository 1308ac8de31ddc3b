#!/bin/bash
# NOTE: the variant this script measured lost and its code was removed ("Experiments that lost"); kept as the record of how it was measured.
# x-pair gathers (PNR_GRID_XPAIR) against the default lookup: correctness (frame tests on the variant build), then lego timings + kernel summary per variant
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
R=$PWD; export TMPDIR=/tmp
prof() {
  cd /tmp; rm -rf /tmp/ph_x
  rocprofv3 --kernel-trace --stats -d /tmp/ph_x -o p -- python3 $R/bench.py --workload ${1:-lego} --steps 15 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python3 profiles/summarize.py $(find /tmp/ph_x -name '*.db' | head -1) | head -9 | tail -5 | cut -c1-100
}
IFS='|' read -ra SETS <<< "$FLAGSETS"
for flags in "${SETS[@]}"; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="$flags" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
  if [ -n "$CHECK" ] && [ -n "$flags" ] && [ "$flags" != " " ]; then timeout 900 python -m pytest tests/test_gpu_frames.py -x -q -m gpu -k "native or golden or hosted or fusion" 2>&1 | tail -2; fi
  echo "== [$flags] lego: $(run) | $(run) | $(run --fp16)"
  prof lego
done
