#!/bin/bash
# NOTE: the variant this script measured lost and its code was removed ("Experiments that lost"); kept as the record of how it was measured.
# cost-ordered first march launch (pnr_set_option march_order / PNR_MARCH_ORDER=1) against the plain order, same build, same box
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['step_ms']['median'],3), d['config'].get('rendered_samples_per_step'))"; }
R=$PWD; export TMPDIR=/tmp
prof() {
  cd /tmp; rm -rf /tmp/ph_x
  rocprofv3 --kernel-trace --stats -d /tmp/ph_x -o p -- python3 $R/bench.py --workload ${WL:-lego} --steps 15 --warmup 3 --no-cpu-baseline --no-extras $EXTRA > /dev/null 2>&1
  cd $R
  python3 profiles/summarize.py $(find /tmp/ph_x -name '*.db' | head -1) | grep "k_frame_march<true, true, 1>\|k_frame_order" | cut -c1-100
}
if [ -n "$CHECK" ]; then PNR_MARCH_ORDER=1 timeout 900 python -m pytest tests/test_gpu_frames.py -x -q -m gpu 2>&1 | tail -2; fi
for o in 0 1 0 1; do
  export PNR_MARCH_ORDER=$o
  echo "== order=$o lego: $(run) | static: $(run --static-pose) | garden: $(run --workload garden --steps 20)"
  prof; EXTRA=--static-pose prof; WL=garden prof
done
