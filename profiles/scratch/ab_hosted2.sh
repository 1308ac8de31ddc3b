run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
R=$PWD; export TMPDIR=/tmp
prof() {
  cd /tmp; rm -rf /tmp/ph_x
  rocprofv3 --kernel-trace --stats -d /tmp/ph_x -o p -- python3 $R/bench.py --workload ${WL:-lego} --steps 15 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python3 profiles/summarize.py $(find /tmp/ph_x -name '*.db' | head -1) | head -8 | tail -5 | grep "k_frame_grid\|k_frame_march" | cut -c1-100
}
for flags in "" "-DPNR_HOSTED_MIN_GLOG=2u" "-DPNR_HOSTED_MIN_GLOG=2u -DPNR_MARCH2_JUMPS=false -DPNR_MARCH_WAVES_Q=6" "-DPNR_HOSTED_MIN_GLOG=2u -DPNR_MARCH2_JUMPS=false -DPNR_MARCH_WAVES_Q=6 -DPNR_HOSTED_JUMPS=true" "-DPNR_HOSTED_MIN_GLOG=2u -DPNR_HOSTED_JUMPS=true"; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="$flags" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
  echo "== [$flags] lego: $(run) | $(run)  garden: $(run --workload garden --steps 20)"
  prof
done
