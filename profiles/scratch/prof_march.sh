R=$PWD; export TMPDIR=/tmp
for w in 6 4; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_WAVES_Q=$w" python -m palettenerf_amd.build >/dev/null 2>&1
  cd /tmp; rm -rf /tmp/pm_$w
  rocprofv3 --kernel-trace --stats -d /tmp/pm_$w -o p -- python3 $R/bench.py --workload lego --steps 15 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  echo "== waves_q $w"; python3 profiles/summarize.py $(find /tmp/pm_$w -name '*.db' | head -1) | head -9 | cut -c1-150
done
