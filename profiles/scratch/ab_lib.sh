#!/bin/bash
# A/B of library variants (palettenerf_amd.build --variant NAME): bench.py headline legs per variant, interleaved rounds.  usage: ab_lib.sh ROUNDS base|NAME ...
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r06
rounds=$1; shift
for r in $(seq 1 $rounds); do
for v in "$@"; do
  if [ "$v" = base ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
  for wl in lego garden lego_palette; do
    ms=$(python3 $R/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.3f ms  lookup %.1f us' % (l['ms_per_step'], l['roofline']['avg_launch_ms']*1e3))")
    echo "round $r  $v  $wl  $ms"
  done
done
done
