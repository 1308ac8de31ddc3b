#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
for i in 1 2 3; do
  for v in main gw6 gw8; do
  if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $O/bench_${v}_$i.log 2>&1
  done
done
