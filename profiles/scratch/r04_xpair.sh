#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_palpairs.so timeout 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "palette" > $O/pytest_palpairs.log 2>&1; echo "rc $?" >> $O/pytest_palpairs.log
for i in 1 2 3; do
  for v in main palpairs; do
  if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
  for w in garden lego_palette; do
  timeout 300 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $O/bench_${v}_${w}_$i.log 2>&1
  done
  done
done
