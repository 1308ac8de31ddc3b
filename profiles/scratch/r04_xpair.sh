#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_fwm.log 2>&1; echo "rc $?" >> $O/pytest_fwm.log
for i in 1 2 3; do
  for v in main fwm0; do
  if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $O/bench_${v}_$i.log 2>&1
  done
done
