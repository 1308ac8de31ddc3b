#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frames.py -m gpu -x -q -k "palette" > $O/pytest_kargs.log 2>&1; echo "rc $?" >> $O/pytest_kargs.log
for i in 1 2 3; do
  for w in garden lego_palette; do
  timeout 300 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $O/bench_kargs_${w}_$i.log 2>&1
  PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_nokargs.so timeout 300 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $O/bench_nokargs_${w}_$i.log 2>&1
  done
done
