#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_fullsize.py -m gpu -x -q -k "frame or native or crop or layout" > $O/pytest_xpair.log 2>&1; echo "rc $?" >> $O/pytest_xpair.log
for i in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $O/bench_xpair_$i.log 2>&1
  PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_noxpair.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $O/bench_noxpair_$i.log 2>&1
done
