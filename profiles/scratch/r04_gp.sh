#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_d3pair.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "grid" > $O/pytest_d3pair.log 2>&1; echo "rc $?" >> $O/pytest_d3pair.log
for v in main d3pair; do
if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
timeout 300 python profiles/grid_op_bench.py > $O/grid_op_$v.log 2>&1
done
