#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
for v in main d3b128w8 d3b512w8 d3b1024w8 d3b256w4 d3b64w8; do
if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
timeout 300 python profiles/grid_op_bench.py > $O/grid_op_$v.log 2>&1
done
