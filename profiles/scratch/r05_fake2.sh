#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
for v in product f4 f2 w12 f4w12; do
  lib=$R/palettenerf_amd/libpnr_hip_$v.so; [ $v = product ] && lib=$R/palettenerf_amd/libpnr_hip.so
  echo "== $v"; PNR_LIB_PATH=$lib timeout 200 python profiles/field_kernel_bench.py --rows 1089480 335180 --prec f16x3 --reps 60 2>&1 | grep "palette field"
done; done > $O/fake2.txt 2>&1
