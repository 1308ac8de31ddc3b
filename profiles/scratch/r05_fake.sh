#!/bin/bash
# round 5: what the PaletteNeRF field's parts cost alone (timing-only FAKE builds; stand-alone op, 1.09 M rows)
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for v in product oldacc f2 f2w16 f1 f3 w8; do
  lib=$R/palettenerf_amd/libpnr_hip_$v.so; [ $v = product ] && lib=$R/palettenerf_amd/libpnr_hip.so
  echo "== $v"; PNR_LIB_PATH=$lib timeout 200 python profiles/field_kernel_bench.py --rows 1089480 335180 --prec f16x3 2>&1 | grep "palette field"
done > $O/fake.txt 2>&1
