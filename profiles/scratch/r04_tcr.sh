#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
for v in main tcr1 tcr2; do
if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
timeout 300 python profiles/scratch/train_coop_ab.py > $O/train_coop_$v.log 2>&1
done
