#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_mstats.so
{ python profiles/march_stats.py 2>&1 | tail -12; python profiles/march_stats.py --workload garden 2>&1 | tail -6; } > $O/march_stats.txt
