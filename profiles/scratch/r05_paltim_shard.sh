#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_paltiming.so
{ python profiles/pal_timing.py --workload garden --shards 8 --frames 5; python profiles/pal_timing.py --workload garden --frames 3; } > $O/pal_timing_shard.txt 2>&1
