#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "binned or encode_mlp" > $O/pytest_bins.log 2>&1; echo "rc $?" >> $O/pytest_bins.log
for v in main fwdjobs; do
cd /tmp; rm -rf /tmp/prof_tp
if [ $v = main ]; then unset PNR_LIB_PATH; else export PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_$v.so; fi
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o p -- python3 $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5 > $O/train_nerf_$v.log 2>&1
db=$(find /tmp/prof_tp -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db > $O/train_nerf_$v.txt
done
