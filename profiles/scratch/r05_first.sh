#!/bin/bash
# round 5, first GPU call: the new tests + the N>1 default line + the palette field's phase timing + a baseline of garden / lego_palette
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_reference_kernels.py -x -q -m gpu \
  -k "pair or align_corners or palette_crops or more_than_one_rank or headline_crop or morton or fused_training_step" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
for wl in garden lego_palette; do
  PNR_LIB_PATH=$R/palettenerf_amd/libpnr_hip_paltiming.so timeout 300 python profiles/pal_timing.py --workload $wl > $O/pal_timing_$wl.txt 2>&1
done
for wl in garden lego_palette lego; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 > $O/base_$wl.json
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_garden -o garden -- python3 $R/bench.py --workload garden --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $O/prof_garden.log 2>&1
