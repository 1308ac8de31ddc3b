#!/bin/bash
# round 5 final measurements: same commands as profiles/r04_profile.sh (kernel-trace summaries), the default bench line, the N > 1 default over a one-rank communicator,
# the reference-kernel comparison with the bare-call timing, the 5k-step training run
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
bash profiles/r05_profile.sh > $O/profile.log 2>&1
cd $R
(time python bench.py) > $O/bench_default.json 2> $O/bench_default.err
PNR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 timeout 900 python bench.py --dist-default --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dist_default.json 2> $O/bench_dist_default.err
timeout 600 python profiles/reference_kernels.py --json > $O/reference_kernels.json 2> $O/reference_kernels.err
timeout 900 python profiles/train_palette.py > $O/train_palette_5k.txt 2>&1
