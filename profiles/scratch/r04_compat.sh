#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --workload lego --mode compat --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $O/bench_compat_f$i.log 2>&1; done
