#!/bin/bash
# round 4, first GPU call: the new lookup op against the old one, compat-mode frames (what an unchanged run_cuda issues)
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grid" > $O/pytest_grid.log 2>&1; echo "pytest rc $?" >> $O/pytest_grid.log
timeout 600 python profiles/grid_op_bench.py > $O/grid_op_bench.log 2>&1
timeout 600 python bench.py --mode compat --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_lego_compat.log 2>&1
timeout 600 python bench.py --mode compat --model palette --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_palette_compat.log 2>&1
cd /tmp
prof() { name=$1; shift; rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $O/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 4)"; python3 $R/profiles/summarize.py $db; } > $O/$name.txt; }
prof bench_lego_compat $R/bench.py --mode compat --steps 10 --warmup 3 --no-cpu-baseline --no-extras
prof grid_op_once $R/profiles/grid_op_bench.py --once
