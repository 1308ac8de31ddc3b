#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python profiles/grid_op_bench.py > $O/grid_op_bench.log 2>&1
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench rc $?" >> $O/bench_default.log
