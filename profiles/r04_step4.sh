#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1 ) 2> $O/bench_time.log; echo "bench rc $?" >> $O/bench_default.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
