#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_reference_kernels.py -x -q -m gpu > $O/pytest_refk.log 2>&1; echo "rc $?" >> $O/pytest_refk.log
