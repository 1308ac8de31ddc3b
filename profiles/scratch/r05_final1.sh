#!/bin/bash
# round 5: parity of the last kernel change, PMC passes, the reference's kernels next to ours at HEAD
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "palette" > $O/pytest_final1.log 2>&1; echo "rc $?" >> $O/pytest_final1.log
bash profiles/r05_pmc.sh > $O/pmc.log 2>&1
timeout 600 python profiles/reference_kernels.py --json > $O/reference_kernels.json 2> $O/reference_kernels.err
timeout 600 python profiles/reference_kernels.py > $O/reference_kernels.txt 2>&1
