#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-s}
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_full_${TAG}.log 2>&1; echo "rc $?" >> $O/pytest_full_${TAG}.log
TAG=$TAG WLS="garden lego_palette lego" bash profiles/scratch/r05_prof.sh
