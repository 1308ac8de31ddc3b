#!/bin/bash
# what is left of the training MLP kernels when three of four matrix instructions are dropped (PNR_MLP_FAKE=4: a timing experiment, results are wrong)?
R=$PWD; export TMPDIR=/tmp
for flags in "" "-DPNR_MLP_FAKE=4" "-DPNR_MLP_FAKE=16"; do
  touch palettenerf_amd/csrc/mlp.hip; PNR_EXTRA_HIPCC_FLAGS="$flags" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
  cd /tmp; rm -rf /tmp/ph_t
  rocprofv3 --kernel-trace --stats -d /tmp/ph_t -o p -- python3 $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5 > /tmp/ph_t.log 2>&1
  cd $R; echo "== [$flags]"
  python3 profiles/summarize.py $(find /tmp/ph_t -name '*.db' | head -1) | grep "total kernel\|k_mlp_" | cut -c1-110
done
