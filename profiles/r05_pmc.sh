#!/bin/bash
# Round-5 PMC passes (separate rocprofv3 --pmc runs, kernel-trace only): HBM-side traffic, matrix-pipe / wave cycles, instruction mix and waits of the
# PaletteNeRF frames (the 16-wave slab-free field kernel) and of the headline.  Run on the GPU box from the repo root.
R=$PWD
for wl in garden lego_palette lego; do
  bash profiles/pmc_pass.sh r05_pmc_$wl "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" \
       "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
done
