#!/bin/bash
# Round-6 PMC passes (separate rocprofv3 --pmc runs, kernel-trace only): HBM-side traffic, matrix-pipe / wave cycles, instruction mix and waits of the headline,
# the PaletteNeRF frames and the reference's -O mode (fp16 table).  Run on the GPU box from the repo root.
R=$PWD
for wl in lego garden lego_palette; do
  bash profiles/pmc_pass.sh r06_pmc_$wl "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" \
       "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
done
bash profiles/pmc_pass.sh r06_pmc_lego_fp16 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- --workload lego --fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
