#!/bin/bash
# Round-4 PMC passes over the configs[3] training step (PaletteNeRF): instruction mix and busy cycles of the MLP and table-gradient kernels.
# Separate rocprofv3 --pmc runs, kernel-trace only.  Run on the GPU box from the repo root.
PNR_PMC_SCRIPT=profiles/train_step_bench.py bash profiles/pmc_pass.sh r04_pmc_train "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" -- --model palette --steps 6 --warmup 2
