#!/bin/bash
# where do the PaletteNeRF field kernel's cycles go on the garden workload: LDS, VALU, waits (separate --pmc passes, kernel-trace only)
bash profiles/pmc_pass.sh r03_pmc_field_lds \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
  "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" \
  "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
  "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" \
  -- --workload garden --steps 8 --warmup 3 --no-cpu-baseline --no-extras
cat gpurun_out/r03_pmc_field_lds.txt | grep -v "^$" | cut -c1-260
