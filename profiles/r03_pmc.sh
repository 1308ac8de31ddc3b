#!/bin/bash
# Round-3 PMC passes (separate rocprofv3 --pmc runs, kernel-trace only): HBM-side traffic of the lookup kernel for the three bench workloads,
# matrix-pipe / wave cycles, L2 hits; and the occupancy sweep's kernels.  Run on the GPU box from the repo root.
R=$PWD
for wl in lego lego_palette garden; do
  bash profiles/pmc_pass.sh r03_pmc_$wl "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" -- --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras
  python3 profiles/make_traffic.py $wl gpurun_out/r03_pmc_$wl gpurun_out/r03_traffic.json
done
PNR_PMC_SCRIPT=profiles/extra_state_bench.py bash profiles/pmc_pass.sh r03_pmc_occupancy "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" --
