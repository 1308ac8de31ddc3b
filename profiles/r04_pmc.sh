#!/bin/bash
# Round-4 PMC passes (separate rocprofv3 --pmc runs, kernel-trace only): HBM-side traffic of the lookup kernel for the three bench workloads,
# matrix-pipe / wave cycles, L2 hits; and the occupancy sweep's kernels.  Run on the GPU box from the repo root.
R=$PWD
for wl in lego lego_palette garden; do
  bash profiles/pmc_pass.sh r04_pmc_$wl "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" -- --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
  python3 profiles/make_traffic.py $wl gpurun_out/r04_pmc_$wl gpurun_out/r04_traffic.json
done

# the stand-alone lookup op (k_grid_fwd_d3c2 / k_grid_fwd) on the four batches of profiles/grid_op_bench.py: HBM-side bytes and L2 hits per launch
PNR_PMC_SCRIPT=profiles/grid_op_bench.py bash profiles/pmc_pass.sh r04_pmc_grid_op "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU" -- --once
