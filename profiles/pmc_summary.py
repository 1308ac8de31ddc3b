#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, csv output) per kernel of interest.
usage: pmc_summary.py <dir with one sub-directory per pass>"""
import collections
import csv
import glob
import os
import sys

KERNELS = {"k_grid_fwd_d3c2": "grid_fwd_d3c2", "k_grid_fwd<": "grid_fwd_generic", "k_nerf_field_fwd": "nerf_field_fwd", "k_march_rays": "march_rays", "k_composite_rays": "composite_rays",
           "k_frame_grid": "frame_grid", "k_frame_field": "frame_field", "k_palette_field_fwd": "palette_field", "k_frame_march": "frame_march",
           "k_frame_composite": "frame_composite", "k_occ_lookup": "occ_lookup", "k_occ_sigma": "occ_sigma", "k_occ_ema": "occ_ema", "k_occ_pack": "occ_pack", "k_occ_points": "occ_points",
           "k_fused": "fused", "k_bin_gather": "bin_gather", "k_bin_scatter": "bin_scatter", "k_bin_count": "bin_count", "k_linear_wgrad<": "linear_wgrad", "k_mlp_bwd": "mlp_bwd", "k_mlp_fwd": "mlp_fwd", "k_coarse_image": "coarse_image", "k_march_train_count": "march_train_count", "k_grid_fwd_d3c2_pair": "grid_fwd_pair"}


def main(root):
    print(f"# source: {root} (rocprofv3 --pmc <counters> --kernel-trace, one pass per sub-directory; values are per-dispatch means)")
    for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            for pat, short in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur[short][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"\n## pass {os.path.basename(os.path.dirname(f))}")
        for k, v in agg.items():
            d = list(dur[k].values())
            line = ", ".join(f"{c}={sum(x) / len(x):.1f}" for c, x in sorted(v.items()))
            print(f"{k:16s} dispatches={len(d):4d} mean_us={sum(d) / len(d):8.2f}  {line}")


if __name__ == "__main__":
    main(sys.argv[1])
