#!/usr/bin/env python3
"""HBM-side bytes per k_frame_grid launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/r03_traffic.json entry.
usage: make_traffic.py <workload (bench.py --workload name)> <dir with FETCH_SIZE/ and WRITE_SIZE/ pass sub-directories> [existing json]
Only launches that did work count (duration above 1/4 of the longest).  FETCH_SIZE (KiB) is doubled: gfx950 reports half of the
bytes of a coalesced read (MI355X_MICROARCH.md, HBM section; re-calibrated here on k_frame_field whose fetch is known);
WRITE_SIZE (KiB) is exact on this kernel's 128 B/row."""
import csv
import glob
import json
import os
import sys


def per_launch(root, counter):
    f = glob.glob(os.path.join(root, counter, "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_frame_grid" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
    keep = [i for i, d in enumerate(dur) if d > max(dur) / 4]
    vals = [float(rows[i]["Counter_Value"]) for i in keep]
    return sum(vals) / len(vals), len(vals)


def main(model, root, out):
    fetch, n = per_launch(root, "FETCH_SIZE")
    write, _ = per_launch(root, "WRITE_SIZE")
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[model] = {"kernel": "k_frame_grid", "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "launches_counted": n,
                   "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
                   "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --workload <this>` (profiles/r03_pmc_*.txt); FETCH_SIZE "
                           "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of coalesced read bytes; factor re-calibrated in round 1 on k_frame_field whose "
                           "fetch is known), WRITE_SIZE uncorrected; mean over the launches that did work; moving camera, density_scale 100, tile-ordered rays"}
    json.dump(data, open(out, "w"), indent=1)
    print(model, data[model]["traffic_bytes_per_launch"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "r03_traffic.json"))
