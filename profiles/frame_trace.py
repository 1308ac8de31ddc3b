#!/usr/bin/env python3
"""Per-iteration kernel durations of ONE frame out of a rocprofv3 --kernel-trace --output-format csv run of bench.py:
usage: frame_trace.py <kernel_trace.csv> [frame index counted from the end, default 2]
Prints, for every iteration of that frame, the march / lookup / field / composite launch times (us) and the gaps between launches."""
import csv
import sys


def main(path, back=2):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_frame_init" in r["Kernel_Name"]]
    a = starts[-back]
    b = starts[-back + 1] if back > 1 else len(rows)
    frame = rows[a:b]
    t0 = int(frame[0]["Start_Timestamp"])
    it, line, prev_end, busy = -1, {}, None, 0
    print("iter   march    grid   field    comp   (us; gap = idle time in front of the launch)")
    for r in frame:
        name = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = (e - s) / 1e3
        kind = "march" if "k_frame_march" in name else "grid" if "k_frame_grid" in name else "field" if ("k_frame_field" in name or "k_palette_field" in name) else "comp" if "k_frame_composite" in name else None
        if kind is None:
            continue
        if kind == "march":
            if line:
                print(f"{it:4d} {line.get('march', 0):7.1f} {line.get('grid', 0):7.1f} {line.get('field', 0):7.1f} {line.get('comp', 0):7.1f}   gaps {line.get('gap', 0):5.1f}")
            it += 1
            line = {"gap": 0.0}
        line[kind] = d
        busy += d
        if prev_end is not None:
            line["gap"] += max(0, s - prev_end) / 1e3
        prev_end = e
    print(f"frame: {(int(frame[-1]['End_Timestamp']) - t0) / 1e3:.1f} us wall, {busy:.1f} us in the four loop kernels, {it + 1} iterations launched")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
