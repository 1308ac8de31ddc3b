#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) out of a rocpd sqlite .db into text.
usage: summarize.py results.db > summary.txt"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"# source: {path}\n# total kernel time: {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, tot, avg, pct in rows[:top]:
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {short}")
    # The frame loops enqueue one iteration more than a frame needs (its launches find nothing to do: ~4 us each), and bench.py renders a small crop for its parity
    # check: both count as dispatches above and pull a kernel's plain average below what bench.py's own events report (its `roofline.avg_launch_ms` is over the
    # launches that did work, timed steps only).  The comparable figure: launches that ran at least a quarter of the kernel's median.
    try:
        per = {}
        for name, start, end in c.execute("select name, start, end from kernels"):
            if "k_frame_grid" in name or "k_frame_field" in name or "k_palette_field" in name or "k_frame_march" in name or "k_composite_rays_flex" in name:
                per.setdefault(name.split("(")[0][-48:], []).append((end - start) / 1e3)
        if per:
            print("# launches that did work (duration >= a quarter of the kernel's median): what bench.py's in-run events average over")
        for name, d in per.items():
            d.sort()
            med = d[len(d) // 2]
            w = [x for x in d if x >= 0.25 * med]
            print(f"#   {len(w):5d} of {len(d):5d} launches  avg {sum(w) / len(w):8.2f} us  median {med:8.2f} us  {name}")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
