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


if __name__ == "__main__":
    main(sys.argv[1])
