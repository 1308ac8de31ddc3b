#!/usr/bin/env python3
"""Per-dispatch duration quantiles of the kernels whose name contains a substring, out of a rocprofv3 rocpd sqlite .db.
usage: kernel_quantiles.py results.db substring [substring ...]"""
import sqlite3
import sys


def main(path, subs):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else None
    if view is None:
        print("no 'kernels' view; tables:", tabs[:40])
        return
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else [k for k in cols if "name" in k][0]
    for sub in subs:
        rows = [r for r in c.execute(f"select start, end from {view} where {name_col} like ? order by start", (f"%{sub}%",))]
        d = sorted((e - s) / 1e3 for s, e in rows)
        if not d:
            print(sub, "no dispatches")
            continue
        q = lambda p: d[min(len(d) - 1, int(p * len(d)))]
        print(f"{sub}: n {len(d)} mean {sum(d) / len(d):.1f}  min {d[0]:.1f} p10 {q(.1):.1f} p25 {q(.25):.1f} p50 {q(.5):.1f} p75 {q(.75):.1f} p90 {q(.9):.1f} p99 {q(.99):.1f} max {d[-1]:.1f} us")
        # in dispatch order: mean of every group of 27 consecutive launches' k-th element is not meaningful across frames of different length; print the first 60 instead
        print("   first 60 in launch order:", " ".join(f"{(e - s) / 1e3:.0f}" for s, e in rows[:60]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
