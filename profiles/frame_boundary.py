#!/usr/bin/env python3
"""The launches around one frame boundary of the native loop out of a rocprofv3 --kernel-trace .db (start relative to the end of the frame's last launch,
duration, name): what the device does -- and waits for -- between two frames.  usage: frame_boundary.py results.db [rows, default 14]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ends = [i for i, r in enumerate(rows) if "k_frame_unsort" in r[0] and r[2] - r[1] > 6000]     # (the speculative launches that found the frame unfinished are ~3 us)
e = ends[len(ends) // 2]
t0 = rows[e][2]
for r in rows[e - 3:e + n]:
    print(f"  {(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:7.1f}  {r[0][:100]}")
