#!/usr/bin/env python3
"""Per-iteration kernel durations and idle gaps of ONE frame of the native loop out of a rocprofv3 --kernel-trace .db (frame_trace.py reads the csv form).
usage: frame_trace_db.py results.db [frame index, default 10]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_frame_begin" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a, b = starts[k], starts[k + 1]
frame = rows[a:b]
t0 = frame[0][1]
prev_end, tot_gap, busy = None, 0.0, 0.0
print("   start_us     dur_us   gap_before_us  kernel")
for name, s, e in frame:
    gap = 0.0 if prev_end is None else max(0, s - prev_end) / 1e3
    tot_gap += gap
    busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:10.1f} {gap:12.1f}     {name.split('(')[0][-40:]}")
    prev_end = e
print(f"# frame {k}: {len(frame)} launches up to the next frame's first, busy {busy:.1f} us, idle between them {tot_gap:.1f} us")
