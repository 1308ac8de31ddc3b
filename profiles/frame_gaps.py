#!/usr/bin/env python3
"""Busy time, in-frame gaps and between-frame gaps of the native loop from a rocprofv3 --kernel-trace .db of bench.py.
usage: frame_gaps.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_frame_begin" in r[0]] or [i for i, r in enumerate(rows) if "k_frame_sort_inputs" in r[0]] or [i for i, r in enumerate(rows) if "k_frame_init" in r[0]]
ends = [i for i, r in enumerate(rows) if "k_frame_unsort" in r[0]]
frames = []
for s in starts:
    nxt = [e for e in ends if e > s]
    if nxt:
        frames.append((s, nxt[0]))
for s, e in frames[-2:]:
    fr = rows[s:e + 1]
    busy = sum(r[2] - r[1] for r in fr)
    print(f"frame: {len(fr)} kernels, span {(fr[-1][2] - fr[0][1]) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
for (s, e), (s2, e2) in list(zip(frames[:-1], frames[1:]))[-2:]:
    between = rows[e + 1:s2]
    print(f"between frames: {(rows[s2][1] - rows[e][2]) / 1e3:.1f} us, {len(between)} kernels busy {sum(r[2] - r[1] for r in between) / 1e3:.1f} us:",
          [r[0].split('(')[0][-40:] for r in between])
