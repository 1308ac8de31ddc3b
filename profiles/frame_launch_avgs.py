#!/usr/bin/env python3
"""Per frame of a rocprofv3 --kernel-trace .db of bench.py: number and average duration of the lookup launches that did work.  bench.py times the roofline kernel
in a pass of its own (HIP events carried by the launches); this shows whether those launches run like the ones of the timed steps.
usage: frame_launch_avgs.py results.db [first_timed_frame n_timed_frames]   (bench.py defaults: 5 20 -- frames 0..4 are the warmup steps)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
frames, cur = [], None
for name, s, e in rows:
    if "k_frame_begin" in name:
        cur = []
        frames.append(cur)
    elif cur is not None and "k_frame_grid" in name:
        cur.append((e - s) / 1e3)
for i, f in enumerate(frames):
    w = [x for x in f if x > 15.0]
    if w:
        print(f"frame {i:3d}: {len(w):3d} working lookup launches, avg {sum(w) / len(w):7.2f} us, sum {sum(w) / 1e3:6.3f} ms")

if len(sys.argv) > 3:
    a, n = int(sys.argv[2]), int(sys.argv[3])
    w = [x for f in frames[a:a + n] for x in f if x > 15.0]
    print(f"# frames {a}..{a + n - 1} (bench.py's timed steps): {len(w)} working lookup launches, avg {sum(w) / len(w):.2f} us (poses differ: 21 ... 29 iterations per frame, two poses with ~90 us launches)")
    f0 = [x for x in frames[a] if x > 15.0]
    print(f"# frame {a} (the FIRST timed step: the only one whose lookup launches carry bench.py's HIP events): {len(f0)} working lookup launches, avg {sum(f0) / len(f0):.2f} us by rocprof's dispatch "
          "time stamps; the events on the same launches read ~6 us more per launch (the completion signal and its release fence are part of an instrumented launch): the bench line's roofline is the conservative one")
    allw = [x for f in frames for x in f if x > 15.0]
    print(f"# all {len(frames)} frames of the process (warmup, timed steps, the legs behind them: other poses of the camera path): {len(allw)} working lookup launches, avg {sum(allw) / len(allw):.2f} us")
