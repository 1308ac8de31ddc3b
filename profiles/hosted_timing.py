#!/usr/bin/env python3
"""Time stamps inside ONE lookup launch of the native frame loop with the hosted march tail (needs a -DPNR_HOSTED_TIMING build:
PNR_EXTRA_HIPCC_FLAGS=-DPNR_HOSTED_TIMING python -m palettenerf_amd.build --force).  usage: hosted_timing.py [--workload W] [iteration ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import _lib, scene  # noqa: E402

argv = list(sys.argv[1:])
wl = "lego"
pose_step = 5
while argv and argv[0].startswith("--"):
    if argv[0] == "--workload":
        wl = argv[1]
    elif argv[0] == "--pose":
        pose_step = int(argv[1])
    argv = argv[2:]
iters = [int(v) for v in argv] or [3, 10, 20]
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--workload", wl]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
m.march_mode = "native"
H, W = args.wl["H"], args.wl["W"]
pose = torch.from_numpy(bench.pose_of(args, pose_step))[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro.to(dev), rd.to(dev)
lib = _lib.load()
NB = 256
NM = 16 * 128
buf = (ctypes.c_ulonglong * (8 + 8 * NB + 2 * NM))()
kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
with torch.no_grad():
    for _ in range(3):
        m.render(ro, rd, **kw)
    for it in iters:
        lib.pnr_debug_hosted_timing(None, it)
        m.render(ro, rd, **kw)
        torch.cuda.synchronize()
        lib.pnr_debug_hosted_timing(buf, it)
        t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
        g, b, mt = t[:8], t[8:8 + 8 * NB].reshape(NB, 8), t[8 + 8 * NB:].reshape(NM, 2)
        mt = mt[mt[:, 0] > 0]
        live = b[:, 0] > 0
        b = b[live]
        if not len(mt):
            print(f"iteration {it}: no lookup work")
            continue
        if not len(b):
            print(f"iteration {it}: no hosted workgroup had work (queued {g[3]}); ordinary workgroups (every 16th): last end {(mt[:, 1].max() - mt[:, 0].min()) / 100.0:.1f} us")
            continue
        t0 = min(mt[:, 0].min(), b[:, 0].min())
        us = lambda v: (v - t0) / 100.0
        print(f"iteration {it}: rays queued {g[3]}, hosted workgroups with work {len(b)}; ordinary workgroups (every 16th): "
              f"last end {us(mt[:, 1].max()):.1f} us; hosted: march done {us(b[:, 2].max()):.1f}, last end {us(b[:, 3].max()):.1f} us, probes of the slowest lane: median {np.median(b[:, 4]):.0f} max {b[:, 4].max()}")
        if os.environ.get("HOSTED_TIMING_BRIEF"):
            continue
        for name, col in (("start", 0), ("mip staged", 1), ("march done", 2), ("lookups done", 3)):
            v = us(b[:, col])
            print(f"    hosted {name:13s} min {v.min():7.1f}  median {np.median(v):7.1f}  p95 {np.percentile(v, 95):7.1f}  max {v.max():7.1f} us")
        print(f"    probes of the slowest lane per workgroup: median {np.median(b[:, 4]):.0f}, max {b[:, 4].max()};  march us per probe (slowest workgroups): "
              f"{np.mean(((b[:, 2] - b[:, 1]) / 100.0 / np.maximum(b[:, 4], 1))[b[:, 4] >= 5]):.2f}")
