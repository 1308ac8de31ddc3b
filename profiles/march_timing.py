#!/usr/bin/env python3
"""Per-wave time stamps inside ONE march launch of the native frame loop (needs a -DPNR_MARCH_TIMING build:
PNR_EXTRA_HIPCC_FLAGS=-DPNR_MARCH_TIMING python -m palettenerf_amd.build --force).  usage: march_timing.py [iteration ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import _lib, scene  # noqa: E402

order_kind = "rowmajor"
pose_step = 0
argv = list(sys.argv[1:])
while argv and argv[0].startswith("--"):
    if argv[0] == "--tile8":
        order_kind = "tile8"
        argv = argv[1:]
    elif argv[0] == "--pose":
        pose_step = int(argv[1])
        argv = argv[2:]
iters = [int(v) for v in argv] or [3]
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
m.march_mode = "native"
H = W = 800
pose = torch.from_numpy(bench.pose_of(args, pose_step))[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro.to(dev), rd.to(dev)
if order_kind == "tile8":
    from palettenerf_amd.fused import tile_ray_order
    m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    m._fused.ray_order = tile_ray_order(torch.arange(H * W), W, 8).to(dev)
print(f"ray order {order_kind}, pose step {pose_step}")
lib = _lib.load()
NW = 8192
buf = (ctypes.c_ulonglong * (NW * 8))()
with torch.no_grad():
    for _ in range(3):
        m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    for it in iters:
        lib.pnr_debug_march_timing(None, it)
        m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
        torch.cuda.synchronize()
        lib.pnr_debug_march_timing(buf, it)
        t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 8).astype(np.int64)
        live = t[:, 0] > 0
        wave_ids = np.nonzero(live)[0]
        t = t[live]
        crowd = t[:, 7] >> 16          # lanes of the wave with >= 5 probes (SIMT probes only: the cooperative tail is not counted)
        t[:, 7] &= 0xffff
        t0 = t[:, 0].min()
        marched = t[:, 5] > 0
        full = t[marched]
        crowd = crowd[marched]
        full_ids = wave_ids[marched]
        tick = 10.0   # ns per wall_clock64 tick (100 MHz)
        print(f"iteration {it}: {live.sum()} waves stamped, {len(full)} marched; launch span {(t[:, :6].max() - t0) * tick / 1e3:.1f} us")
        names = ["start", "after counts/schedule", "after mip staging", "after compaction", "after ctx/clip/skip", "after probes+stores"]
        for k in range(6):
            col = full[:, k] - t0
            print(f"  {names[k]:24s}: first {col.min() * tick / 1e3:6.2f}  median {np.median(col) * tick / 1e3:6.2f}  p90 {np.percentile(col, 90) * tick / 1e3:6.2f}  last {col.max() * tick / 1e3:6.2f} us")
        for k in range(1, 6):
            d = (full[:, k] - full[:, k - 1]) * tick / 1e3
            print(f"  phase {k} ({names[k]:24s}): median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
        order = np.argsort(full[:, 5])[::-1][:12]
        print("  slowest-ending waves (us since launch: start, counts, mip, compaction, ctx, end):")
        for w in order:
            print("   ", " ".join(f"{(full[w, k] - t0) * tick / 1e3:6.2f}" for k in range(6)), f"  max probes {full[w, 6]:3d}  lanes with >= 5 probes {crowd[w]:2d}")
        slow = np.argsort(full[:, 5])[::-1][:200]
        print(f"  the 200 last waves: lanes with >= 5 probes: median {np.median(crowd[slow]):.0f}, <= 2 in {(crowd[slow] <= 2).sum()}, <= 8 in {(crowd[slow] <= 8).sum()}, >= 32 in {(crowd[slow] >= 32).sum()}; max probes median {np.median(full[slow, 6]):.0f}")
        late = full[(full[:, 0] - t0) * tick / 1e3 > 1.0]
        print(f"  waves starting later than 1 us: {len(late)}; their probe phase: median {np.median((late[:, 5] - late[:, 4])) * tick / 1e3 if len(late) else 0:.2f} us")
        hist, edges = np.histogram((full[:, 5] - t0) * tick / 1e3, bins=12)
        print("  end-time histogram (us):", " ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}:{hist[i]}" for i in range(len(hist))))
        print("  probe phase by chunk index (block = chunk; 4 waves each): chunk range: median / max us, median start")
        blk = full_ids // 4
        for lo in range(0, int(blk.max()) + 1, 128):
            sel = (blk >= lo) & (blk < lo + 128)
            if sel.any():
                d = (full[sel, 5] - full[sel, 4]) * tick / 1e3
                print(f"    {lo:5d}-{lo + 127:5d}: {np.median(d):6.2f} / {d.max():6.2f}   start {np.median(full[sel, 0] - t0) * tick / 1e3:6.2f}")
        d = (full[:, 5] - full[:, 4]) * tick / 1e3
        print("  probe phase vs wave-max probes / brick loads:")
        for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 6), (7, 9), (10, 99)):
            sel = (full[:, 6] >= lo) & (full[:, 6] <= hi)
            if sel.any():
                print(f"    probes {lo}-{hi}: {sel.sum():5d} waves, phase median {np.median(d[sel]):6.2f} max {d[sel].max():6.2f} us, brick loads median {np.median(full[sel, 7]):.0f} max {full[sel, 7].max()}")
        A = np.stack([full[:, 6], full[:, 7], np.ones(len(full))], 1).astype(np.float64)
        coef, *_ = np.linalg.lstsq(A, d, rcond=None)
        print(f"    least squares: phase ~ {coef[0]:.2f} us x probes + {coef[1]:.2f} us x brick loads + {coef[2]:.2f} us")
