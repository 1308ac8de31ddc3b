#!/usr/bin/env python3
"""Where a wave of the PaletteNeRF field kernel spends a tile (needs the -DPNR_PAL_TIMING variant:
    python -m palettenerf_amd.build --variant paltiming --only palette_field -- -DPNR_PAL_TIMING
    PNR_LIB_PATH=palettenerf_amd/libpnr_hip_paltiming.so python profiles/pal_timing.py [--workload garden|lego_palette] [--frames 3]).
NOTE (round 5, after the kernel went to 16 waves = 128 registers): the sixteen 64-bit phase sums of this build no longer fit -- it spills to scratch and runs
an order of magnitude slower (29 ms per shard frame against 2.1); build it as `-DPNR_PAL_TIMING -DPNR_PAL_WIDE_WAVES=8` (256 registers) for usable proportions.
Phases are stamped with the 100 MHz wall clock by every wave; a stamp behind a load phase first waits for the loads (s_waitcnt vmcnt(0)), so the *_wait
rows are exposed memory latency.  Sums over all waves and tiles of all field launches of the timed frames."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import _lib  # noqa: E402

NAMES = ["top (issue loads)", "enc wait", "sigma_net", "diff_net", "color_net", "enc_pal load+wait", "basis_net+heads", "epilogue (scalar, ds_write)",
         "leader weights", "aux_map RMW / write-back", "ray state + counts", "skipped tiles", "setup (weights->LDS)", "early_ws + aux rows -> LDS (issue)",
         "wait for the slab rows", "accumulate in LDS"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--shards", type=int, default=1, help="render shard 0 of this many tile shards (8: what one rank of an 8-GPU split renders)")
    a = ap.parse_args()
    args = bench.parse(["--workload", a.workload, "--no-cpu-baseline", "--no-extras"])
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    fn = lib.pnr_debug_pal_timing      # only in a -DPNR_PAL_TIMING build
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    m = bench.build_model(args, dev)
    from palettenerf_amd import dist as pdist
    from palettenerf_amd.fused import tile_ray_order
    H, W = args.wl["H"], args.wl["W"]
    idx, _ = pdist.shard_indices(H, W, 0, a.shards)
    m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
    bank = bench.RayBank(args, 1, idx, dev)
    kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4, gui_mode=False)
    with torch.no_grad():
        for i in range(3):
            m.render(*bank.get(i), **kw)
        torch.cuda.synchronize()
        fn(None, 1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        rendered = 0
        for i in range(a.frames):
            r = m.render(*bank.get(3 + i), **kw)
            rendered += int(r["rendered"].sum())
        ev1.record()
        torch.cuda.synchronize()
    n = len(NAMES) + 3
    buf = (ctypes.c_ulonglong * n)()
    fn(buf, 0)
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)
    phases, tiles, waves, resid = t[:len(NAMES)], t[len(NAMES)], t[len(NAMES) + 1], t[len(NAMES) + 2]
    tot = phases.sum()
    print(f"{a.workload} 1/{a.shards}: {a.frames} frames, {ev0.elapsed_time(ev1) / a.frames:.2f} ms/frame (timing build), {rendered / a.frames / 1e6:.2f} M samples/frame")
    print(f"wave tiles {tiles:.0f}, waves {waves:.0f}, wave residence {resid / 100 / max(waves, 1):.1f} us per wave, {tot / 100 / max(tiles, 1):.2f} us per tile")
    for nme, v in zip(NAMES, phases):
        print(f"  {nme:32s} {v / 100 / max(tiles, 1):8.3f} us/tile  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
