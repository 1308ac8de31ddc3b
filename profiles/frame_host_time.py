#!/usr/bin/env python3
"""Host time of one native frame outside the library call: render() entry -> pnr_*_render_frame entry (the Python prologue), the call itself (launches +
the wait for the device), its return -> render() return (the Python epilogue), and the caller's loop between two render() calls.  perf_counter stamps
around a wrapper of the library entry point; no profiler (cProfile triples these figures).
usage: python profiles/frame_host_time.py [--workload lego|lego_palette|garden] [--shards 8] [--frames 40]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import _lib  # noqa: E402
from palettenerf_amd import dist as pdist  # noqa: E402
from palettenerf_amd.fused import tile_ray_order  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden")
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--frames", type=int, default=40)
    a = ap.parse_args()
    args = bench.parse(["--workload", a.workload, "--no-cpu-baseline", "--static-pose"])
    dev = torch.device("cuda", 0)
    m = bench.build_model(args, dev)
    H, W = args.wl["H"], args.wl["W"]
    idx, _ = pdist.shard_indices(H, W, 0, a.shards)
    bank = bench.RayBank(args, 1, idx, dev)
    m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
    kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    if args.model == "palette":
        kw["gui_mode"] = False
    lib = _lib.load()
    name = "pnr_palette_render_frame" if args.model == "palette" else "pnr_nerf_render_frame"
    real = getattr(lib, name)
    stamps = []

    def wrapped(*xs):
        t0 = time.perf_counter()
        rc = real(*xs)
        stamps.append((t0, time.perf_counter()))
        return rc
    setattr(lib, name, wrapped)
    rows = []
    with torch.no_grad():
        for i in range(5):
            m.render(*bank.get(i), **kw)
        torch.cuda.synchronize()
        stamps.clear()
        prev_exit = None
        for i in range(a.frames):
            t_in = time.perf_counter()
            m.render(*bank.get(i), **kw)
            t_out = time.perf_counter()
            c_in, c_out = stamps[-1]
            rows.append((t_in - prev_exit if prev_exit else 0.0, c_in - t_in, c_out - c_in, t_out - c_out))
            prev_exit = t_out
        torch.cuda.synchronize()
    med = lambda k: sorted(r[k] for r in rows[1:])[len(rows) // 2] * 1e6
    print(f"{a.workload} 1/{a.shards}: {idx.numel()} rays, {a.frames} frames; medians in us: loop between render() calls {med(0):.1f} | prologue (render -> library call) {med(1):.1f} | "
          f"library call {med(2):.1f} | epilogue (call returns -> render returns) {med(3):.1f}")


if __name__ == "__main__":
    main()
