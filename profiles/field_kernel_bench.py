#!/usr/bin/env python3
"""Microbenchmark of the fused field kernels alone (no march, no lookup): us per launch of pnr_nerf_field_forward / pnr_palette_field_forward on B rows of
seeded encoder features, HIP events on the launch stream.  usage: field_kernel_bench.py [--rows 1089480] [--reps 30]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from palettenerf_amd import _lib, network, renderer, scene  # noqa: E402
from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[335180, 1089480])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--num-basis", type=int, default=4)
    ap.add_argument("--pred-clip", action="store_true")
    ap.add_argument("--prec", choices=["both", "f16x3", "fp32"], default="both")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    for B in args.rows:
        enc = [((torch.rand(16, B, 2, generator=g) - 0.5) * 0.4).to(dev) for _ in range(3)]
        d = torch.randn(B, 3, generator=g)
        d = (d / d.norm(dim=1, keepdim=True)).to(dev)
        sig, rgb = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
        n = network.NeRFNetwork(bound=2, cuda_ray=True)
        scene.seed_field_(n, 0)
        n = n.to(dev).eval()
        precs = {"both": (1, 0), "f16x3": (1,), "fp32": (0,)}[args.prec]
        for prec in precs:
            f = NeRFFieldFused(n)
            f.precision = prec
            blob = f._pack(prec)
            fn = lambda: lib.pnr_nerf_field_forward(enc[0].data_ptr(), d.data_ptr(), blob.data_ptr(), B, sig.data_ptr(), rgb.data_ptr(), prec, ctypes.c_float(1.0), stream)
            med, best = timeit(fn, args.reps)
            print(f"nerf field    B={B:8d} prec={'f16x3' if prec else 'fp32 '}: median {med:7.1f} us  best {best:7.1f} us  ({B / med:7.0f} samples/us)")
        opt = renderer.default_opt(num_basis=args.num_basis, pred_clip=args.pred_clip)
        p = network.PaletteNetwork(opt, bound=2, cuda_ray=True)
        scene.seed_field_(p, 0)
        p = p.to(dev).eval()
        for prec in precs:
            f = PaletteFieldFused(p)
            f.precision = prec
            aux = torch.empty(B, f.aux_channels, device=dev)
            a = _lib.PaletteFieldArgs()
            a.ctl, a.B, a.level_stride = None, B, B
            a.enc, a.enc_palette, a.enc_clip = enc[0].data_ptr(), enc[1].data_ptr(), enc[2].data_ptr() if args.pred_clip else None
            a.dirs, a.deltas, a.packed = d.data_ptr(), None, f._pack(prec).data_ptr()
            a.num_basis, a.clip_dim, a.pred_clip = f.nb, f.clip_dim, int(f.pred_clip)
            a.density_scale, a.offsets_weight, a.view_dep_weight, a.aux_stride = 1.0, 1.0, 1.0, f.aux_channels
            a.sigmas, a.rgbs, a.aux, a.precision = sig.data_ptr(), rgb.data_ptr(), aux.data_ptr(), prec
            fn = lambda: lib.pnr_palette_field_forward(ctypes.byref(a), stream)
            med, best = timeit(fn, args.reps)
            print(f"palette field B={B:8d} prec={'f16x3' if prec else 'fp32 '}: median {med:7.1f} us  best {best:7.1f} us  ({B / med:7.0f} samples/us)  nb={f.nb} clip={f.clip_dim}")


if __name__ == "__main__":
    main()
