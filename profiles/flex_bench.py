"""composite_rays_flex at the sizes of a PaletteNeRF 800x800 frame's march iterations (palette/renderer.py:508-516: maps of 3, 3, nb, 3 nb, 3 nb channels + clip_dim):
the six single launches on the one-thread-per-ray kernel (pnr_set_option flex_coop 0), the six on the workgroup-cooperative kernel, and ONE pnr_composite_rays_flex_multi
launch.  HIP events around each group, median of 30."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from palettenerf_amd import _lib, raymarching

dev = torch.device("cuda")
lib = _lib.load()
chans = [3, 3, 4, 12, 12, 16]


def case(N, n_alive, n_step):
    g = torch.Generator().manual_seed(0)
    alive = torch.sort(torch.randperm(N, generator=g)[:n_alive]).values.to(torch.int32).to(dev)
    M = n_alive * n_step
    sig = (torch.rand(M, generator=g) * 60).to(dev)
    dl = (torch.rand(M, 2, generator=g) * 0.02 + 0.003).to(dev)
    ws = (torch.rand(N, generator=g) * 0.7).to(dev)
    t = torch.zeros(N, device=dev)
    ins = [torch.randn(M, c, generator=g).to(dev) for c in chans]
    outs = [torch.zeros(N, c, device=dev) for c in chans]

    def singles():
        for c, i, o in zip(chans, ins, outs):
            raymarching.composite_rays_flex(n_alive, n_step, c, alive, t, sig, i, dl, ws, o, 1e-4)

    def multi():
        raymarching.composite_rays_flex_multi(n_alive, n_step, alive, t, sig, dl, ws, list(zip(chans, ins, outs)), 1e-4)

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2] * 1e3

    lib.pnr_set_option(b"flex_coop", 0)
    a = timed(singles)
    lib.pnr_set_option(b"flex_coop", 1)
    b = timed(singles)
    c = timed(multi)
    nbytes = M * sum(chans) * 4 + 2 * n_alive * sum(chans) * 4
    print(f"N {N:7d} n_alive {n_alive:7d} n_step {n_step}: six singles per-ray {a:7.1f} us | six singles cooperative {b:7.1f} us | one multi {c:7.1f} us "
          f"({nbytes / c / 1e6:.2f} TB/s of input + read-modify-write bytes)")


for N, n_alive, n_step in ((640000, 640000, 1), (640000, 300000, 2), (640000, 100000, 6), (640000, 20000, 8), (640000, 2000, 8)):
    case(N, n_alive, n_step)
