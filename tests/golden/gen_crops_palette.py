#!/usr/bin/env python3
"""tests/golden/crop400_palette.npz and tests/golden/crop256_garden.npz: the CPU oracle's render of a centre crop of pose 0 of the two
PaletteNeRF workloads -- configs[2] (800x800, scene S0, -m palette, dt_gamma 0; 400x400 crop) and configs[4] (1297x840 garden frame, scene S2,
-m palette, dt_gamma 1/128; 256x256 crop) -- next to gen_crop400.py's NeRF digest (full-size PaletteNeRF images were
compared native-vs-compat only, HIP against HIP).  Same recipe: the oracle = this repository's mirror of palette/renderer.py run_cuda over the C
restatement of the kernels + torch CPU nn.Linear stacks; no reference checkout needed.  Stored: the rendered-sample count and the number of
hitting rays (integers, exact on any host) and 8x8 block means (float32) of image, weights_sum, view_dep_rgb, basis_acc and basis_rgb -- a digest a
host whose BLAS rounds the MLPs differently still reproduces to ~1e-7.  ~2 minutes on 8 cores for both."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from palettenerf_amd import scene  # noqa: E402

CASES = {
    "crop400_palette": dict(argv=["--workload", "lego_palette", "--no-extras"], crop=400),
    "crop256_garden": dict(argv=["--workload", "garden", "--no-extras"], crop=256),
}
MAPS = ("image", "view_dep_rgb", "basis_rgb", "basis_acc")     # + weights_sum: what the digest holds as block means


def case_args(name):
    return bench.parse(CASES[name]["argv"])


def crop_index(name):
    a = case_args(name)
    H, W, c = a.wl["H"], a.wl["W"], CASES[name]["crop"]
    y0, x0 = (H - c) // 2, (W - c) // 2
    return (torch.arange(y0, y0 + c)[:, None] * W + torch.arange(x0, x0 + c)[None, :]).reshape(-1)


def crop_rays(name):
    a = case_args(name)
    H, W = a.wl["H"], a.wl["W"]
    pose = torch.from_numpy(bench.pose_of(a, 0))[None]
    ro, rd = scene.get_rays(pose, bench.intrinsics_of(a), H, W)
    idx = crop_index(name)
    return ro[:, idx].contiguous(), rd[:, idx].contiguous()


def block_means(img, c, b=8):
    img = np.asarray(img, dtype=np.float64).reshape(c, c, -1)
    return img.reshape(c // b, b, c // b, b, img.shape[-1]).mean(axis=(1, 3)).astype(np.float32)


def digest(r, c):
    """{name: block means} of a render's maps (torch tensors or arrays, [1, c*c, k] / [c*c])."""
    host = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    d = {k + "_block_means": block_means(host(r[k]).reshape(c * c, -1), c) for k in MAPS}
    d["alpha_block_means"] = block_means(host(r["weights_sum"]).reshape(c * c, 1), c)[..., 0]
    return d


def oracle_crop(name, threads=8):
    """The crop's render dict by the oracle (the code path of bench.cpu_baseline's all-cores leg, PaletteNetwork)."""
    import oracle
    from oracle.facade import make_oracle_modules
    from palettenerf_amd import renderer
    import palettenerf_amd.gridencoder as pge
    import palettenerf_amd.shencoder as psh
    args = case_args(name)
    rm, ge, sh, _ = make_oracle_modules()
    saved = (renderer.raymarching, pge.GridEncoder, psh.SHEncoder)
    renderer.raymarching, pge.GridEncoder, psh.SHEncoder = rm, ge.GridEncoder, sh.SHEncoder
    prev = orc.use_variant("omp")
    orc.set_threads(threads)
    torch.set_num_threads(threads)
    try:
        m = bench.make_model(args, "palette")
        scene.seed_field_(m, 0)
        grid = bench.density_grid_of(args.wl["scene"])
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(oracle.packbits(grid, 0.5)))
        m.eval()
        m.count_rendered = True
        ro, rd = crop_rays(name)
        with torch.no_grad():
            r = m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    finally:
        orc.use_variant(prev)
        renderer.raymarching, pge.GridEncoder, psh.SHEncoder = saved
        torch.set_num_threads(1)
    return r


if __name__ == "__main__":
    import time
    for name, case in CASES.items():
        t0 = time.time()
        r = oracle_crop(name)
        c = case["crop"]
        d = digest(r, c)
        ws = r["weights_sum"].numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rendered=int(r["rendered"].item()), hit_rays=int((ws > 0).sum()),
                            image_mean=np.float64(r["image"].numpy().astype(np.float64).mean()), **d)
        print(name, int(r["rendered"].item()), int((ws > 0).sum()), float(r["image"].mean()), f"{time.time() - t0:.1f} s")
