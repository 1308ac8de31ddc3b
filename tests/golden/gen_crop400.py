#!/usr/bin/env python3
"""tests/golden/crop400.npz: the CPU oracle's render of the centre 400x400 crop of pose 0 of the headline frame (configs[1]: 800x800, scene S0,
-m nerf, density_scale 100, dt_gamma 0) -- the comparison bench.py's `parity` block makes, committed so that a `-m gpu` test holds the HIP path
to it.  Needs no reference checkout: the oracle (C ops + torch CPU MLPs under this repository's mirror of run_cuda) and the seeded field are
all in the repository.  Stored: the rendered-sample count and per-ray sample counts' CRC (integers: exact on any host), and the image as 8x8
block means in float32 (50 x 50 x 3; a digest that a host whose BLAS rounds the MLPs differently still reproduces to ~1e-7 -- a bit hash of
the pixels would not be).  ~25 s on 8 cores."""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from palettenerf_amd import scene  # noqa: E402


def crop_rays(H=800, W=800, c=400):
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    idx = bench.crop_indices(H, W, c)
    return ro[:, idx].contiguous(), rd[:, idx].contiguous()


def oracle_crop(threads=8):
    """(image [160000,3], weights_sum, rendered count) of the crop by the oracle; same code path as bench.cpu_baseline's all-cores leg."""
    import oracle
    from oracle.facade import make_oracle_modules
    from palettenerf_amd import renderer
    import palettenerf_amd.gridencoder as pge
    import palettenerf_amd.shencoder as psh
    args = bench.parse(["--no-extras"])
    rm, ge, sh, _ = make_oracle_modules()
    saved = (renderer.raymarching, pge.GridEncoder, psh.SHEncoder)
    renderer.raymarching, pge.GridEncoder, psh.SHEncoder = rm, ge.GridEncoder, sh.SHEncoder
    prev = orc.use_variant("omp")
    orc.set_threads(threads)
    torch.set_num_threads(threads)
    try:
        m = bench.make_model(args, "nerf")
        scene.seed_field_(m, 0)
        grid = bench.density_grid_of("s0")
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(oracle.packbits(grid, 0.5)))
        m.eval()
        m.count_rendered = True
        ro, rd = crop_rays()
        with torch.no_grad():
            r = m.render(ro, rd, perturb=False, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4)
    finally:
        orc.use_variant(prev)
        renderer.raymarching, pge.GridEncoder, psh.SHEncoder = saved
        torch.set_num_threads(1)
    return r["image"][0].numpy(), r["weights_sum"].numpy(), int(r["rendered"].item())


def block_means(img, c=400, b=8):
    return img.reshape(c // b, b, c // b, b, 3).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)


if __name__ == "__main__":
    img, ws, n = oracle_crop()
    np.savez_compressed(os.path.join(HERE, "crop400.npz"), image_block_means=block_means(img), alpha_block_means=block_means(np.repeat(ws[:, None], 3, 1))[..., 0],
                        rendered=n, hit_rays=int((ws > 0).sum()), image_mean=np.float64(img.astype(np.float64).mean()))
    print("crop400", n, int((ws > 0).sum()), float(img.mean()))
