#!/usr/bin/env python3
"""How many samples of a training step get an all-zero gradient from the two train composites (rows behind the sample at which a ray's
transmittance fell below T_thresh: composite_rays_train stops there, raymarching.cu:660-672, 736-743)?  Those rows cost a full share of
every backward kernel and contribute exact zeros.  Synthetic configs[3] step (bench.make_training_step) and the trained scene of
profiles/train_palette.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import raymarching  # noqa: E402


def probe(tag, m, step):
    """step(i) runs one training step of model m (through palettenerf_amd.renderer's composites)."""
    seen = {}
    orig_t, orig_f = raymarching.composite_rays_train, raymarching.composite_rays_flex_train
    import palettenerf_amd.renderer as R

    def wrap_t(sigmas, rgbs, deltas, rays, T):
        if rgbs.requires_grad:
            rgbs.register_hook(lambda g: seen.__setitem__("rgbs", (int((g.abs().amax(dim=1) == 0).sum()), g.shape[0])))
        return orig_t(sigmas, rgbs, deltas, rays, T)

    def wrap_f(sigmas, buf, deltas, rays, T):
        if buf.requires_grad:
            buf.register_hook(lambda g: seen.__setitem__("all_buffer", (int((g.abs().amax(dim=1) == 0).sum()), g.shape[0])))
        return orig_f(sigmas, buf, deltas, rays, T)
    R.raymarching.composite_rays_train, R.raymarching.composite_rays_flex_train = wrap_t, wrap_f
    try:
        step(0)
        step(1)
    finally:
        R.raymarching.composite_rays_train, R.raymarching.composite_rays_flex_train = orig_t, orig_f
    for k, (z, n) in seen.items():
        print(f"{tag}: {k}: {z} of {n} rows have an all-zero gradient ({100.0 * z / n:.1f} %)")


def main():
    dev = torch.device("cuda:0")
    for kind in ("palette", "nerf"):
        m, step = bench.make_training_step(kind, 4096, dev)
        probe(f"bench step ({kind})", m, step)


if __name__ == "__main__":
    main()
