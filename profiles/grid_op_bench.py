"""The stand-alone hash-grid lookup op (pnr_grid_encode_forward[_layout], what gridencoder.grid_encode calls under an unchanged run_cuda) against
the HBM roofline, at the condition SURVEY.md 8(d) states: B >= 2^20 ray-coherent samples, 16 levels, fp32 table, D = 3, C = 2.

Batches (all produced by this repo's march on the benchmark scenes; nothing is read from /root/reference):
  coherent   2^20 consecutive rows of march_rays_train over the 800x800 lego frame (scene S0, dt_gamma 0): samples of a ray are consecutive,
             rays in row-major pixel order -- the order an unchanged run_cuda / training step hands the encoder
  frame      2^20 first-iteration rows of the 8x8-tile-ordered frame (the coherence the device-driven frame loop sees), n_step = 2
  train      the configs[3] training batch: 4 096 random rays of the forward-facing slab scene, dt_gamma 1/128 (~ 626 k samples)
  random     2^20 uniform random points (no coherence at all: the floor)
For each: the generic kernel (grid_fast 0), the D3C2 kernel with [L,B,C] output, and with [B,L*C] rows; HIP events over 50 launches.
Algorithmic bytes per sample = 1 164.  `--once` runs every variant exactly once (for rocprofv3 --kernel-trace --stats).
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from palettenerf_amd import _lib, raymarching, scene
from palettenerf_amd._torch_glue import call, ptr
from palettenerf_amd.fused import tile_ray_order

_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int
ONCE = "--once" in sys.argv
BYTES = 1164.0
dev = torch.device("cuda", 0)
args = bench.parse(["--no-cpu-baseline"])
m = bench.build_model(args, dev)
enc = m.encoder
emb = enc.embeddings.detach()
H = W = 800
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro[0].contiguous().to(dev), rd[0].contiguous().to(dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
N = 1 << 20
batches = {}

# coherent: the rows march_rays_train emits for the centre rows of the frame
band = slice(H * W // 2 - 40 * W, H * W // 2 + 40 * W)     # 64 000 rays through the object: > 2^20 samples on S0
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, _, _, _ = raymarching.march_rays_train(ro[band].contiguous(), rd[band].contiguous(), m.bound, m.density_bitfield, m.cascade, m.grid_size, nears[band].contiguous(),
                                             fars[band].contiguous(), counter, -1, False, 128, True, 0.0, 1024)
assert int(counter[0]) >= N, int(counter[0])
batches["coherent"] = ((xyzs[:N] + m.bound) / (2 * m.bound)).contiguous()

# frame: first-iteration rows of the tile-ordered frame
order = tile_ray_order(torch.arange(H * W), W, 8).long().to(dev)
alive = torch.arange(H * W, dtype=torch.int32, device=dev)
x2, _, d2 = raymarching.march_rays(H * W, 2, alive, nears[order].clone(), ro[order].contiguous(), rd[order].contiguous(), m.bound, m.density_bitfield, m.cascade, m.grid_size,
                                   nears[order].contiguous(), fars[order].contiguous(), -1, False, 0.0, 1024)
keep = d2[:, 0] > 0
xf = ((x2[keep] + m.bound) / (2 * m.bound)).contiguous()
batches["frame"] = xf[:N] if xf.shape[0] >= N else xf

# train: configs[3]-shaped batch (bench.make_training_step's rig: camera 0 of the 17 on the 0.3-radius disc, slab scene, 4 096 random pixels)
tm, _ = bench.make_training_step("nerf", 4096, dev)
TH, TW = 756, 1008
p0 = np.eye(4, dtype=np.float32)
p0[:3, 0], p0[:3, 1], p0[:3, 2], p0[:3, 3] = [1, 0, 0], [0, -1, 0], [0, 0, -1], [0.3, 0.0, 1.5]
tro, trd = scene.get_rays(torch.from_numpy(p0[None]), scene.intrinsics_from_fov(TH, TW, 0.9), TH, TW)
inds = torch.randint(0, TH * TW, [4096], generator=torch.Generator().manual_seed(0))
tro, trd = tro[0, inds].contiguous().to(dev), trd[0, inds].contiguous().to(dev)
tn, tf = raymarching.near_far_from_aabb(tro, trd, tm.aabb_train, tm.min_near)
tc = torch.zeros(2, dtype=torch.int32, device=dev)
tx, _, _, _ = raymarching.march_rays_train(tro, trd, tm.bound, tm.density_bitfield, tm.cascade, tm.grid_size, tn, tf, tc, -1, False, 128, True, 1 / 128, 1024)
batches["train"] = ((tx[: int(tc[0])] + tm.bound) / (2 * tm.bound)).contiguous()
del tm

g = torch.Generator(device="cpu").manual_seed(0)
batches["random"] = torch.rand(N, 3, generator=g).to(dev)

lib = _lib.load()
L = enc.num_levels
S = float(np.log2(enc.per_level_scale))
results = {}
for name, x in batches.items():
    B = x.shape[0]
    out = torch.empty(B * L * 2, device=dev)
    res = {"B": B}

    def run(layout):
        call("pnr_grid_encode_forward_layout", ptr(x), ptr(emb), ptr(enc.offsets), ptr(out), _u32(B), _u32(3), _u32(2), _u32(L), _f32(S), _u32(enc.base_resolution), None,
             _u32(enc.gridtype_id), _int(int(enc.align_corners)), _int(0), _int(layout))

    for label, fast, layout, nt in (("generic_levels", 0, 0, 0), ("d3c2_levels", 1, 0, 0), ("d3c2_levels_nt_store", 1, 0, 1), ("d3c2_levels_nt_load", 1, 0, 2),
                                    ("d3c2_levels_nt_both", 1, 0, 3), ("d3c2_rows", 1, 1, 0)):
        lib.pnr_set_option(b"grid_fast", fast)
        lib.pnr_set_option(b"grid_nt", nt)
        if ONCE:
            run(layout)
            torch.cuda.synchronize()
            continue
        for _ in range(5):
            run(layout)
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            run(layout)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tbs = B * BYTES / (us * 1e-6) / 1e12
        res[label] = {"us": round(us, 2), "TB_per_s": round(tbs, 3), "frac_of_8TBs": round(tbs / 8.0, 3)}
    lib.pnr_set_option(b"grid_fast", 1)
    lib.pnr_set_option(b"grid_nt", 0)
    if not ONCE:   # what the permute-copy of the [L,B,C] form costs on top (gridencoder/grid.py:57)
        lm = out.view(L, B, 2)
        for _ in range(3):
            lm.permute(1, 0, 2).reshape(B, L * 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            lm.permute(1, 0, 2).reshape(B, L * 2)
        e1.record()
        torch.cuda.synchronize()
        res["permute_copy_us"] = round(e0.elapsed_time(e1) / 20 * 1e3, 2)
    results[name] = res
    print(name, json.dumps(res))
print(json.dumps({"grid_op_bench": results, "bytes_per_sample": BYTES, "peak_TB_per_s": 8.0}))
