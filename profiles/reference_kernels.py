#!/usr/bin/env python3
"""The REFERENCE's own kernels on this MI355X next to this repository's: raymarching.cu, shencoder.cu and palette.cu of /root/reference, compiled
unmodified for gfx950 by torch's CUDA-extension toolchain (hipify + hipcc; oracle/ref_build.py: build_hip -> oracle/_ref/ref_*.so), called through
their own pybind modules with the argument lists of raymarching.h / shencoder.h / palette_func.h.  For every kernel: largest difference and the
fraction of bit-identical outputs against this repository's HIP kernel on the same inputs, and microseconds per call for both.
Usage: reference_kernels.py [--json]   (needs oracle/_ref/ref_*.so; nothing here reads /root/reference)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from oracle import ref_build
from palettenerf_amd import palette_utils, raymarching, scene, shencoder

dev = torch.device("cuda:0")
F32, I32 = torch.float32, torch.int32


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cmp(a, b):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    fin = np.isfinite(a) & np.isfinite(b)
    same_nonfinite = bool(np.array_equal(np.isfinite(a), np.isfinite(b)))
    d = float(np.abs(a[fin].astype(np.float64) - b[fin].astype(np.float64)).max()) if fin.any() else 0.0
    scale = float(np.abs(b[fin]).max()) if fin.any() else 1.0
    return {"max_abs": d, "max_rel_to_largest": d / max(scale, 1e-30), "bit_identical_frac": float((a.view(np.uint8) == b.view(np.uint8)).reshape(a.shape + (-1,)).all(-1).mean()) if a.dtype != np.bool_ else float((a == b).mean()),
            "nonfinite_pattern_equal": same_nonfinite}


def main():
    rm, sh, pal = ref_build.load_hip("raymarching"), ref_build.load_hip("shencoder"), ref_build.load_hip("palette")
    out = {}
    g = torch.Generator().manual_seed(0)
    # ---------------------------------------------------------------- Morton / packbits / near-far
    N = 1 << 20
    coords = torch.randint(0, 128, (N, 3), generator=g, dtype=I32).to(dev)
    ind_ref = torch.empty(N, dtype=I32, device=dev)
    rm.morton3D(coords, N, ind_ref)
    ind = raymarching.morton3D(coords)
    back_ref = torch.empty(N, 3, dtype=I32, device=dev)
    rm.morton3D_invert(ind_ref, N, back_ref)
    out["morton3D"] = {**cmp(ind, ind_ref), "ref_us": timed(lambda: rm.morton3D(coords, N, ind_ref)), "ours_us": timed(lambda: raymarching.morton3D(coords))}
    out["morton3D_invert"] = {**cmp(raymarching.morton3D_invert(ind), back_ref), "roundtrip": bool(torch.equal(back_ref, coords))}
    grid = torch.from_numpy(scene.brick_density_grid()).to(dev)
    grid_n = (grid + torch.rand(grid.shape, generator=g).to(dev) * 0.2 - 0.1).contiguous()
    bits_ref = torch.empty(grid_n.numel() // 8, dtype=torch.uint8, device=dev)
    rm.packbits(grid_n, bits_ref.numel(), 0.5, bits_ref)
    bits = raymarching.packbits(grid_n, 0.5)
    out["packbits"] = {**cmp(bits, bits_ref), "ref_us": timed(lambda: rm.packbits(grid_n, bits_ref.numel(), 0.5, bits_ref)), "ours_us": timed(lambda: raymarching.packbits(grid_n, 0.5, bits))}
    H = W = 800
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro[0].contiguous().to(dev), rd[0].contiguous().to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=F32, device=dev)
    n_ref, f_ref = torch.empty(H * W, device=dev), torch.empty(H * W, device=dev)
    rm.near_far_from_aabb(ro, rd, aabb, H * W, 0.2, n_ref, f_ref)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    out["near_far_from_aabb"] = {"nears": cmp(nears, n_ref), "fars": cmp(fars, f_ref), "ref_us": timed(lambda: rm.near_far_from_aabb(ro, rd, aabb, H * W, 0.2, n_ref, f_ref)),
                                 "ours_us": timed(lambda: raymarching.near_far_from_aabb(ro, rd, aabb, 0.2))}
    # like for like with ref_us (a bare binding call on preallocated outputs; `ours_us` above is the whole operator: autograd.Function, three .contiguous(),
    # two allocations): the C-ABI entry through ctypes with plain ints
    from palettenerf_amd import _lib as _plib
    from palettenerf_amd._torch_glue import stream_ptr
    n2, f2 = torch.empty_like(n_ref), torch.empty_like(f_ref)
    p_ro, p_rd, p_ab, p_n, p_f = ro.data_ptr(), rd.data_ptr(), aabb.data_ptr(), n2.data_ptr(), f2.data_ptr()
    out["near_far_from_aabb"]["ours_bare_call_us"] = timed(lambda: _plib.call("pnr_near_far_from_aabb", p_ro, p_rd, p_ab, H * W, 0.2, p_n, p_f, stream_ptr()))
    # ---------------------------------------------------------------- march_rays_train (a 200 x 200 window of the frame: 40 000 rays)
    bitfield = raymarching.packbits(grid, 0.5)
    rows = torch.arange(300, 500)[:, None] * W + torch.arange(300, 500)[None, :]
    sel = rows.reshape(-1).to(dev)
    tro, trd, tn, tf = ro[sel].contiguous(), rd[sel].contiguous(), nears[sel].contiguous(), fars[sel].contiguous()
    NT = tro.shape[0]
    for dt_gamma in (0.0, 1.0 / 128):
        M = NT * 1024
        x_r, d_r, dl_r = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays_r, cnt_r, noises = torch.empty(NT, 3, dtype=I32, device=dev), torch.zeros(2, dtype=I32, device=dev), torch.zeros(NT, device=dev)
        rm.march_rays_train(tro, trd, bitfield, 2.0, dt_gamma, 1024, NT, 2, 128, M, tn, tf, x_r, d_r, dl_r, rays_r, cnt_r, noises)
        cnt = torch.zeros(2, dtype=I32, device=dev)
        x, d, dl, rays = raymarching.march_rays_train(tro, trd, 2.0, bitfield, 2, 128, tn, tf, cnt, -1, False, 128, True, dt_gamma, 1024)
        torch.cuda.synchronize()
        train_dl, train_rays, train_total = dl, rays, int(cnt[0])      # (the last pass, dt_gamma 1/128, feeds the training composites below)
        # the reference's row order is atomics-ordered: bring its rows into ray order
        order = torch.argsort(rays_r[:, 0].long())
        rr = rays_r[order]
        counts_equal = bool(torch.equal(rr[:, 2], rays[:, 2])) and bool(torch.equal(rr[:, 0], rays[:, 0]))
        total = int(cnt_r[0])
        idx_ref = torch.repeat_interleave(rr[:, 1].long(), rr[:, 2].long()) + (torch.arange(total, device=dev) - torch.repeat_interleave(torch.cumsum(rr[:, 2].long(), 0) - rr[:, 2].long(), rr[:, 2].long()))
        rec = {"counter_ref": cnt_r.tolist(), "counter_ours": cnt.tolist(), "per_ray_counts_equal": counts_equal, "samples": total}
        if counts_equal:
            rec["xyzs"] = cmp(x[:total], x_r[idx_ref])
            rec["dirs"] = cmp(d[:total], d_r[idx_ref])
            rec["deltas"] = cmp(dl[:total], dl_r[idx_ref])
        else:
            rec["rays_with_different_count"] = int((rr[:, 2] != rays[:, 2]).sum())
            rec["max_count_difference"] = int((rr[:, 2] - rays[:, 2]).abs().max())

        def ref_call():
            cnt_r.zero_()
            rm.march_rays_train(tro, trd, bitfield, 2.0, dt_gamma, 1024, NT, 2, 128, M, tn, tf, x_r, d_r, dl_r, rays_r, cnt_r, noises)

        def our_call():
            cnt.zero_()
            raymarching.march_rays_train(tro, trd, 2.0, bitfield, 2, 128, tn, tf, cnt, -1, False, 128, True, dt_gamma, 1024)
        rec["ref_us"], rec["ours_us_incl_wrapper"] = timed(ref_call, 5), timed(our_call, 5)
        out[f"march_rays_train_dt{dt_gamma:.4f}"] = rec
    # ---------------------------------------------------------------- march_rays (inference, first iteration of the window) + composite_rays
    for n_step in (1, 4):
        alive = torch.arange(NT, dtype=I32, device=dev)
        rays_t = tn.clone()
        Mi = NT * n_step + (128 - (NT * n_step) % 128)
        x_r, d_r, dl_r = torch.zeros(Mi, 3, device=dev), torch.zeros(Mi, 3, device=dev), torch.zeros(Mi, 2, device=dev)
        nz = torch.zeros(NT, device=dev)
        rm.march_rays(NT, n_step, alive, rays_t, tro, trd, 2.0, 0.0, 1024, 2, 128, bitfield, tn, tf, x_r, d_r, dl_r, nz)
        xi, di, dli = raymarching.march_rays(NT, n_step, alive, rays_t, tro, trd, 2.0, bitfield, 2, 128, tn, tf, 128, False, 0.0, 1024)
        rec = {"xyzs": cmp(xi, x_r), "dirs": cmp(di, d_r), "deltas": cmp(dli, dl_r), "rows": Mi,
               "ref_us": timed(lambda: rm.march_rays(NT, n_step, alive, rays_t, tro, trd, 2.0, 0.0, 1024, 2, 128, bitfield, tn, tf, x_r, d_r, dl_r, nz)),
               "ours_us_incl_wrapper": timed(lambda: raymarching.march_rays(NT, n_step, alive, rays_t, tro, trd, 2.0, bitfield, 2, 128, tn, tf, 128, False, 0.0, 1024))}
        out[f"march_rays_nstep{n_step}"] = rec
        sig = (torch.rand(Mi, generator=g) * 60).to(dev)
        rgb = torch.rand(Mi, 3, generator=g).to(dev)
        st = lambda: (alive.clone(), rays_t.clone(), torch.zeros(NT, device=dev), torch.zeros(NT, device=dev), torch.zeros(NT, 3, device=dev))
        a1, a2 = st(), st()
        rm.composite_rays(NT, n_step, 1e-4, a1[0], a1[1], sig, rgb, dl_r, a1[2], a1[3], a1[4])
        raymarching.composite_rays(NT, n_step, a2[0], a2[1], sig, rgb, dli, a2[2], a2[3], a2[4], 1e-4)
        out[f"composite_rays_nstep{n_step}"] = {"rays_alive": cmp(a2[0], a1[0]), "rays_t": cmp(a2[1], a1[1]), "weights_sum": cmp(a2[2], a1[2]), "depth": cmp(a2[3], a1[3]), "image": cmp(a2[4], a1[4])}
        inp = torch.rand(Mi, 50, generator=g).to(dev)
        o1, o2 = torch.zeros(NT, 50, device=dev), torch.zeros(NT, 50, device=dev)
        b1, b2 = st(), st()
        rm.composite_rays_flex(NT, n_step, 50, 1e-4, b1[0], b1[1], sig, inp, dl_r, b1[2], o1)
        raymarching.composite_rays_flex(NT, n_step, 50, b2[0], b2[1], sig, inp, dli, b2[2], o2, 1e-4)
        out[f"composite_rays_flex_nstep{n_step}"] = {"output": cmp(o2, o1)}
    # ---------------------------------------------------------------- training composites on the dt_gamma 1/128 march
    total, rays = train_total, train_rays
    sig = (torch.rand(total, generator=g) * 20).to(dev)
    rgb = torch.rand(total, 3, generator=g).to(dev)
    dlt = train_dl[:total].contiguous()
    assert dlt.shape[0] == total and int(rays[:, 2].sum()) == total
    ws_r, dp_r, im_r = torch.empty(NT, device=dev), torch.empty(NT, device=dev), torch.empty(NT, 3, device=dev)
    rm.composite_rays_train_forward(sig, rgb, dlt, rays, total, NT, 1e-4, ws_r, dp_r, im_r)
    ws, dp, im = raymarching.composite_rays_train(sig, rgb, dlt, rays, 1e-4)
    gws, gim = torch.rand(NT, generator=g).to(dev), torch.rand(NT, 3, generator=g).to(dev)
    gs_r, gc_r = torch.zeros_like(sig), torch.zeros_like(rgb)
    rm.composite_rays_train_backward(gws, gim, sig, rgb, dlt, rays, ws_r, im_r, total, NT, 1e-4, gs_r, gc_r)
    s2, c2 = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    w2, _, i2 = raymarching.composite_rays_train(s2, c2, dlt, rays, 1e-4)
    (w2 * gws).sum().backward(retain_graph=True)
    (i2 * gim).sum().backward()
    out["composite_rays_train"] = {"weights_sum": cmp(ws, ws_r), "depth": cmp(dp, dp_r), "image": cmp(im, im_r), "grad_sigmas": cmp(s2.grad, gs_r), "grad_rgbs": cmp(c2.grad, gc_r),
                                   "ref_fwd_us": timed(lambda: rm.composite_rays_train_forward(sig, rgb, dlt, rays, total, NT, 1e-4, ws_r, dp_r, im_r)),
                                   "ours_fwd_us_incl_wrapper": timed(lambda: raymarching.composite_rays_train(sig, rgb, dlt, rays, 1e-4))}
    inp = torch.rand(total, 33, generator=g).to(dev)
    of_r = torch.empty(NT, 33, device=dev)
    rm.composite_rays_flex_train_forward(sig, inp, dlt, rays, total, NT, 33, 1e-4, of_r)
    of = raymarching.composite_rays_flex_train(sig, inp, dlt, rays, 1e-4)
    go = torch.rand(NT, 33, generator=g).to(dev)
    gi_r = torch.zeros_like(inp)
    rm.composite_rays_flex_train_backward(go, sig, inp, dlt, rays, of_r, total, NT, 33, 1e-4, gi_r)
    i3 = inp.clone().requires_grad_(True)
    (raymarching.composite_rays_flex_train(sig, i3, dlt, rays, 1e-4) * go).sum().backward()
    out["composite_rays_flex_train"] = {"output": cmp(of, of_r), "grad_input": cmp(i3.grad, gi_r),
                                        "ref_fwd_us": timed(lambda: rm.composite_rays_flex_train_forward(sig, inp, dlt, rays, total, NT, 33, 1e-4, of_r)),
                                        "ours_fwd_us_incl_wrapper": timed(lambda: raymarching.composite_rays_flex_train(sig, inp, dlt, rays, 1e-4))}
    # ---------------------------------------------------------------- SH, HSV
    B = 1 << 20
    dd = torch.randn(B, 3, generator=g)
    dd = (dd / dd.norm(dim=1, keepdim=True)).to(dev)
    enc = shencoder.SHEncoder(degree=4).to(dev)
    y_r = torch.empty(B, 16, device=dev)
    sh.sh_encode_forward(dd, y_r, B, 3, 4, None)
    out["sh_encode_forward_deg4"] = {**cmp(enc(dd), y_r), "ref_us": timed(lambda: sh.sh_encode_forward(dd, y_r, B, 3, 4, None)), "ours_us_incl_wrapper": timed(lambda: enc(dd))}
    px = torch.rand(B, 3, generator=g).to(dev)
    h_r = torch.empty(B, 3, device=dev)
    pal.rgb_to_hsv(B, px, h_r)
    hsv = palette_utils.rgb_to_hsv(px)
    b_r = torch.empty(B, 3, device=dev)
    pal.hsv_to_rgb(B, h_r, b_r)
    out["rgb_to_hsv"] = {**cmp(hsv, h_r), "ref_us": timed(lambda: pal.rgb_to_hsv(B, px, h_r)), "ours_us_incl_wrapper": timed(lambda: palette_utils.rgb_to_hsv(px))}
    out["hsv_to_rgb"] = cmp(palette_utils.hsv_to_rgb(h_r), b_r)
    print(json.dumps(out, indent=None if "--json" in sys.argv else 1))


if __name__ == "__main__":
    main()
