"""This repository's HIP kernels against the REFERENCE'S OWN KERNELS, both executed on the MI355X.

oracle/_ref/ref_{raymarching,shencoder,palette}.so are raymarching.cu, shencoder.cu and palette.cu of the reference, compiled unmodified for gfx950
by the image's toolchain for CUDA extensions on ROCm (torch.utils.cpp_extension: hipify + hipcc; oracle/ref_build.py: build_hip), called through
their own pybind modules with the argument lists of raymarching.h / shencoder.h / palette_func.h.  (gridencoder.cu does not compile for HIP --
one atomicAdd(__half2*) overload is missing from ROCm 7.2 -- and stays pinned through the reference's Python wrapper over the oracle.)

What holds (measured, profiles/r04_reference_kernels.json), and is asserted here:
  bit for bit   morton3D / invert, packbits, near_far_from_aabb, march_rays_train (counter, per-ray counts, every position / direction / delta of
                15.6 M samples), march_rays, composite_rays (alive list, t, weights, depth, image), composite_rays_flex, composite_rays_flex_train
                forward and backward
  to rounding   composite_rays_train (this repository's 16-lanes-per-ray scan adds in another order: <= 2e-6 of the largest value),
                SH (product form against the reference's expanded polynomials: <= 1e-6), HSV (<= 2e-7 relative)
The build container has no GPU, the GPU box no reference checkout: the .so files are built there and loaded here.  Missing files FAIL these
tests under `-m gpu` (PNR_ALLOW_NO_REF=1 turns that into a skip for trees built without the reference)."""
import numpy as np
import pytest
import torch

from palettenerf_amd import palette_utils, raymarching, scene, shencoder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(cuda):
    import os
    from oracle import ref_build, ref_ops
    if not ref_ops.available():
        # the suite's strongest check: on a GPU box its absence is a FAILURE (a lost build step must not leave the suite green), unless the caller
        # says explicitly that this tree was built without the reference checkout
        msg = "oracle/_ref/ref_*.so are not built (oracle/ref_build.py: build_hip needs the reference checkout at build time)"
        if os.environ.get("PNR_ALLOW_NO_REF") == "1":
            pytest.skip(msg + " -- PNR_ALLOW_NO_REF=1")
        pytest.fail(msg + "; set PNR_ALLOW_NO_REF=1 to run the GPU suite without the reference's kernels")
    return {"rm": ref_build.load_hip("raymarching"), "sh": ref_build.load_hip("shencoder"), "pal": ref_build.load_hip("palette")}


@pytest.fixture(scope="module")
def window(cuda):
    """A 160 x 160 window of the 800 x 800 configs[1] frame (25 600 rays through the object), scene S0."""
    H = W = 800
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    rows = (torch.arange(320, 480)[:, None] * W + torch.arange(320, 480)[None, :]).reshape(-1)
    ro, rd = ro[0][rows].contiguous().to(cuda), rd[0][rows].contiguous().to(cuda)
    grid = torch.from_numpy(scene.brick_density_grid()).to(cuda)
    bitfield = raymarching.packbits(grid, 0.5)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    return ro, rd, nears, fars, bitfield, grid, aabb


def same_bits(a, b, what=""):
    assert a.shape == b.shape and a.dtype == b.dtype, what
    assert torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)), what


def test_morton_packbits_near_far_bit_identical(cuda, ref, window):
    rm = ref["rm"]
    ro, rd, nears, fars, bitfield, grid, aabb = window
    g = torch.Generator().manual_seed(1)
    N = 300007
    coords = torch.randint(0, 128, (N, 3), generator=g, dtype=torch.int32).to(cuda)
    ind_ref = torch.empty(N, dtype=torch.int32, device=cuda)
    rm.morton3D(coords, N, ind_ref)
    same_bits(raymarching.morton3D(coords), ind_ref, "morton3D")
    back_ref = torch.empty(N, 3, dtype=torch.int32, device=cuda)
    rm.morton3D_invert(ind_ref, N, back_ref)
    same_bits(raymarching.morton3D_invert(ind_ref), back_ref, "morton3D_invert")
    noisy = (grid + torch.rand(grid.shape, generator=g).to(cuda) * 0.2 - 0.1).contiguous()
    noisy.view(-1)[:16] = torch.tensor([0.5, 0.5000001, 0.4999999, 0.0] * 4, device=cuda)     # strict `>` at the threshold
    bits_ref = torch.empty(noisy.numel() // 8, dtype=torch.uint8, device=cuda)
    rm.packbits(noisy, bits_ref.numel(), 0.5, bits_ref)
    same_bits(raymarching.packbits(noisy, 0.5), bits_ref, "packbits")
    n_ref, f_ref = torch.empty_like(nears), torch.empty_like(fars)
    rd2 = rd.clone()
    rd2[:7] = torch.tensor([[0, 0, 1.0], [1, 0, 0], [0, -1, 0], [0, 0.6, 0.8], [1e-30, 0, 1], [0, 0, -1], [0.6, 0.8, 0]], device=cuda)    # 1/0 and 0 * inf lanes
    rm.near_far_from_aabb(ro, rd2, aabb, ro.shape[0], 0.2, n_ref, f_ref)
    n, f = raymarching.near_far_from_aabb(ro, rd2, aabb, 0.2)
    same_bits(n, n_ref, "nears")
    same_bits(f, f_ref, "fars")


@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_rays_train_every_sample_bit_identical(cuda, ref, window, dt_gamma):
    rm = ref["rm"]
    ro, rd, nears, fars, bitfield, _, _ = window
    N = ro.shape[0]
    M = N * 1024
    x_r, d_r, dl_r = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda), torch.zeros(M, 2, device=cuda)
    rays_r, cnt_r, noises = torch.empty(N, 3, dtype=torch.int32, device=cuda), torch.zeros(2, dtype=torch.int32, device=cuda), torch.zeros(N, device=cuda)
    rm.march_rays_train(ro, rd, bitfield, 2.0, dt_gamma, 1024, N, 2, 128, M, nears, fars, x_r, d_r, dl_r, rays_r, cnt_r, noises)
    cnt = torch.zeros(2, dtype=torch.int32, device=cuda)
    x, d, dl, rays = raymarching.march_rays_train(ro, rd, 2.0, bitfield, 2, 128, nears, fars, cnt, -1, False, 128, True, dt_gamma, 1024)
    assert cnt.tolist() == cnt_r.tolist() and int(cnt[0]) > 500_000
    # the reference's rows are in atomics order (raymarching.cu:408-409), this repository's in ray order: same (ray, offset, count) mapping
    rr = rays_r[torch.argsort(rays_r[:, 0].long())]
    assert torch.equal(rr[:, 0], rays[:, 0]) and torch.equal(rr[:, 2], rays[:, 2])
    total = int(cnt[0])
    counts = rr[:, 2].long()
    starts_ours = torch.cumsum(counts, 0) - counts
    within = torch.arange(total, device=cuda) - torch.repeat_interleave(starts_ours, counts)
    idx_ref = torch.repeat_interleave(rr[:, 1].long(), counts) + within
    same_bits(x[:total], x_r[idx_ref], "xyzs")
    same_bits(d[:total], d_r[idx_ref], "dirs")
    same_bits(dl[:total], dl_r[idx_ref], "deltas")


@pytest.mark.parametrize("n_step", [1, 3, 8])
def test_inference_march_and_composites_bit_identical(cuda, ref, window, n_step):
    rm = ref["rm"]
    ro, rd, nears, fars, bitfield, _, _ = window
    N = ro.shape[0]
    g = torch.Generator().manual_seed(10 + n_step)
    alive = torch.sort(torch.randperm(N, generator=g)[: N // 2]).values.int().to(cuda)
    n_alive = alive.shape[0]
    rays_t = nears.clone()
    rays_t[alive[::3].long()] += 0.4
    M = n_alive * n_step + (128 - (n_alive * n_step) % 128)
    x_r, d_r, dl_r = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda), torch.zeros(M, 2, device=cuda)
    rm.march_rays(n_alive, n_step, alive, rays_t, ro, rd, 2.0, 1.0 / 256, 1024, 2, 128, bitfield, nears, fars, x_r, d_r, dl_r, torch.zeros(n_alive, device=cuda))
    x, d, dl = raymarching.march_rays(n_alive, n_step, alive, rays_t, ro, rd, 2.0, bitfield, 2, 128, nears, fars, 128, False, 1.0 / 256, 1024)
    same_bits(x, x_r, "xyzs")
    same_bits(d, d_r, "dirs")
    same_bits(dl, dl_r, "deltas")
    sig = (torch.rand(M, generator=g) * 80).to(cuda)
    rgb = torch.rand(M, 3, generator=g).to(cuda)
    ws0 = torch.rand(N, generator=torch.Generator().manual_seed(3)) * 0.5
    ws0[::5] = 0.99995          # transmittance already below T_thresh: these rays terminate in this call whatever they sample
    ws0 = ws0.to(cuda)
    state = lambda: (alive.clone(), rays_t.clone(), ws0.clone(), torch.zeros(N, device=cuda), torch.zeros(N, 3, device=cuda))
    a, b = state(), state()
    rm.composite_rays(n_alive, n_step, 1e-4, a[0], a[1], sig, rgb, dl_r, a[2], a[3], a[4])
    raymarching.composite_rays(n_alive, n_step, b[0], b[1], sig, rgb, dl, b[2], b[3], b[4], 1e-4)
    for u, v, what in zip(b, a, ("rays_alive", "rays_t", "weights_sum", "depth", "image")):
        same_bits(u, v, what)
    assert int((b[0] < 0).sum()) > 0
    for nc in (3, 50):
        inp = torch.rand(M, nc, generator=g).to(cuda)
        o_r, o = torch.zeros(N, nc, device=cuda), torch.zeros(N, nc, device=cuda)
        a, b = state(), state()
        rm.composite_rays_flex(n_alive, n_step, nc, 1e-4, a[0], a[1], sig, inp, dl_r, a[2], o_r)
        raymarching.composite_rays_flex(n_alive, n_step, nc, b[0], b[1], sig, inp, dl, b[2], o, 1e-4)
        same_bits(o, o_r, f"flex output nc={nc}")


def test_training_composites(cuda, ref, window):
    rm = ref["rm"]
    ro, rd, nears, fars, bitfield, _, _ = window
    N = ro.shape[0]
    cnt = torch.zeros(2, dtype=torch.int32, device=cuda)
    _, _, dl, rays = raymarching.march_rays_train(ro, rd, 2.0, bitfield, 2, 128, nears, fars, cnt, -1, False, 128, True, 1.0 / 128, 1024)
    total = int(cnt[0])
    dl = dl[:total].contiguous()
    g = torch.Generator().manual_seed(4)
    sig = (torch.rand(total, generator=g) * 20).to(cuda)
    rgb = torch.rand(total, 3, generator=g).to(cuda)
    ws_r, dp_r, im_r = torch.empty(N, device=cuda), torch.empty(N, device=cuda), torch.empty(N, 3, device=cuda)
    rm.composite_rays_train_forward(sig, rgb, dl, rays, total, N, 1e-4, ws_r, dp_r, im_r)
    s2, c2 = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    ws, dp, im = raymarching.composite_rays_train(s2, c2, dl, rays, 1e-4)
    gws, gim = torch.rand(N, generator=g).to(cuda), torch.rand(N, 3, generator=g).to(cuda)
    gs_r, gc_r = torch.zeros_like(sig), torch.zeros_like(rgb)
    rm.composite_rays_train_backward(gws, gim, sig, rgb, dl, rays, ws_r, im_r, total, N, 1e-4, gs_r, gc_r)
    ((ws * gws).sum() + (im * gim).sum()).backward()
    close = lambda u, v, what: np.testing.assert_allclose(u.detach().cpu().numpy(), v.cpu().numpy(), rtol=0, atol=2e-6 * float(v.abs().max()), err_msg=what)
    close(ws, ws_r, "weights_sum")       # (16 lanes per ray, prefix products: another summation order than the reference's one-thread loop)
    close(dp, dp_r, "depth")
    close(im, im_r, "image")
    close(s2.grad, gs_r, "grad_sigmas")
    close(c2.grad, gc_r, "grad_rgbs")
    for nc in (1, 33):                 # the flex pair, quirks included (>= M overflow test, break-before-write in the backward): bit for bit
        inp = torch.rand(total, nc, generator=g).to(cuda)
        of_r = torch.empty(N, nc, device=cuda)
        rm.composite_rays_flex_train_forward(sig, inp, dl, rays, total, N, nc, 1e-4, of_r)
        i2 = inp.clone().requires_grad_(True)
        of = raymarching.composite_rays_flex_train(sig, i2, dl, rays, 1e-4)
        same_bits(of.detach(), of_r, f"flex train output nc={nc}")
        go = torch.rand(N, nc, generator=g).to(cuda)
        gi_r = torch.zeros_like(inp)
        rm.composite_rays_flex_train_backward(go, sig, inp, dl, rays, of_r, total, N, nc, 1e-4, gi_r)
        (of * go).sum().backward()
        same_bits(i2.grad, gi_r, f"flex train grad nc={nc}")


def test_sh_and_hsv_against_the_reference_kernels(cuda, ref):
    sh, pal = ref["sh"], ref["pal"]
    g = torch.Generator().manual_seed(5)
    B = 100003
    d = torch.randn(B, 3, generator=g)
    d = (d / d.norm(dim=1, keepdim=True)).to(cuda)
    for degree in (1, 2, 3, 4, 5, 8):
        y_r = torch.empty(B, degree * degree, device=cuda)
        dy_r = torch.empty(B, 3 * degree * degree, device=cuda)
        sh.sh_encode_forward(d, y_r, B, 3, degree, dy_r)
        x = d.clone().requires_grad_(True)
        y = shencoder.SHEncoder(degree=degree).to(cuda)(x)
        # degree <= 4 (what the models use): 2e-7 measured; the degree-8 polynomials cancel in fp32 on both sides (4.6e-6 at 0.015 % of the entries)
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_r.cpu().numpy(), rtol=0, atol=1e-6 if degree <= 4 else 1e-5, err_msg=f"SH degree {degree}")
        gy = torch.rand(B, degree * degree, generator=g).to(cuda)
        gi_r = torch.zeros(B, 3, device=cuda)
        sh.sh_encode_backward(gy, d, B, 3, degree, dy_r, gi_r)
        (y * gy).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), gi_r.cpu().numpy(), rtol=0, atol=3e-5 * degree, err_msg=f"SH grad degree {degree}")
    px = torch.rand(B, 3, generator=g).to(cuda)
    px[:4] = torch.tensor([[0.5, 0.5, 0.5], [1, 0, 0], [0, 0, 0], [0.2, 0.7, 0.7]], device=cuda)
    h_r = torch.empty(B, 3, device=cuda)
    pal.rgb_to_hsv(B, px, h_r)
    hsv = palette_utils.rgb_to_hsv(px)
    np.testing.assert_allclose(hsv.cpu().numpy(), h_r.cpu().numpy(), rtol=0, atol=360 * 2e-7)
    b_r = torch.empty(B, 3, device=cuda)
    pal.hsv_to_rgb(B, h_r, b_r)
    np.testing.assert_allclose(palette_utils.hsv_to_rgb(h_r).cpu().numpy(), b_r.cpu().numpy(), rtol=0, atol=2e-7)


def test_frame_under_the_reference_kernels_equals_the_frame_under_ours(cuda, ref):
    """The per-op loop (this repository's mirror of run_cuda in compat mode) once over this repository's kernels and once with the reference's own
    march / composite / SH kernels swapped in (the hash grid stays this repository's: gridencoder.cu is unbuildable for HIP): same samples, and
    images that agree to 1e-6 -- the two kernel sets are interchangeable under the reference's control flow."""
    from oracle import ref_ops
    from palettenerf_amd import network
    H = W = 96
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro.to(cuda), rd.to(cuda)

    def frame():
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=1.0, min_near=0.2)
        scene.seed_field_(m, 0)
        m = m.to(cuda).eval()
        m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
        m.density_bitfield.copy_(raymarching.packbits(m.density_grid, 0.5))
        m.count_rendered = True
        with torch.no_grad():
            return m.render(ro, rd, perturb=False, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4)

    ours = frame()
    with ref_ops.swapped_in():
        theirs = frame()
    assert int(ours["rendered"].sum()) == int(theirs["rendered"].sum()) > 50_000
    assert float((ours["image"] - theirs["image"]).abs().max()) <= 1e-6
    assert torch.equal(ours["weights_sum"] > 0, theirs["weights_sum"] > 0)
