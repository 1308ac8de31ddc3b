#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only: it reads
/root/reference, which does not exist on the GPU box; the fixtures it writes are data).

What is pinned by the reference itself:
  1. sh_torch.npz       -- the reference's own pure-PyTorch SH encoder `SHEncoder_torch`
                           (testing/test_shencoder.py:8-89) evaluated on seeded unit vectors, deg 1..5.
  2. sh_cuda_expr.npz   -- the 64 + 3*64 polynomial expressions of kernel_sh
                           (shencoder/src/shencoder.cu:50-120, 131-349) parsed as arithmetic expressions
                           and evaluated with NumPy float64 at seeded OFF-sphere points, degree 8 incl.
                           the analytic derivative tables.  (No compilation: the expressions are data.)
  3. frame_nerf_*.npz, frame_palette_*.npz, train_*.npz
                        -- the reference's Python callers imported as they are
                           (nerf/renderer.py:NeRFRenderer.run_cuda, nerf/network.py:NeRFNetwork,
                           palette/renderer.py:PaletteRenderer.run_cuda, palette/network.py:PaletteNetwork,
                           encoding.py, activation.py) with the CPU oracle injected as the
                           `raymarching` / `gridencoder` / `shencoder` / `palette.utils` extension
                           modules (those are CUDA extensions that cannot be built here).  This pins
                           the control flow: n_step schedule, order-preserving compaction, composite
                           call order, bg mix, depth normalisation, palette colour-basis composite.
Weights are NOT stored (50 MB tables): fixtures carry the seed; palettenerf_amd.scene.seed_field_
regenerates them bit-identically with the CPU torch.Generator.
"""
import ast
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from palettenerf_amd import scene  # noqa: E402


# ------------------------------------------------------------------------------------------ SH
def gen_sh_torch():
    src = open(os.path.join(REF, "testing", "test_shencoder.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "SHEncoder_torch"][0]
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(ast.Module([cls], []), "SHEncoder_torch", "exec"), ns)
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((257, 3)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    out = {"x": x}
    for deg in range(1, 6):
        out[f"y{deg}"] = ns["SHEncoder_torch"](degree=deg)(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_torch.npz"), **out)


def gen_sh_cuda_expr():
    src = open(os.path.join(REF, "shencoder", "src", "shencoder.cu")).read()
    rng = np.random.default_rng(4321)
    P = rng.uniform(-1.2, 1.2, size=(64, 3)).astype(np.float32).astype(np.float64)  # off the unit sphere on purpose; fp32-representable
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    env = dict(x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z, xyz=x * y * z, pow=np.power)
    env.update(x4=env["x2"] ** 2, y4=env["y2"] ** 2, z4=env["z2"] ** 2)
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])
    tables = {}
    for name in ("outputs", "dx", "dy", "dz"):
        vals = np.zeros((64, 64))
        found = 0
        for m in re.finditer(r"^\s*" + name + r"\[(\d+)\]\s*=\s*([^;]+);", src, flags=re.M):
            k, expr = int(m.group(1)), m.group(2)
            expr = re.sub(r"(\d+\.?\d*(?:[eE][-+]?\d+)?)f\b", r"\1", expr)  # drop float suffixes
            vals[:, k] = eval(expr, {"__builtins__": {}}, env) * np.ones(64)
            found += 1
        assert found == 64, (name, found)
        tables[name] = vals
    np.savez_compressed(os.path.join(HERE, "sh_cuda_expr.npz"), points=P, y=tables["outputs"], dx=tables["dx"], dy=tables["dy"], dz=tables["dz"])


# ------------------------------------------------------------------------------------------ frames
from oracle.facade import make_oracle_modules, _t  # noqa: E402


class _Stub(types.ModuleType):
    """Stands in for harness-only third-party imports (trimesh, cv2, ...) the hot path never calls."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        raise RuntimeError(f"harness-only dependency {self.__name__} was called on the hot path")


def import_reference():
    """The reference's Python, imported as it is, over the CPU oracle:
      * its operator wrappers -- gridencoder/grid.py (GridEncoder, _grid_encode), shencoder/sphere_harmonics.py (SHEncoder, _sh_encoder) and
        raymarching/raymarching.py -- are the reference's OWN files: each tries `import _<name> as _backend` first (raymarching.py:9-12),
        and oracle/native_facade.py provides those three pybind modules with the reference's C++ signatures over the C oracle.  So level
        offsets, per_level_scale, the [L,B,C] buffer and its permute, the half-table cast under autocast, the (x + bound) / (2 bound) map,
        the zero-fill contracts and the autograd.Function plumbing of every fixture are the reference's;
      * six functions of raymarching.py force `.cuda()` on their inputs (near_far_from_aabb :34, morton3D :94, morton3D_invert :116,
        packbits :141, march_rays_train :187, march_rays :373) and cannot run here: the package attributes are replaced by the oracle's
        facades (oracle/facade.py), which restate those wrappers;
      * palette/utils.py imports cv2 / skimage / rgbsg: the module is a facade with the three functions the renderer takes from it;
      * harness-only third-party imports are stubs that raise when called."""
    from oracle.native_facade import make_native_backends
    rm_f, _ge_f, _sh_f, pu = make_oracle_modules()
    nrm, nge, nsh = make_native_backends()
    sys.modules.update({"_raymarching": nrm, "_gridencoder": nge, "_shencoder": nsh, "palette.utils": pu})
    for name in ("trimesh", "cv2", "mcubes", "tensorboardX", "torch_ema", "lpips", "kornia", "imageio"):
        sys.modules.setdefault(name, _Stub(name))
    nu = types.ModuleType("nerf.utils")
    nu.custom_meshgrid = lambda *a: torch.meshgrid(*a, indexing="ij")
    nu.srgb_to_linear = lambda x: torch.where(x < 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    sys.modules["nerf.utils"] = nu
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)     # torch.cuda.amp.custom_fwd is deprecated in torch 2.10; the reference uses it
        import raymarching as ref_raymarching  # noqa  (the reference's package: /root/reference/raymarching)
        import gridencoder as ref_gridencoder  # noqa
        import shencoder as ref_shencoder  # noqa
    assert ref_raymarching.__file__.startswith(REF) and ref_gridencoder.__file__.startswith(REF) and ref_shencoder.__file__.startswith(REF)
    for name in ("near_far_from_aabb", "march_rays", "march_rays_train", "morton3D", "morton3D_invert", "packbits"):
        setattr(ref_raymarching, name, getattr(rm_f, name))
    import nerf.network as ref_nerf_network  # noqa
    import palette.network as ref_palette_network  # noqa
    import palette.renderer as ref_palette_renderer  # noqa
    return ref_nerf_network, ref_palette_network, ref_palette_renderer


def frame_inputs(H, W, elev=30.0, azim=45.0):
    pose = torch.from_numpy(scene.lookat_pose(elevation_deg=elev, azimuth_deg=azim))[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    return ro, rd


def setup_model(model, density_grid, thresh=0.5):
    model.density_grid.copy_(torch.from_numpy(density_grid))
    model.density_bitfield.copy_(torch.from_numpy(oracle.packbits(density_grid, thresh)))


FRAME_CASES = [
    # name, H, W, dt_gamma, density_scale, seed
    ("a", 40, 40, 0.0, 1.0, 0),          # blender-style, opaque-ish field
    ("b", 36, 28, 1.0 / 128, 0.02, 1),   # cone stepping (LLFF/Mip360 default), translucent: long marches, many iterations
]


def gen_frames():
    ref_nerf, ref_pal, ref_pal_r = import_reference()
    grid = scene.brick_density_grid()
    for name, H, W, dt_gamma, dscale, seed in FRAME_CASES:
        ro, rd = frame_inputs(H, W)
        # ---------------- NeRF inference
        m = ref_nerf.NeRFNetwork(bound=2, cuda_ray=True, density_scale=dscale, min_near=0.2)
        scene.seed_field_(m, seed)
        setup_model(m, grid)
        m.eval()
        with torch.no_grad():
            r = m.run_cuda(ro, rd, dt_gamma=dt_gamma, bg_color=None, perturb=False, max_steps=1024, T_thresh=1e-4)
        np.savez_compressed(os.path.join(HERE, f"frame_nerf_{name}.npz"), H=H, W=W, dt_gamma=dt_gamma, density_scale=dscale, seed=seed,
                            image=r["image"].numpy(), depth=r["depth"].numpy(), weights_sum=r["weights_sum"].numpy())
        print("nerf", name, float(r["weights_sum"].mean()), float(r["image"].mean()))
        # ---------------- NeRF training-mode forward + grads
        m.train()
        m.zero_grad()
        r = m.run_cuda(ro, rd, dt_gamma=dt_gamma, perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
        loss = (r["image"] ** 2).mean() + 0.1 * r["weights_sum"].mean()
        loss.backward()
        ge_ = m.encoder.embeddings.grad
        nz = ge_.abs().sum(1).nonzero()[:, 0]
        sel = nz[:: max(1, nz.numel() // 512)][:512]
        np.savez_compressed(os.path.join(HERE, f"train_nerf_{name}.npz"), H=H, W=W, dt_gamma=dt_gamma, density_scale=dscale, seed=seed,
                            image=r["image"].detach().numpy(), depth=r["depth"].detach().numpy(), weights_sum=r["weights_sum"].detach().numpy(),
                            counter=m.step_counter[0].numpy(), loss=float(loss),
                            grad_color0=m.color_net[0].weight.grad.numpy(), grad_sigma1=m.sigma_net[1].weight.grad.numpy(),
                            grad_emb_rows=sel.numpy(), grad_emb_vals=ge_[sel].numpy(), grad_emb_abs_sum=float(ge_.abs().sum()))
        print("nerf train", name, int(m.step_counter[0, 0]), float(loss))
        # ---------------- Palette inference (all 7 composites) and gui_mode
        opt = types.SimpleNamespace(num_basis=4, clip_dim=16, pred_clip=(name == "b"), use_initialization_from_rgbxy=False, test=True,
                                    color_space="srgb", smooth_sigma_xyz=0.005, smooth_sigma_color=0.2, smooth_sigma_clip=0.0)
        p = ref_pal.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=dscale, min_near=0.2)
        scene.seed_field_(p, seed + 100)
        setup_model(p, grid)
        p.eval()
        with torch.no_grad():
            r = p.run_cuda(ro, rd, dt_gamma=dt_gamma, perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=False)
            # regional edit active: exercises the HSV operators inside the loop
            p.edit = ref_pal_r.RegionEdit(opt)
            p.edit.update_cent(mean_xyz=torch.tensor([0.1, 0.0, -0.2]))
            p.edit.update_std(std_xyz=0.5)
            p.edit.update_delta_hsv(p.basis_color.data.clamp(0, 1), (p.basis_color.data * 0.6 + 0.2).flip(0).clamp(0, 1))
            r2 = p.run_cuda(ro, rd, dt_gamma=dt_gamma, perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=True)
            p.edit = None
        keys = ["image", "depth", "depth_origin", "weights_sum", "clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"]
        np.savez_compressed(os.path.join(HERE, f"frame_palette_{name}.npz"), H=H, W=W, dt_gamma=dt_gamma, density_scale=dscale, seed=seed + 100,
                            pred_clip=opt.pred_clip, edit_image=r2["image"].numpy(), edit_delta_hsv=np.zeros(1),
                            **{k: r[k].numpy() for k in keys})
        print("palette", name, float(r["weights_sum"].mean()), float((r["image"] - r2["image"]).abs().max()))
        # ---------------- Palette training-mode forward + grads (config 3 path)
        p.train()
        p.zero_grad()
        r = p.run_cuda(ro, rd, dt_gamma=dt_gamma, perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
        loss = (r["image"] ** 2).mean() + 0.01 * r["omega_sparsity"].mean() + 0.1 * r["offsets_norm"].mean() + (r["direct_rgb"] ** 2).mean() \
            + 0.1 * (r["clip_feat"] ** 2).mean() + 0.1 * r["basis_acc"].mean()
        loss.backward()
        gp = p.encoder_palette.embeddings.grad
        nz = gp.abs().sum(1).nonzero()[:, 0]
        sel = nz[:: max(1, nz.numel() // 512)][:512]
        tk = ["image", "depth", "weights_sum", "omega_sparsity", "view_dep_norm", "offsets_norm", "direct_rgb", "view_dep_rgb", "diffuse_rgb", "clip_feat", "basis_acc"]
        np.savez_compressed(os.path.join(HERE, f"train_palette_{name}.npz"), H=H, W=W, dt_gamma=dt_gamma, density_scale=dscale, seed=seed + 100,
                            pred_clip=opt.pred_clip, loss=float(loss), counter=p.step_counter[0].numpy(),
                            grad_offsets_radiance=p.offsets_radiance_net.weight.grad.numpy(), grad_basis_color=p.basis_color.grad.numpy(),
                            grad_diff0=p.diff_net[0].weight.grad.numpy(), grad_emb_rows=sel.numpy(), grad_emb_vals=gp[sel].numpy(),
                            grad_emb_abs_sum=float(gp.abs().sum()), encoder_grad_is_none=(p.encoder.embeddings.grad is None),
                            **{k: r[k].detach().numpy() for k in tk})
        print("palette train", name, int(p.step_counter[0, 0]), float(loss))


# ------------------------------------------------------------------------------------------ the reference's GridEncoder under fp16 autocast (-O mode)
def gen_grid_autocast():
    """grid_autocast.npz: the reference's own GridEncoder / _grid_encode (gridencoder/grid.py:19-153, imported from /root/reference) with
    torch's autocast flag forced on -- the CPU build cannot enter CUDA autocast, and the flag is all grid.py:36-39 looks at: it then hands
    the kernel a half copy of the table and returns half features.  Pins the `-O` path of the operator (`embeddings.to(torch.half)`, half
    accumulator, [L,B,C] -> [B, L*C]) against the reference's wrapper; the fp32 call of the same module is stored beside it."""
    import_reference()
    import gridencoder as ref_ge
    enc = ref_ge.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        enc.embeddings.copy_((torch.rand(enc.embeddings.shape, generator=g) - 0.5))
    x = torch.rand(2048, 3, generator=g) * 4 - 2
    x[:4] = torch.tensor([[2.0, 2.0, 2.0], [-2.0, -2.0, -2.0], [2.0000005, 0.0, 0.0], [0.0, 0.0, 0.0]])
    with torch.no_grad():
        y32 = enc(x, bound=2)
        real = torch.is_autocast_enabled
        torch.is_autocast_enabled = lambda *a, **k: True
        try:
            y16 = enc(x, bound=2)
        finally:
            torch.is_autocast_enabled = real
    assert y16.dtype == torch.float16 and y32.dtype == torch.float32
    np.savez_compressed(os.path.join(HERE, "grid_autocast.npz"), seed=77, x=x.numpy(), bound=2.0, y32=y32.numpy(), y16_bits=y16.numpy().view(np.uint16),
                        offsets=enc.offsets.numpy(), per_level_scale=np.float64(enc.per_level_scale))
    print("grid_autocast", y16.shape, float(y32.abs().mean()), float((y16.float() - y32).abs().max()))


# ------------------------------------------------------------------------------------------ palette extras: Stylizer, edit windows, more bases
EXTRA_CASES = [
    # name, H, W, dt_gamma, density_scale, seed, num_basis, pred_clip
    ("style_a", 40, 40, 0.0, 1.0, 100, 4, False),
    ("style_b", 36, 28, 1.0 / 128, 0.02, 101, 4, True),
    ("nb6", 32, 32, 0.0, 1.0, 102, 6, False),
    ("nb8", 30, 26, 1.0 / 128, 0.05, 103, 8, True),
]


def gen_palette_extra():
    """The reference's PaletteRenderer.run_cuda with its own Stylizer / RegionEdit classes (palette/renderer.py:84-183) and with more than
    5 palette bases (main_palette.py:141: num_basis = rows of the extracted palette), the oracle injected as the kernel layer."""
    ref_nerf, ref_pal, ref_pal_r = import_reference()
    grid = scene.brick_density_grid()
    for name, H, W, dt_gamma, dscale, seed, nb, pred_clip in EXTRA_CASES:
        ro, rd = frame_inputs(H, W)
        opt = types.SimpleNamespace(num_basis=nb, clip_dim=16, pred_clip=pred_clip, use_initialization_from_rgbxy=False, test=True,
                                    color_space="srgb", smooth_sigma_xyz=0.005, smooth_sigma_color=0.2, smooth_sigma_clip=0.0)
        p = ref_pal.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=dscale, min_near=0.2)
        scene.seed_field_(p, seed)
        setup_model(p, grid)
        p.eval()
        kw = dict(dt_gamma=dt_gamma, perturb=False, max_steps=1024, T_thresh=1e-4)
        out = dict(H=H, W=W, dt_gamma=dt_gamma, density_scale=dscale, seed=seed, num_basis=nb, pred_clip=pred_clip)
        with torch.no_grad():
            r = p.run_cuda(ro, rd, gui_mode=False, **kw)
            for k in ("image", "depth", "weights_sum", "clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"):
                out[k] = r[k].numpy()
            # Stylizer (gui_mode only: the reference defines no basis maps on that branch)
            p.stylizer = ref_pal_r.Stylizer(opt)
            st = scene.stylizer_state(nb, seed)
            for k, v in st.items():
                getattr(p.stylizer, k).data.copy_(v)
            out["style_image"] = p.run_cuda(ro, rd, gui_mode=True, **kw)["image"].numpy()
            p.stylizer = None
            # RegionEdit with a spatial AND a semantic window; then its weight-preview mode
            p.edit = ref_pal_r.RegionEdit(opt)
            p.edit.update_cent(mean_xyz=torch.tensor([0.1, 0.0, -0.2]), mean_clip=torch.linspace(-0.3, 0.3, 16))
            p.edit.update_std(std_xyz=0.5, std_clip=2.0)
            p.edit.update_delta_hsv(p.basis_color.data.clamp(0, 1), (p.basis_color.data * 0.6 + 0.2).flip(0).clamp(0, 1))
            e1 = p.run_cuda(ro, rd, gui_mode=False, **kw)
            out["edit_image"], out["edit_basis_rgb"] = e1["image"].numpy(), e1["basis_rgb"].numpy()
            p.edit.weight_mode = True
            out["edit_weight_image"] = p.run_cuda(ro, rd, gui_mode=True, **kw)["image"].numpy()
            p.edit = None
        np.savez_compressed(os.path.join(HERE, f"frame_palette_{name}.npz"), **out)
        print("palette extra", name, float(r["weights_sum"].mean()), float(np.abs(out["style_image"] - out["image"]).max()),
              float(np.abs(out["edit_image"] - out["image"]).max()), float(out["edit_weight_image"].mean()))


# ------------------------------------------------------------------------------------------ uniform-sampling path (configs[0])
RUN_CASES = [
    # name, H, W, num_steps, upsample_steps, density_scale, seed
    ("a", 20, 20, 128, 128, 1.0, 0),     # nerf/renderer.py:127 defaults
    ("b", 16, 12, 512, 0, 0.05, 1),      # main_nerf.py:31-32 (--num_steps 512 --upsample_steps 0), translucent field
]


def gen_run():
    """NeRFRenderer.run (nerf/renderer.py:127-255) of the reference as it is, cuda_ray=False, eval mode (deterministic upsampling)."""
    ref_nerf, _, _ = import_reference()
    for name, H, W, ns, us, dscale, seed in RUN_CASES:
        ro, rd = frame_inputs(H, W)
        m = ref_nerf.NeRFNetwork(bound=2, cuda_ray=False, density_scale=dscale, min_near=0.2)
        scene.seed_field_(m, seed)
        m.eval()
        with torch.no_grad():
            r = m.run(ro, rd, num_steps=ns, upsample_steps=us, bg_color=None, perturb=False)
        np.savez_compressed(os.path.join(HERE, f"run_nerf_{name}.npz"), H=H, W=W, num_steps=ns, upsample_steps=us, density_scale=dscale, seed=seed,
                            image=r["image"].numpy(), depth=r["depth"].numpy(), weights_sum=r["weights_sum"].numpy())
        print("run", name, float(r["weights_sum"].mean()), float(r["image"].mean()), float(r["depth"].mean()))


# ------------------------------------------------------------------------------------------ get_rays / checkpoint layout
def _reference_get_rays():
    """nerf/utils.py:get_rays as it is (function body executed from the reference file; its module imports cv2, tensorboardX, ...,
    which are absent here, so only the two function definitions are evaluated)."""
    src = open(os.path.join(REF, "nerf", "utils.py")).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("custom_meshgrid", "get_rays")]
    for f in fns:
        f.decorator_list = []
    from packaging import version as pver
    ns = {"torch": torch, "pver": pver, "np": np}
    exec(compile(ast.Module(fns, []), "nerf_utils_get_rays", "exec"), ns)
    return ns["get_rays"]


def gen_get_rays():
    get_rays = _reference_get_rays()
    H, W = 36, 52
    poses = torch.from_numpy(np.stack([scene.lookat_pose(elevation_deg=30.0, azimuth_deg=45.0), scene.lookat_pose(elevation_deg=-10.0, azimuth_deg=200.0)]))
    intr = scene.intrinsics_from_fov(H, W)
    out = {"poses": poses.numpy(), "intrinsics": np.asarray(intr, np.float64), "HW": np.array([H, W])}
    r = get_rays(poses, intr, H, W, -1)
    out.update(full_o=r["rays_o"].numpy(), full_d=r["rays_d"].numpy())
    torch.manual_seed(11)
    r = get_rays(poses, intr, H, W, 64)
    out.update(rand_inds=r["inds"].numpy(), rand_o=r["rays_o"].numpy(), rand_d=r["rays_d"].numpy())
    torch.manual_seed(12)
    r = get_rays(poses, intr, H, W, 64, patch_size=4)
    out.update(patch_inds=r["inds"].numpy(), patch_d=r["rays_d"].numpy())
    torch.manual_seed(13)
    r = get_rays(poses, intr, H, W, 64, random_size=3)
    out.update(pair_inds=r["inds"].numpy(), pair_d=r["rays_d"].numpy())
    torch.manual_seed(14)
    emap = torch.rand(2, 128 * 128)
    torch.manual_seed(15)
    r = get_rays(poses, intr, H, W, 64, error_map=emap)
    out.update(err_seed_map=np.array([14]), err_inds=r["inds"].numpy(), err_coarse=r["inds_coarse"].numpy(), err_d=r["rays_d"].numpy())
    np.savez_compressed(os.path.join(HERE, "get_rays.npz"), **out)
    print("get_rays", out["full_d"].shape)


def gen_state_dict_layout():
    """Names, shapes and dtypes of the reference models' state_dict (what its checkpoints hold under 'model')."""
    import json
    ref_nerf, ref_pal, _ = import_reference()
    layout = {}
    n = ref_nerf.NeRFNetwork(bound=2, cuda_ray=True)
    layout["nerf"] = {k: [list(v.shape), str(v.dtype)] for k, v in n.state_dict().items()}
    opt = types.SimpleNamespace(num_basis=4, clip_dim=16, pred_clip=True, use_initialization_from_rgbxy=False, test=True,
                                color_space="srgb", smooth_sigma_xyz=0.005, smooth_sigma_color=0.2, smooth_sigma_clip=0.0)
    p = ref_pal.PaletteNetwork(opt, bound=2, cuda_ray=True)
    layout["palette"] = {k: [list(v.shape), str(v.dtype)] for k, v in p.state_dict().items()}
    json.dump(layout, open(os.path.join(HERE, "state_dict_layout.json"), "w"), indent=0, sort_keys=True)
    print("state_dict layout", len(layout["nerf"]), len(layout["palette"]))


# ------------------------------------------------------------------------------------------ RGB histogram (the reference's own compiled C++)
def gen_hist():
    """compute_RGB_histogram of the reference itself: palette/src/bindings.cpp compiled unmodified from /root/reference (oracle/ref_build.py),
    called the way palette/utils.py:129-146 calls it (flattened float32 arrays).  Inputs include the clamp edges (< 0, exactly 0.999, >= 1)."""
    sys.path.insert(0, ROOT)
    from oracle import ref_build
    assert ref_build.build() is not None, "the reference sources are needed to generate this fixture"
    mod = ref_build.load()
    rng = np.random.default_rng(2024)
    rgb = rng.uniform(-0.05, 1.05, size=(4096, 3)).astype(np.float32)
    rgb[:8] = np.array([[0, 0, 0], [1, 1, 1], [0.999, 0.999, 0.999], [0.9990001, 0.5, 0.25], [-1, 2, 0.5], [0.124999, 0.125, 0.125001],
                        [0.5, 0.5, 0.5], [0.998, 0.0009765625, 0.99899995]], np.float32)
    w = rng.uniform(0.0, 3.0, size=4096).astype(np.float32)
    out = {"colors_rgb": rgb, "weights": w}
    for bpc in (1, 2, 3, 5):
        bw, bc = mod.compute_RGB_histogram(rgb.flatten(), w.flatten(), bpc)
        out[f"bin_weights_{bpc}"], out[f"bin_centers_{bpc}"] = np.asarray(bw), np.asarray(bc)
    np.savez_compressed(os.path.join(HERE, "hist.npz"), **out)
    print("hist", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ occupancy maintenance
def gen_occupancy(G=32):
    """occupancy.npz: the reference's own NeRFRenderer.update_extra_state and mark_untrained_grid (nerf/renderer.py:395-561, imported
    unmodified; oracle injected as the kernel layer) run on the CPU with a G^3 grid (the attribute the reference hard-codes to 128 is set
    to G before its buffers are rebuilt -- every expression of the two methods is in terms of self.grid_size).  The random numbers the
    methods draw are recorded (torch.rand_like is made to return multiples of 1/256 so that they store as bytes) together with the points
    they query, the sigmas the field returns and the grids / bitfields they leave."""
    ref_nerf, _, _ = import_reference()
    torch.set_num_threads(1)      # `tmp_grid[cas, indices] = sigmas` with repeated indices: sequential on one thread, the last write wins
    m = ref_nerf.NeRFNetwork(bound=2, cuda_ray=True, density_scale=0.5, min_near=0.2, density_thresh=1e9)
    scene.seed_field_(m, 31)
    m.grid_size = G
    m.density_grid = torch.zeros(m.cascade, G ** 3)
    m.density_bitfield = torch.zeros(m.cascade * G ** 3 // 8, dtype=torch.uint8)
    m.train()
    log = {"rand": [], "randint": [], "points": [], "sigma": []}
    real_rand_like, real_randint, real_density = torch.rand_like, torch.randint, m.density
    gen = torch.Generator().manual_seed(99)

    def rand_like(t, **kw):
        r = real_randint(0, 256, t.shape, generator=gen).to(torch.uint8)
        log["rand"].append(r.numpy().copy())
        return r.to(t.dtype) / 256

    def randint(lo, hi, size, **kw):
        r = real_randint(lo, hi, size, generator=gen, dtype=kw.get("dtype", torch.int64))
        log["randint"].append(r.numpy().copy())
        return r

    def density(x):
        log["points"].append(x.detach().numpy().copy())
        out = real_density(x)
        log["sigma"].append(out["sigma"].detach().numpy().reshape(-1).copy())
        return out

    torch.rand_like, torch.randint, m.density = rand_like, randint, density
    try:
        out = {"G": G, "seed": 31, "bound": 2.0, "density_scale": 0.5}
        # ---- full sweep from an empty grid; threshold = the mean
        m.local_step = 3
        m.step_counter[:3, 0] = torch.tensor([100, 200, 330], dtype=torch.int32)
        m.update_extra_state()
        C = m.cascade
        idx = oracle.morton3D(np.stack(np.meshgrid(*[np.arange(G, dtype=np.int32)] * 3, indexing="ij"), -1).reshape(-1, 3)).astype(np.int64)
        noise = np.zeros((C, G ** 3, 3), np.uint8)
        pts = np.zeros((C, G ** 3, 3), np.float32)
        sig = np.zeros((C, G ** 3), np.float32)
        for c in range(C):       # the reference visits the cells in meshgrid order; rows are stored by Morton index
            noise[c, idx], pts[c, idx], sig[c, idx] = log["rand"][c], log["points"][c], log["sigma"][c]
        # where the reference's CPU arithmetic (torch divides by G - 1) and the canonical GPU form (multiplies by the fp32 reciprocal) give the very
        # same point: only there can sigmas be compared tightly -- one ulp of a coordinate moves sigma by up to ~1e-4 through the finest levels
        oxyz = oracle.occupancy_points(C, G, 2.0, noise.astype(np.float32) / 256)[0].reshape(C, G ** 3, 3)
        out["full_points_same"] = np.packbits((oxyz == pts).all(-1))
        out.update(full_noise_u8=noise, full_points_every16=pts[:, ::16].copy(), full_sigma=sig, full_grid=m.density_grid.numpy().copy(),
                   full_bitfield=m.density_bitfield.numpy().copy(), full_mean=np.float32(m.mean_density), full_mean_count=m.mean_count)
        print("occupancy full: mean", m.mean_density, "occupied bits", int(np.unpackbits(m.density_bitfield.numpy()).sum()), "mean_count", m.mean_count)
        # ---- partial sweep on top of it (iter_density >= 16); threshold = density_thresh
        for k in log:
            log[k].clear()
        m.iter_density = 16
        m.density_thresh = 0.48
        m.density_grid[0, :64] = -1.0           # a few cells mark_untrained_grid would have retired: never updated, never drawn as occupied
        before = m.density_grid.numpy().copy()
        m.update_extra_state(decay=0.9)
        n = G ** 3 // 4
        coords = np.stack([log["randint"][2 * c] for c in range(C)]).astype(np.uint8)            # [C, n, 3]
        rand_mask = np.stack([log["randint"][2 * c + 1] for c in range(C)]).astype(np.int32)       # [C, n]: already < the number of occupied cells
        oxyz = oracle.occupancy_points(C, G, 2.0, np.stack(log["rand"]).astype(np.float32) / 256, coords=coords.astype(np.int32), occ_rand=rand_mask,
                                       density_grid=before, n_partial=n)[0].reshape(C, 2 * n, 3)
        out["part_points_same"] = np.packbits((oxyz == np.stack(log["points"])).all(-1))
        out.update(part_before=before, part_coords_u8=coords, part_occ_rand=rand_mask, part_noise_u8=np.stack(log["rand"]), part_sigma=np.stack(log["sigma"]),
                   part_points_every16=np.stack(log["points"])[:, ::16].copy(), part_grid=m.density_grid.numpy().copy(),
                   part_bitfield=m.density_bitfield.numpy().copy(), part_mean=np.float32(m.mean_density), part_decay=0.9, part_density_thresh=0.48)
        assert coords.shape == (C, n, 3) and rand_mask.shape == (C, n)
        print("occupancy partial: mean", m.mean_density, "occupied bits", int(np.unpackbits(m.density_bitfield.numpy()).sum()))
    finally:
        torch.rand_like, torch.randint = real_rand_like, real_randint
    # ---- mark_untrained_grid: six cameras on a ring looking at the origin + one sitting inside the grid (too-close cells)
    m2 = ref_nerf.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.2)
    m2.grid_size = G
    m2.density_grid = torch.zeros(m2.cascade, G ** 3)
    m2.density_bitfield = torch.zeros(m2.cascade * G ** 3 // 8, dtype=torch.uint8)
    poses = np.stack([scene.lookat_pose(elevation_deg=20.0 + 5 * k, azimuth_deg=60.0 * k, radius=3.2) for k in range(6)]
                     + [scene.lookat_pose(elevation_deg=10.0, azimuth_deg=200.0, radius=0.9)]).astype(np.float32)
    intr = (55.0, 60.0, 32.0, 24.0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        m2.mark_untrained_grid(poses, intr)
    print(buf.getvalue().strip())
    marks = {"mark_poses": poses, "mark_intrinsics": np.array(intr, np.float32), "mark_grid": (m2.density_grid.numpy() < 0)}
    m2.density_grid.zero_()
    m2.filter_close_point = True
    with contextlib.redirect_stdout(io.StringIO()):
        m2.mark_untrained_grid(poses, intr)
    marks["mark_grid_filter_close"] = (m2.density_grid.numpy() < 0)
    out.update({k: (np.packbits(v) if v.dtype == bool else v) for k, v in marks.items()})
    np.savez_compressed(os.path.join(HERE, "occupancy.npz"), **out)
    print("occupancy.npz", os.path.getsize(os.path.join(HERE, "occupancy.npz")) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["sh", "frames"]
    if "occupancy" in which:
        gen_occupancy()
    if "grid_autocast" in which:
        gen_grid_autocast()
    if "hist" in which:
        gen_hist()
    if "palette_extra" in which:
        gen_palette_extra()
    if "run" in which:
        gen_run()
    if "rays" in which:
        gen_get_rays()
    if "layout" in which:
        gen_state_dict_layout()
    if "sh" in which:
        gen_sh_torch()
        gen_sh_cuda_expr()
    if "frames" in which:
        gen_frames()
