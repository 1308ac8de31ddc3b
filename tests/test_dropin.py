"""The drop-in installer exposes the reference's import names and call signatures."""
import inspect
import os
import sys

import pytest


def test_dropin_registers_reference_import_names():
    from palettenerf_amd import dropin
    saved = {k: sys.modules.get(k) for k in ("raymarching", "gridencoder", "shencoder")}
    try:
        dropin.install()
        import raymarching
        from gridencoder import GridEncoder
        from gridencoder.grid import _grid_encode, grid_encode  # testing/test_hashgrid_grad.py imports _grid_encode
        from shencoder import SHEncoder
        from shencoder.sphere_harmonics import sh_encode  # noqa: F401
        for fn in ("near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train", "composite_rays_train",
                   "composite_rays_flex_train", "march_rays", "composite_rays", "composite_rays_flex", "spread_ray_to_sample"):
            assert callable(getattr(raymarching, fn)), fn
        # positional signatures the callers rely on
        sig = lambda f: list(inspect.signature(f).parameters)
        assert sig(raymarching._march_rays.forward)[1:] == ["n_alive", "n_step", "rays_alive", "rays_t", "rays_o", "rays_d", "bound",
                                                           "density_bitfield", "C", "H", "near", "far", "align", "perturb", "dt_gamma", "max_steps"]
        assert sig(raymarching._march_rays_train.forward)[1:] == ["rays_o", "rays_d", "bound", "density_bitfield", "C", "H", "nears", "fars", "step_counter",
                                                                 "mean_count", "perturb", "align", "force_all_rays", "dt_gamma", "max_steps"]
        assert sig(raymarching._composite_rays.forward)[1:] == ["n_alive", "n_step", "rays_alive", "rays_t", "sigmas", "rgbs", "deltas", "weights_sum",
                                                               "depth", "image", "T_thresh"]
        assert sig(raymarching._composite_rays_flex.forward)[1:] == ["n_alive", "n_step", "n_channel", "rays_alive", "rays_t", "sigmas", "input", "deltas",
                                                                    "weights_sum", "output", "T_thresh"]
        assert inspect.signature(raymarching._composite_rays.forward).parameters["T_thresh"].default == 1e-2
        assert inspect.signature(raymarching._composite_rays_train.forward).parameters["T_thresh"].default == 1e-4
        assert sig(_grid_encode.forward)[1:] == ["inputs", "embeddings", "offsets", "per_level_scale", "base_resolution", "calc_grad_inputs", "gridtype",
                                                 "align_corners"]
        assert sig(GridEncoder.__init__)[1:] == ["input_dim", "num_levels", "level_dim", "per_level_scale", "base_resolution", "log2_hashmap_size",
                                                 "desired_resolution", "gridtype", "align_corners"]
        assert sig(SHEncoder.__init__)[1:] == ["input_dim", "degree"] and sig(SHEncoder.forward)[1:] == ["inputs", "size"]
        enc = GridEncoder(level_dim=2, desired_resolution=4096)
        assert enc.output_dim == 32 and tuple(enc.embeddings.shape) == (6328848, 2)
    finally:
        for k, v in saved.items():
            for name in [n for n in sys.modules if n == k or n.startswith(k + ".")]:
                del sys.modules[name]
            if v is not None:
                sys.modules[k] = v


@pytest.mark.skipif(not os.path.isdir("/root/reference/nerf"), reason="reference tree only exists in the build container")
def test_reference_network_constructs_on_dropin_modules():
    """The reference's own nerf/network.py + encoding.py import and build against the drop-in modules."""
    import types
    from palettenerf_amd import dropin, gridencoder, shencoder
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    try:
        dropin.install()
        for name in ("trimesh", "cv2", "mcubes", "tensorboardX", "torch_ema", "lpips", "kornia", "imageio"):
            sys.modules.setdefault(name, types.ModuleType(name))
        nu = types.ModuleType("nerf.utils")
        nu.custom_meshgrid = lambda *a: None
        sys.modules["nerf.utils"] = nu
        sys.path.insert(0, "/root/reference")
        import nerf.network as ref
        m = ref.NeRFNetwork(bound=2, cuda_ray=True)
        assert isinstance(m.encoder, gridencoder.GridEncoder) and isinstance(m.encoder_dir, shencoder.SHEncoder)
        import palettenerf_amd.network as mine
        mm = mine.NeRFNetwork(bound=2, cuda_ray=True)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in mm.state_dict().items()}
        # dropin.fuse_field accepts the REFERENCE's class as it is (same attribute names as the mirror: encoder, encoder_dir, sigma_net, color_net,
        # hidden_dim, ...): forward() becomes an instance attribute, the class and its renderer base stay untouched
        class_forward = type(m).forward
        assert dropin.fuse_field(m) is m and "forward" in m.__dict__ and type(m).forward is class_forward
        assert [tuple(w.shape) for w in m._fused._weights()] == [(64, 32), (16, 64), (64, 31), (64, 64), (3, 64)]
        with pytest.raises(RuntimeError):      # a different architecture is refused, not silently run on the wrong kernel
            dropin.fuse_field(ref.NeRFNetwork(bound=2, cuda_ray=True, hidden_dim=32))
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved]:
            del sys.modules[k]
        sys.modules.update(saved)
