"""ctypes binding of libpnr_hip.so -- the C ABI declared in include/pnr.h.

There is NO fallback: if the library is missing or a symbol is absent this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNR_LIB_PATH") or os.path.join(_HERE, "libpnr_hip.so")   # PNR_LIB_PATH: an experiment build (palettenerf_amd.build --variant); still no fallback

_u32, _f32, _int, _ptr, _u64 = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/pnr.h one to one
SIGNATURES = {
    "pnr_abi_version": [],
    "pnr_set_option": [ctypes.c_char_p, _int],
    "pnr_error_string": [_int],
    "pnr_scan_scratch_bytes": [_u32],
    "pnr_near_far_from_aabb": [_ptr, _ptr, _ptr, _u32, _f32, _ptr, _ptr, _ptr],
    "pnr_sph_from_ray": [_ptr, _ptr, _f32, _u32, _ptr, _ptr],
    "pnr_morton3d": [_ptr, _u32, _ptr, _ptr],
    "pnr_morton3d_invert": [_ptr, _u32, _ptr, _ptr],
    "pnr_packbits": [_ptr, _u32, _f32, _ptr, _ptr],
    "pnr_march_rays_train": [_ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                             _ptr, _ptr, _ptr],
    "pnr_composite_rays_train_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr],
    "pnr_composite_rays_train_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _f32, _ptr, _ptr, _ptr],
    "pnr_composite_rays_flex_train_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _f32, _ptr, _ptr],
    "pnr_composite_rays_flex_train_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _f32, _ptr, _ptr],
    "pnr_spread_ray_to_sample": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "pnr_march_rays": [_u32, _u32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_composite_rays": [_u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_composite_rays_flex": [_u32, _u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_composite_rays_flex_multi": [_u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr],
    "pnr_occupancy_mip_bytes": [_u32, _u32],
    "pnr_build_occupancy_mip": [_ptr, _u32, _u32, _f32, _ptr, _ptr],
    "pnr_march_rays_mip": [_u32, _u32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_march_rays_fill": [_u32, _u32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr],
    "pnr_march_rays_train_mip": [_ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                 _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_compact_alive": [_u32, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_grid_encode_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32, _int, _int, _ptr],
    "pnr_grid_encode_forward_layout": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32, _int, _int, _int, _ptr],
    "pnr_grid_encode_forward_pair": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _f32, _u32, _u32, _int, _ptr],
    "pnr_grid_encode_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _ptr, _u32, _int, _int, _ptr],
    "pnr_nerf_field_packed_bytes": [],
    "pnr_nerf_field_pack": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _ptr],
    "pnr_nerf_field_forward": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr, _int, _f32, _ptr],
    "pnr_nerf_density_forward": [_ptr, _ptr, _u32, _f32, _ptr, _ptr, _int, _f32, _ptr],
    "pnr_nerf_frame_workspace_bytes": [_u32],
    "pnr_nerf_render_frame": [_ptr, _ptr],
    "pnr_nerf_render_frame_submit": [_ptr, _ptr],
    "pnr_nerf_render_frame_finish": [_ptr, _ptr],
    "pnr_palette_field_packed_bytes": [_u32, _u32, _int],
    "pnr_palette_aux_channels": [_u32, _u32],
    "pnr_palette_field_pack": [_ptr, _ptr, _ptr],
    "pnr_palette_field_forward": [_ptr, _ptr],
    "pnr_palette_frame_workspace_bytes": [_u32, _u32, _u32, _int],
    "pnr_palette_render_frame": [_ptr, _ptr],
    "pnr_palette_render_frame_submit": [_ptr, _ptr],
    "pnr_palette_render_frame_finish": [_ptr, _ptr],
    "pnr_sh_encode_forward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "pnr_sh_encode_cat_forward": [_ptr, _ptr, _u32, _ptr, _u32, _u32, _ptr],
    "pnr_sigma_geo_cat_forward": [_ptr, _u32, _ptr, _u32, _u32, _ptr, _ptr, _ptr],
    "pnr_sigma_geo_cat_backward": [_ptr, _u32, _ptr, _ptr, _u32, _u32, _ptr, _ptr],
    "pnr_sh_encode_backward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "pnr_rgb_to_hsv": [_u32, _ptr, _ptr, _ptr],
    "pnr_hsv_to_rgb": [_u32, _ptr, _ptr, _ptr],
    "pnr_rgb_histogram": [_ptr, _ptr, _u32, _int, _ptr, _ptr, _ptr],
    "pnr_grid_backward_binned_workspace_bytes": [_u32, _u32, _u64],
    "pnr_grid_encode_backward_binned": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u64, _ptr, _u64, _ptr],
    "pnr_linear_wgrad_workspace_bytes": [_u32, _u32, _u32],
    "pnr_linear_wgrad": [_ptr, _int, _ptr, _int, _u32, _u32, _u32, _ptr, _int, _ptr, _u64, _ptr],
    "pnr_mlp_packed_bytes": [_ptr],
    "pnr_mlp_pack": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_mlp_forward": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr],
    "pnr_mlp_backward_workspace_bytes": [_ptr, _u32],
    "pnr_mlp_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _ptr],
    "pnr_mlp_forward_lm": [_ptr, _ptr, _ptr, _u32, _ptr, _u32, _ptr, _ptr],
    "pnr_mlp_backward_lm": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _ptr],
    "pnr_linear_bgrad": [_ptr, _int, _u32, _u32, _ptr, _int, _ptr, _u64, _ptr],
    "pnr_train_loss_workspace_bytes": [_u32],
    "pnr_train_loss_forward": [_ptr, _ptr],
    "pnr_train_loss_backward": [_ptr, _ptr],
    "pnr_palette_field_stages_aux": [_u32, _u32, _int],
    "pnr_interleave_tables": [_ptr, _ptr, _u64, _ptr, _ptr],
    "pnr_interleave_tables3": [_ptr, _ptr, _ptr, _u64, _ptr, _ptr],
    "pnr_palette_train_shade_workspace_bytes": [_u32],
    "pnr_palette_train_shade_forward": [_u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "pnr_palette_train_shade_backward": [_u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _ptr],
    "pnr_palette_heads_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "pnr_palette_heads_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "pnr_image_to_uint8": [_ptr, _u64, _int, _ptr, _ptr],
    "pnr_adam_max_tensors": [],
    "pnr_adam_step": [_ptr, _u32, _ptr, _ptr],
    "pnr_get_rays": [_ptr, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _ptr, _u32, _ptr, _ptr, _ptr],
    "pnr_checksum": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr],
    "pnr_occupancy_workspace_bytes": [_u32, _u32, _u32],
    "pnr_occupancy_samples": [_ptr],
    "pnr_occupancy_update": [_ptr, _ptr],
    "pnr_occupancy_begin": [_ptr, _ptr],
    "pnr_occupancy_points": [_ptr, _u32, _u32, _ptr, _ptr],
    "pnr_occupancy_scatter": [_ptr, _ptr, _ptr, _u32, _ptr],
    "pnr_occupancy_commit": [_ptr, _ptr],
    "pnr_mark_untrained_grid": [_ptr, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _f32, _f32, _int, _ptr, _ptr, _ptr],
}
_RESTYPES = {"pnr_occupancy_workspace_bytes": _u64, "pnr_occupancy_samples": _u32, "pnr_adam_max_tensors": _u32, "pnr_error_string": ctypes.c_char_p, "pnr_scan_scratch_bytes": _u64, "pnr_nerf_field_packed_bytes": _u64, "pnr_occupancy_mip_bytes": _u64, "pnr_nerf_frame_workspace_bytes": _u64, "pnr_palette_field_packed_bytes": _u64, "pnr_palette_frame_workspace_bytes": _u64,
             "pnr_palette_aux_channels": _u32, "pnr_linear_wgrad_workspace_bytes": _u64, "pnr_grid_backward_binned_workspace_bytes": _u64,
             "pnr_palette_train_shade_workspace_bytes": _u64, "pnr_train_loss_workspace_bytes": _u64, "pnr_mlp_packed_bytes": _u64, "pnr_mlp_backward_workspace_bytes": _u64}

class AdamTensor(ctypes.Structure):
    """Mirror of `pnr_adam_tensor` (include/pnr.h)."""
    _fields_ = [("param", _ptr), ("grad", _ptr), ("exp_avg", _ptr), ("exp_avg_sq", _ptr), ("n", _u64)]


class AdamScalars(ctypes.Structure):
    """Mirror of `pnr_adam_scalars` (include/pnr.h)."""
    _fields_ = [(n, _f32) for n in ("one_minus_beta1", "beta2", "one_minus_beta2", "bias_correction2_sqrt", "eps", "neg_step_size", "inv_grad_scale")]


MLP_OUT_SIGMOID = 0x100   # PNR_MLP_OUT_SIGMOID


FLEX_MAX_MAPS = 8


class FlexMap(ctypes.Structure):
    """Mirror of `pnr_flex_map` (include/pnr.h)."""
    _fields_ = [("n_channel", _u32), ("input", _ptr), ("output", _ptr)]


class MlpDesc(ctypes.Structure):
    """Mirror of `pnr_mlp_desc` (include/pnr.h)."""
    _fields_ = [("n_layers", _u32), ("dims", _u32 * 4), ("activation", _int)]


class OccupancyArgs(ctypes.Structure):
    """Mirror of `pnr_occupancy_args` (include/pnr.h)."""
    _fields_ = [("C", _u32), ("H", _u32), ("bound", _f32), ("density_grid", _ptr), ("density_bitfield", _ptr), ("mip", _ptr), ("density_scale", _f32),
                ("decay", _f32), ("density_thresh", _f32), ("mode", _int), ("n_partial", _u32), ("noise", _ptr), ("coords", _ptr), ("occ_rand", _ptr),
                ("embeddings", _ptr), ("offsets", _ptr), ("num_levels", _u32), ("S", _f32), ("base_resolution", _u32), ("gridtype", _u32),
                ("packed_sigma_net", _ptr), ("workspace", _ptr), ("workspace_bytes", _u64), ("state", _ptr), ("points_out", _ptr)]


class NerfFrameArgs(ctypes.Structure):
    """Mirror of `pnr_nerf_frame_args` (include/pnr.h)."""
    _fields_ = [("N", _u32), ("rays_o", _ptr), ("rays_d", _ptr), ("nears", _ptr), ("fars", _ptr), ("bitfield", _ptr), ("mip", _ptr), ("bound", _f32),
                ("C", _u32), ("H", _u32), ("dt_gamma", _f32), ("max_steps", _u32), ("T_thresh", _f32), ("embeddings", _ptr), ("offsets", _ptr),
                ("num_levels", _u32), ("S", _f32), ("base_resolution", _u32), ("gridtype", _u32), ("packed_weights", _ptr), ("field_precision", _int), ("density_scale", _f32),
                ("weights_sum", _ptr), ("depth", _ptr), ("image", _ptr), ("workspace", _ptr), ("workspace_bytes", _u64), ("stats", _ptr), ("kernel_ms", _ptr), ("ray_order", _ptr),
                ("finish", _int), ("bg_color", _f32 * 3), ("bg_map", _ptr), ("table_dtype", _int), ("enc_scale", _f32 * 3), ("watch_overflow", _int),
                ("aabb", _ptr), ("min_near", _f32), ("depth_raw", _ptr)]


MAX_BASIS, MAX_CLIP = 10, 32   # PNR_MAX_BASIS, PNR_MAX_CLIP


class PaletteEdit(ctypes.Structure):
    """Mirror of `pnr_palette_edit` (include/pnr.h)."""
    _fields_ = [("mode", _int), ("delta_hsv", (_f32 * 3) * MAX_BASIS), ("has_mean_xyz", _int), ("mean_xyz", _f32 * 3), ("std_xyz", _f32),
                ("has_mean_clip", _int), ("mean_clip", _f32 * MAX_CLIP), ("std_clip", _f32), ("weight_mode", _int),
                ("dI", _f32 * MAX_BASIS), ("dP", (_f32 * 3) * MAX_BASIS), ("ddelta", ((_f32 * 3) * 3) * MAX_BASIS)]


class PaletteFrameArgs(ctypes.Structure):
    """Mirror of `pnr_palette_frame_args` (include/pnr.h)."""
    _fields_ = [("base", NerfFrameArgs), ("embeddings_palette", _ptr), ("embeddings_clip", _ptr),
                ("num_basis", _u32), ("clip_dim", _u32), ("pred_clip", _int), ("offsets_weight", _f32), ("view_dep_weight", _f32), ("aux_map", _ptr), ("embeddings_pair", _ptr), ("embeddings_triple", _ptr),
                ("edit", _ptr)]


class PaletteWeights(ctypes.Structure):
    """Mirror of `pnr_palette_weights` (include/pnr.h)."""
    _fields_ = [(n, _ptr) for n in ("sigma0", "sigma1", "diff0", "diff1", "diff2", "color0", "color1", "color2", "basis0", "basis1",
                                    "offsets_radiance", "omega", "clip0", "clip1", "basis_color", "or_bias")] + [("num_basis", _u32), ("clip_dim", _u32), ("pred_clip", _int), ("precision", _int)]


class PaletteFieldArgs(ctypes.Structure):
    """Mirror of `pnr_palette_field_args` (include/pnr.h)."""
    _fields_ = [("ctl", _ptr), ("B", _u32), ("enc", _ptr), ("enc_palette", _ptr), ("enc_clip", _ptr), ("level_stride", _u32), ("dirs", _ptr),
                ("deltas", _ptr), ("packed", _ptr), ("num_basis", _u32), ("clip_dim", _u32),
                ("pred_clip", _int), ("density_scale", _f32), ("offsets_weight", _f32), ("view_dep_weight", _f32), ("aux_stride", _u32),
                ("sigmas", _ptr), ("rgbs", _ptr), ("aux", _ptr), ("rays_alive", _ptr), ("weights_sum", _ptr), ("aux_map", _ptr), ("T_thresh", _f32),
                ("precision", _int), ("edit", _ptr), ("xyzs", _ptr), ("edit_device", _ptr), ("enc_scale", _f32 * 3), ("overflow_flag", _ptr), ("tile_counter", _ptr),
                ("rays_t", _ptr), ("weights_sum_rw", _ptr), ("depth", _ptr), ("image", _ptr), ("rays_alive_rw", _ptr), ("counts_cur", _ptr)]   # frame loop only: leave NULL


class TrainLossArgs(ctypes.Structure):
    """Mirror of `pnr_train_loss_args` (include/pnr.h)."""
    _fields_ = ([(n, _u32) for n in ("N", "n_channel", "num_basis", "clip_dim")]
                + [(n, _ptr) for n in ("weights_sum", "depth_raw", "image_raw", "all_map", "nears", "fars", "bg_color")]
                + [("bg_const", _f32), ("bg_mode", ctypes.c_int32)]
                + [(n, _ptr) for n in ("gt_rgb", "gt_clip", "gt_weights", "basis_color", "basis_color_origin")]
                + [(n, _f32) for n in ("lambda_sparsity", "lambda_offsets", "lambda_view_dep", "lambda_smooth", "lambda_weight", "lambda_palette")]
                + [(n, _ptr) for n in ("image", "depth", "direct_rgb", "loss_ray", "terms", "grad_loss", "grad_weights_sum", "grad_image_raw",
                                       "grad_all_map", "grad_basis_color", "workspace")]
                + [("workspace_bytes", _u64)])


TRAIN_LOSS_TERMS = 10   # PNR_TRAIN_LOSS_TERMS


_lib = None


def load():
    """Load the HIP library; raises RuntimeError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"palettenerf_amd: {LIB_PATH} is missing. Build it with `python -m palettenerf_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"palettenerf_amd: symbol {name} missing from {LIB_PATH}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _int)
    _lib = lib
    return lib


def check(rc, what):
    """Turn a non-zero C-ABI return into RuntimeError (the reference raises RuntimeError via TORCH_CHECK)."""
    if rc != 0:
        msg = load().pnr_error_string(rc)
        raise RuntimeError(f"{what}: {msg.decode() if msg else rc}")


_FN = {}     # name -> the loaded entry point (a getattr on the CDLL per call is a dict miss + a descriptor each time)


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc:
        check(rc, name)
