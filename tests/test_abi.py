"""The C-ABI library loads and exports every symbol include/pnr.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pnr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ("pnr_march_rays", "pnr_march_rays_train", "pnr_composite_rays", "pnr_composite_rays_flex", "pnr_grid_encode_forward",
                 "pnr_grid_encode_backward", "pnr_sh_encode_forward", "pnr_rgb_to_hsv", "pnr_hsv_to_rgb", "pnr_morton3d", "pnr_packbits",
                 "pnr_near_far_from_aabb", "pnr_compact_alive"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from palettenerf_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_table_matches_header():
    from palettenerf_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    assert lib.pnr_abi_version() >= 2
    assert lib.pnr_error_string(0) == b"ok"
    assert b"unsupported" in lib.pnr_error_string(-2)
    assert lib.pnr_scan_scratch_bytes(1000) >= 4 * 1000


def test_argument_validation_without_gpu():
    """Entry points validate before launching: these return error codes without touching a device."""
    from palettenerf_amd import _lib
    lib = _lib.load()
    u32, f32, i32 = ctypes.c_uint32, ctypes.c_float, ctypes.c_int
    # C not in {1,2,4,8}  (reference: "GridEncoding: C must be 1, 2, 4, or 8.")
    assert lib.pnr_grid_encode_backward(None, None, None, None, None, u32(8), u32(3), u32(3), u32(16), f32(1.0), u32(16), None, None, u32(0), i32(0), i32(0), None) == -2
    # D not in 1..5
    assert lib.pnr_grid_encode_backward(None, None, None, None, None, u32(8), u32(6), u32(2), u32(16), f32(1.0), u32(16), None, None, u32(0), i32(0), i32(0), None) == -2
    # n_channel > 128 (reference CHECK_CHANNEL)
    assert lib.pnr_composite_rays_flex(u32(1), u32(1), u32(129), f32(1e-4), None, None, None, None, None, None, None, None) == -2
    # SH degree outside [1,8], input_dim != 3
    assert lib.pnr_sh_encode_forward(None, None, u32(4), u32(3), u32(9), None, None) == -2
    assert lib.pnr_sh_encode_forward(None, None, u32(4), u32(2), u32(4), None, None) == -2
    # null pointers
    assert lib.pnr_morton3d(None, u32(4), None, None) == -1
    # empty inputs are a no-op
    assert lib.pnr_morton3d(None, u32(0), None, None) == 0
    # round 6 (ABI 7): the multi-map flex composite and the frame call's two halves validate before anything is launched
    maps = (_lib.FlexMap * 9)()
    for m in maps:
        m.n_channel = 3
    assert lib.pnr_composite_rays_flex_multi(u32(4), u32(1), f32(1e-4), None, None, None, None, None, ctypes.cast(maps, ctypes.c_void_p), u32(9), None) == -2   # more than PNR_FLEX_MAX_MAPS
    assert lib.pnr_composite_rays_flex_multi(u32(4), u32(1), f32(1e-4), None, None, None, None, None, None, u32(2), None) == -1                                  # no map array
    maps[0].n_channel = 129
    assert lib.pnr_composite_rays_flex_multi(u32(4), u32(1), f32(1e-4), None, None, None, None, None, ctypes.cast(maps, ctypes.c_void_p), u32(1), None) == -2   # CHECK_CHANNEL per map
    maps[0].n_channel = 3
    assert lib.pnr_composite_rays_flex_multi(u32(4), u32(1), f32(1e-4), None, None, None, None, None, ctypes.cast(maps, ctypes.c_void_p), u32(1), None) == -1   # map without buffers
    assert lib.pnr_composite_rays_flex_multi(u32(0), u32(1), f32(1e-4), None, None, None, None, None, ctypes.cast(maps, ctypes.c_void_p), u32(1), None) == 0    # no alive ray: a no-op
    maps[0].n_channel = 0
    assert lib.pnr_composite_rays_flex_multi(u32(4), u32(1), f32(1e-4), None, None, None, None, None, ctypes.cast(maps, ctypes.c_void_p), u32(1), None) == 0    # only empty maps: a no-op
    for fn in (lib.pnr_nerf_render_frame_submit, lib.pnr_nerf_render_frame_finish, lib.pnr_palette_render_frame_submit, lib.pnr_palette_render_frame_finish):
        assert fn(None, None) == -1
    assert b"16-byte" in lib.pnr_error_string(-4) and lib.pnr_abi_version() >= 7
    assert lib.pnr_march_rays(u32(0), u32(4), *([None] * 4), f32(2), f32(0), u32(1024), u32(2), u32(128), *([None] * 8)) == 0
    # round 4 entry points: the layout flag and the self-filling march validate before they launch
    assert lib.pnr_grid_encode_forward_layout(None, None, None, None, u32(8), u32(3), u32(2), u32(16), f32(1.0), u32(16), None, u32(0), i32(0), i32(0), i32(2), None) == -1   # unknown layout
    assert lib.pnr_grid_encode_forward_layout(None, None, None, None, u32(0), u32(3), u32(2), u32(16), f32(1.0), u32(16), None, u32(0), i32(0), i32(0), i32(1), None) == 0    # empty batch
    assert lib.pnr_grid_encode_forward_layout(None, None, None, None, u32(8), u32(3), u32(2), u32(16), f32(1.0), u32(16), None, u32(0), i32(0), i32(0), i32(1), None) == -1   # null pointers
    assert lib.pnr_grid_encode_forward_layout(None, None, None, None, u32(8), u32(3), u32(2), u32(40), f32(1.0), u32(16), None, u32(0), i32(0), i32(0), i32(0), None) == -2   # > 32 levels
    # fill_rows smaller than the rows the march itself may write is a caller bug, not a silent overrun
    assert lib.pnr_march_rays_fill(u32(10), u32(4), *([None] * 4), f32(2), f32(0), u32(1024), u32(2), u32(128), *([None] * 8), u32(39), None) == -1
    assert lib.pnr_march_rays_fill(u32(0), u32(4), *([None] * 4), f32(2), f32(0), u32(1024), u32(2), u32(128), *([None] * 8), u32(0), None) == 0
    assert lib.pnr_set_option(b"grid_fast", 1) == 0 and lib.pnr_set_option(b"grid_nt", 0) == 0 and lib.pnr_set_option(b"no_such_option", 1) == -1


def test_palette_field_sizes_and_limits_without_gpu():
    """Shapes the fused PaletteNeRF field accepts (1..10 bases, clip heads up to 32 wide) and the sizes it reports, on the host."""
    from palettenerf_amd import _lib
    lib = _lib.load()
    u32, i32 = ctypes.c_uint32, ctypes.c_int
    tail = 2 * 768 + 256                       # the two 64 -> 3 heads as fp32 vectors (round 5: no longer 8 matrix blocks) + the palette / bias tables
    assert lib.pnr_palette_field_packed_bytes(u32(4), u32(0), i32(0)) == 42 * 2048 + tail        # 42 matrix blocks of 2 KiB
    assert lib.pnr_palette_field_packed_bytes(u32(8), u32(0), i32(0)) == 43 * 2048 + tail        # + the second offsets_radiance tile
    assert lib.pnr_palette_field_packed_bytes(u32(4), u32(16), i32(1)) == 51 * 2048 + tail       # + clip_net
    assert lib.pnr_palette_field_packed_bytes(u32(4), u32(32), i32(1)) == 55 * 2048 + tail       # + its second output tile
    assert lib.pnr_palette_aux_channels(u32(4), u32(16)) == 52 and lib.pnr_palette_aux_channels(u32(8), u32(16)) == 80
    assert lib.pnr_palette_field_stages_aux(u32(4), u32(0), i32(0)) == 1 and lib.pnr_palette_field_stages_aux(u32(10), u32(32), i32(1)) == 0
    a = _lib.PaletteFieldArgs()
    a.num_basis, a.clip_dim, a.precision, a.aux_stride = 11, 16, 1, 100
    assert lib.pnr_palette_field_forward(ctypes.byref(a), None) == -2                     # more than PNR_MAX_BASIS bases
    a.num_basis, a.precision = 4, 5
    assert lib.pnr_palette_field_forward(ctypes.byref(a), None) == -2                     # unknown precision
    a.precision, a.aux_stride = 0, 50
    assert lib.pnr_palette_field_forward(ctypes.byref(a), None) == -1                     # aux_stride not a multiple of 4
    p = _lib.PaletteFrameArgs()
    p.num_basis = 11
    assert lib.pnr_palette_render_frame(ctypes.byref(p), None) == -2


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "palettenerf_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "liborc" not in text and "pnr_oracle" not in text, f


def test_ops_fail_loudly_without_library(monkeypatch, tmp_path):
    from palettenerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_training_entry_points_validate_without_gpu():
    """The training-side entries added in round 1 (fused MLP, palette train shade, density, bias gradient) size and validate on the host."""
    from palettenerf_amd import _lib
    lib = _lib.load()
    u32, f32, i32, u64 = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_uint64

    def desc(dims, act=0):
        d = _lib.MlpDesc()
        d.n_layers = len(dims) - 1
        for i, v in enumerate(dims):
            d.dims[i] = v
        d.activation = act
        return d

    colour = desc([31, 64, 64, 3])
    # W and W^T slots: (2x1 + 2x2 + 1x2) tiles each way, 4 KiB per tile -- once as fp32 and once as split-fp16 pairs -- and the layers' inverse scales
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(colour)) == 2 * 2 * (2 + 4 + 2) * 4096 + 16
    assert lib.pnr_mlp_backward_workspace_bytes(ctypes.byref(colour), u32(627000)) == 256 * (31 * 64 + 64 * 64 + 64 * 3) * 4
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(desc([31, 65, 3]))) == 0            # a width above 64
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(desc([31, 64, 64, 3], act=_lib.MLP_OUT_SIGMOID))) == 2 * 2 * (2 + 4 + 2) * 4096 + 16   # + sigmoid on the output
    assert lib.pnr_mlp_backward(ctypes.byref(desc([31, 64, 64, 3], act=_lib.MLP_OUT_SIGMOID)), *[ctypes.c_void_p(8)] * 2, None, ctypes.c_void_p(8), u32(8),
                                *[None] * 4, ctypes.c_void_p(8), u64(1 << 30), None) == -1                                                # ... needs the forward's y
    assert lib.pnr_mlp_backward_lm(ctypes.byref(desc([32, 64, 16], act=_lib.MLP_OUT_SIGMOID)), ctypes.c_void_p(8), ctypes.c_void_p(8), u32(16), None,
                                   ctypes.c_void_p(8), u32(8), *[None] * 4, ctypes.c_void_p(8), u64(1 << 30), None) == -2
    bad = desc([31, 64, 3], act=2)
    assert lib.pnr_mlp_forward(ctypes.byref(bad), None, None, u32(8), None, None) == -2
    assert lib.pnr_mlp_forward(ctypes.byref(colour), None, None, u32(8), None, None) == -1      # null pointers
    assert lib.pnr_mlp_forward(ctypes.byref(colour), None, None, u32(0), None, None) == 0       # empty batch
    assert lib.pnr_mlp_forward_lm(ctypes.byref(colour), ctypes.c_void_p(8), ctypes.c_void_p(8), u32(16), None, u32(8), ctypes.c_void_p(8), None) == -2  # 31 < 32 columns
    assert lib.pnr_palette_train_shade_workspace_bytes(u32(4)) == 512 * 4 * 3 * 4
    assert lib.pnr_palette_train_shade_forward(u32(8), u32(17), u32(0), None, None, None, None, None, None, None, None, None, None) == -2   # nb > 16
    assert lib.pnr_palette_train_shade_forward(u32(0), u32(4), u32(16), None, None, None, None, None, None, None, None, None, None) == 0
    assert lib.pnr_palette_heads_forward(None, None, None, None, u32(8), u32(11), u32(15), None, None, None) == -2                          # nb > PNR_MAX_BASIS
    assert lib.pnr_palette_heads_forward(None, None, None, None, u32(8), u32(4), u32(17), None, None, None) == -2                           # in_dim > 16
    assert lib.pnr_palette_heads_forward(None, None, None, None, u32(0), u32(4), u32(15), None, None, None) == 0
    assert lib.pnr_palette_heads_forward(None, None, None, None, u32(8), u32(4), u32(15), None, None, None) == -1
    assert lib.pnr_palette_heads_backward(None, None, None, None, None, u32(8), u32(4), u32(15), None, None, None) == -1
    assert lib.pnr_nerf_density_forward(None, None, u32(8), f32(1.0), None, None, i32(7), f32(1.0), None) == -2                                      # precision
    assert lib.pnr_nerf_density_forward(None, None, u32(8), f32(1.0), None, None, i32(1), f32(1.0), None) == -1
    assert lib.pnr_linear_bgrad(None, i32(0), u32(0), u32(13), None, i32(1), None, u64(0), None) == -1                                      # no output
    assert lib.pnr_set_option(b"composite_fusion", 2) == 0
    assert lib.pnr_sh_encode_cat_forward(None, None, u32(15), None, u32(8), u32(8), None) == -2     # 64 + 15 columns: no room in the LDS tile
    assert lib.pnr_sh_encode_cat_forward(None, None, u32(0), None, u32(8), u32(4), None) == -2      # nothing to append
    assert lib.pnr_sh_encode_cat_forward(None, None, u32(15), None, u32(8), u32(4), None) == -1     # null pointers
    assert lib.pnr_sh_encode_cat_forward(None, None, u32(15), None, u32(0), u32(4), None) == 0
    # the round-3 switches of the frame loops (speed only): names exist, values are clamped, unknown names are refused
    for name, value in ((b"hosted_tail", 1), (b"march_budget", 2), (b"march_budget0", 0), (b"march_blocks", 0), (b"coop_march", 1),
                        (b"train_coop", 1), (b"mlp_f16x3", 1), (b"coarse_image", 1), (b"cell_merge", 1)):      # (round 4)
        assert lib.pnr_set_option(name, value) == 0
    assert lib.pnr_set_option(b"no_such_switch", 1) != 0 and lib.pnr_set_option(None, 1) != 0
    assert lib.pnr_abi_version() >= 5
    # pnr_palette_field_args grew at its end (frame-loop-only ray-state pointers): the ctypes mirror has them and leaves them NULL
    from palettenerf_amd import _lib as L
    names = [f[0] for f in L.PaletteFieldArgs._fields_]
    assert names[-6:] == ["rays_t", "weights_sum_rw", "depth", "image", "rays_alive_rw", "counts_cur"] and L.PaletteFieldArgs().rays_t is None


def test_ctypes_mirrors_have_the_layout_gcc_gives_the_header_structs(tmp_path):
    """Every argument struct of include/pnr.h against its ctypes mirror (palettenerf_amd/_lib.py): size and the offset of EVERY field, as gcc lays the
    header's struct out -- a field added on one side only (or in another place) would shift what the library reads without any error."""
    import subprocess
    from palettenerf_amd import _lib
    pairs = {"pnr_adam_tensor": _lib.AdamTensor, "pnr_adam_scalars": _lib.AdamScalars, "pnr_mlp_desc": _lib.MlpDesc, "pnr_occupancy_args": _lib.OccupancyArgs,
             "pnr_nerf_frame_args": _lib.NerfFrameArgs, "pnr_palette_edit": _lib.PaletteEdit, "pnr_palette_frame_args": _lib.PaletteFrameArgs,
             "pnr_palette_weights": _lib.PaletteWeights, "pnr_palette_field_args": _lib.PaletteFieldArgs, "pnr_train_loss_args": _lib.TrainLossArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pnr.h"', 'int main(void) {']
    for cname, mirror in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in mirror._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])     # (a field name the header lacks fails to compile)
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, mirror in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(mirror), cname
        for fname, *_ in mirror._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(mirror, fname).offset, f"{cname}.{fname}"
    # and the header has no field the mirror lacks: the sizes agree (above) and every mirrored field sits where the header puts it, so a missing field
    # could only hide in tail padding -- count the members gcc sees against the mirror's
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pnr.h")).read(), flags=re.S)
    for cname, mirror in pairs.items():
        m = re.search(r"typedef struct(?: \w+)? \{([^}]*)\} " + cname + ";", hdr)
        assert m, cname
        body = re.sub(r"\[[^\]]*\]", "", m.group(1))
        n_members = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_members == len(mirror._fields_), (cname, n_members, len(mirror._fields_))
