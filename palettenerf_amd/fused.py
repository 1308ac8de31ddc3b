"""MI355X-first fused evaluation of the fields (no reference counterpart as one call): level-major
hash-grid lookup feeding the MFMA tiny-MLP kernel directly, no permute/copy, no per-layer launches."""
import ctypes
import math
import os
import threading

import numpy as np
import torch

from . import _lib
from ._torch_glue import call, ptr, require, stream_ptr

_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int


def grid_encode_raw(encoder, x01):
    """[B,3] in [0,1] -> raw level-major [L,B,C] fp32 encoder output (the layout the kernels produce)."""
    B = x01.shape[0]
    L, C = encoder.num_levels, encoder.level_dim
    out = torch.empty(L, B, C, device=x01.device, dtype=torch.float32)
    emb = encoder.embeddings.detach()
    call("pnr_grid_encode_forward", ptr(require(x01, torch.float32, "inputs")), ptr(require(emb, torch.float32, "embeddings")),
         ptr(encoder.offsets), ptr(out), _u32(B), _u32(encoder.input_dim), _u32(C), _u32(L), _f32(np.log2(encoder.per_level_scale)),
         _u32(encoder.base_resolution), None, _u32(encoder.gridtype_id), _int(int(encoder.align_corners)), _int(0), units=B)
    return out


def _no_pair_copy():
    return None


class _PairCopy:
    """The interleaved [rows][2][2] copy of two encoders' tables that the training pair lookup reads.  It lives ON the first encoder (attribute
    `_pnr_pair`: freed with the model, no global registry, no id() aliasing after a model is freed) and remembers its partner by weak reference.
    A half is re-copied when torch can tell its table changed (_pkey: Parameter identity, storage, version -- optimiser steps, load_state_dict,
    in-place ops).  Writes through `.data` move none of those: invalidate_fused_caches(model) drops the copy (load_state_dict and
    initialize_palette call it), and as a safety net every call leaves a sampled checksum of both tables (pnr_checksum, one tiny launch, read
    back asynchronously) that the NEXT call compares with the one taken when the copy was made -- a silent `.data` rewrite is therefore noticed
    one call late, the copy rebuilt and a warning raised; PNR_PARANOID_CACHE=1 rebuilds on every call."""

    def __init__(self, enc_b, ea):
        import weakref
        self.partner = weakref.ref(enc_b)
        self.keys = [None, None]
        self.table = torch.empty(ea.shape[0], 2, 2, dtype=torch.float32, device=ea.device)
        self.sum_dev = torch.zeros(2, dtype=torch.int64, device=ea.device)
        self.sum_host = torch.zeros(2, dtype=torch.int64).pin_memory()
        self.sum_event = None       # recorded behind the asynchronous read-back of the last call's checksums
        self.sum_ref = [None, None]  # checksums of the tables as they were when their halves were copied

    # a copy or a pickle of the encoder (copy.deepcopy(model), torch.save(model)) does not take the derived table along: the copy rebuilds its own
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_pair_copy, ())

    def matches(self, enc_b, ea):
        return self.partner() is enc_b and self.table.device == ea.device and self.table.shape[0] == ea.shape[0]

    def lagged_check(self):
        """Look at the previous call's checksums (complete by now: a whole training step lies between two calls); a half whose table was rewritten
        behind torch's counters is marked stale."""
        if self.sum_event is None:
            return
        self.sum_event.synchronize()
        now = self.sum_host.tolist()
        self.sum_event = None
        for i in (0, 1):
            if self.sum_ref[i] is None:
                self.sum_ref[i] = now[i]
            elif self.sum_ref[i] != now[i]:
                import warnings
                warnings.warn("pair lookup: a hash table was rewritten behind torch's version counters (a `.data` write) -- the interleaved copy "
                              "was stale for one call; call invalidate_fused_caches(model) after such writes")
                self.keys[i] = None
                self.sum_ref[i] = None

    def refresh(self, tabs):
        with torch.no_grad():       # only the half whose table has changed (PaletteNeRF training: the density table is frozen)
            for i, t in enumerate(tabs):
                k = _pkey(t)
                if self.keys[i] != k or PARANOID:
                    self.table[:, i].copy_(t.detach())
                    self.keys[i] = k
                    self.sum_ref[i] = None      # the checksum this call takes becomes the reference
        n = 2
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tabs])
        sizes = (ctypes.c_uint64 * n)(*[t.numel() * t.element_size() for t in tabs])
        strides = (_u32 * n)(*[TABLE_CHECK_STRIDE] * n)
        call("pnr_checksum", ptrs, sizes, strides, _u32(n), ptr(self.sum_dev))
        self.sum_host.copy_(self.sum_dev, non_blocking=True)
        self.sum_event = torch.cuda.Event()
        self.sum_event.record()


def drop_pair_copies(model):
    """Forget the training pair lookup's interleaved table copies held by the model's encoders (see _PairCopy)."""
    for name in ("encoder", "encoder_palette", "encoder_clip"):
        enc = getattr(model, name, None)
        if enc is not None and "_pnr_pair" in getattr(enc, "__dict__", {}):
            del enc.__dict__["_pnr_pair"]


def grid_encode_raw_pair(enc_a, enc_b, x01):
    """Two hash-grid encoders of the same geometry at the same points in one launch (pnr_grid_encode_forward_pair): returns their two raw
    level-major [L,B,2] outputs, each bit for bit what grid_encode_raw gives.  The tables are read from an interleaved copy [rows][2][2] that is
    rebuilt (one strided copy, ~30 us for 2 x 49 MB) whenever either parameter has changed since -- every optimiser step in training, where the pair
    still costs ~1.3 lookups + that copy instead of 2 lookups.  The copy's life and staleness rules: _PairCopy."""
    ea, eb = enc_a.embeddings, enc_b.embeddings
    hit = enc_a.__dict__.get("_pnr_pair")
    if hit is None or not hit.matches(enc_b, ea):
        hit = _PairCopy(enc_b, ea)
        enc_a.__dict__["_pnr_pair"] = hit      # (__dict__: a plain attribute, not a registered buffer -- it must stay out of state_dict)
    hit.lagged_check()
    hit.refresh((ea, eb))
    pair = hit.table
    B = x01.shape[0]
    L = enc_a.num_levels
    out0 = torch.empty(L, B, 2, device=x01.device, dtype=torch.float32)
    out1 = torch.empty(L, B, 2, device=x01.device, dtype=torch.float32)
    call("pnr_grid_encode_forward_pair", ptr(require(x01, torch.float32, "inputs")), ptr(pair), ptr(enc_a.offsets), ptr(out0), ptr(out1), _u32(B), _u32(L),
         _f32(np.log2(enc_a.per_level_scale)), _u32(enc_a.base_resolution), _u32(enc_a.gridtype_id), _int(int(enc_a.align_corners)), units=B)
    return out0, out1


def pairable(enc_a, enc_b):
    """Same geometry (offsets, levels, scales), fp32 tables with two features per level on one device: what the pair lookup needs."""
    try:
        return (enc_a.num_levels == enc_b.num_levels and enc_a.level_dim == 2 and enc_b.level_dim == 2 and enc_a.input_dim == 3 and enc_b.input_dim == 3
                and enc_a.per_level_scale == enc_b.per_level_scale and enc_a.base_resolution == enc_b.base_resolution and enc_a.gridtype_id == enc_b.gridtype_id
                and enc_a.align_corners == enc_b.align_corners and enc_a.embeddings.shape == enc_b.embeddings.shape
                and enc_a.embeddings.dtype == torch.float32 and enc_b.embeddings.dtype == torch.float32 and enc_a.embeddings.device == enc_b.embeddings.device
                and enc_a.embeddings.is_cuda)
    except AttributeError:
        return False


def tile_ray_order(pixel_index, W, tile=8):
    """Permutation that visits rays tile by tile (tile x tile pixels, row-major inside a tile): one wave = one 8x8 tile.
    pixel_index: int64 [N] row-major pixel id of every ray (arange(H*W) for a full frame, the shard's ids otherwise)."""
    y, x = pixel_index // W, pixel_index % W
    if tile == 0:                                   # full Z-order over pixels
        def spread(v):
            v = (v | (v << 8)) & 0x00FF00FF
            v = (v | (v << 4)) & 0x0F0F0F0F
            v = (v | (v << 2)) & 0x33333333
            return (v | (v << 1)) & 0x55555555
        return torch.argsort(spread(x) | (spread(y) << 1)).to(torch.int32)
    key = ((y // tile) * ((W + tile - 1) // tile) + (x // tile)) * (tile * tile) + (y % tile) * tile + (x % tile)
    return torch.argsort(key).to(torch.int32)


PARANOID = os.environ.get("PNR_PARANOID_CACHE") == "1"   # rebuild every derived blob on every use (debugging aid for code that writes through `.data`)


_held_keys = threading.local()     # .memo: {id(tensor): key} while a frame / field call holds its answers (_PrecisionGuard._held), else absent / None


def _pkey(p):
    """Identity of a parameter's CURRENT value as far as torch can tell: the Parameter object, its storage and its version counter.
    In-place torch ops, optimizer steps and load_state_dict bump the version; replacing the Parameter changes the identity.  Writes through
    `p.data` (torch_ema's copy_to / restore, `p.data.uniform_()`) change neither: call invalidate_fused_caches(model) after those.
    Inside a held call a tensor's key is formed once (the source watch, the guard and the packer each ask for the same twenty tensors)."""
    memo = getattr(_held_keys, "memo", None)
    if memo is None:
        return (id(p), p.data_ptr(), p._version)
    k = memo.get(id(p))
    if k is None:
        k = memo[id(p)] = (id(p), p.data_ptr(), p._version)
    return k


def _weight_of(module, *path):
    """module.<path[0]>[path[1]]....weight through the modules' own dictionaries: what `m.sigma_net[0].weight` returns, without nn.Module.__getattr__'s
    fallback chain and ModuleList's index arithmetic per step (a frame asks for a dozen weights; the walk was a tenth of its host time)."""
    for name in path:
        module = module._modules[str(name)]
    return module._parameters["weight"]


def invalidate_fused_caches(model):
    """Forget every blob derived from the model's parameters (packed MFMA weights, interleaved / half tables, host-side parameter copies).
    Called by load_state_dict and initialize_palette; call it yourself after writing parameters through `.data` (EMA swap-in / restore:
    nerf/utils.py:829-839, 959-961).  Whole-tensor `.data` rewrites are also caught by the per-frame checksum (_SourceWatch); a write to only
    PART of a hash table is not (tables are checksummed on every 1 021st word): after such a write this call is required."""
    for attr in ("_fused", "_density_fused"):
        f = getattr(model, attr, None)
        if f is not None:
            f.invalidate_caches()
    drop_pair_copies(model)
    for ref in list(model.__dict__.get("_fused_twins", [])):      # handles made by pipeline.clone_for_concurrent_frames: same weights, own blobs
        twin = ref()
        if twin is not None and twin is not model:
            invalidate_fused_caches(twin)


def _half_copy(owner, attr, t):
    """fp16 copy of a table, cached on `owner` until the table changes (identity / data pointer / version)."""
    key = _pkey(t)
    cached = getattr(owner, attr, None)
    if cached is None or cached[0] != key or PARANOID:
        cached = (key, t.to(torch.float16).contiguous())
        setattr(owner, attr, cached)
    return cached[1]



F16X3_SAFE_ACTIVATION = 3.0e4   # fp16 overflows at 65 504: above this bound on any split operand the field runs on the exact fp32 path
F16X3_MAX_WEIGHT = 6.0e4        # the weights are split into fp16 halves at pack time: one beyond this cannot be represented at all


def _l1(w):
    """max_j sum_i |W[j][i]|: the factor by which a dense layer can grow the largest |activation|."""
    return w.detach().abs().sum(dim=1).max()


def _prescale_of(table_max):
    """Power of two that lifts a table's largest |entry| into [0.5, 1) when it is below 1/8 (else 1): below that the `lo` halves of the
    split features start to fall into fp16's subnormal range (the reference initialises its tables U(-1e-4, 1e-4), gridencoder/grid.py:107)."""
    if not math.isfinite(table_max) or table_max <= 0.0 or table_max >= 0.125:
        return 1.0
    return float(2.0 ** min(14, math.floor(-math.log2(table_max))))


class _PrecisionGuard:
    """What the split-fp16 matrix path needs to be safe for ANY weights: operands are split into two fp16 halves, so
    (a) very small encoder features lose their low half to the subnormal range -> they are pre-multiplied by a power of two inside the kernel
        (enc_scale, exact: the first stack is bias-free and positively homogeneous; undone on its outputs);
    (b) an activation beyond fp16's range becomes inf -> a static bound on every split operand (largest table entry times the layers' L1 row
        norms) is computed when the blob is (re)packed; if it can exceed F16X3_SAFE_ACTIVATION the call runs on the exact fp32 path instead.
    `precision` is what the caller asked for; `effective_precision()` is what runs.  A handful of tiny device reductions and ONE host read per
    repack (never per frame)."""

    def _guard_tables(self):
        raise NotImplementedError

    def _guard_weights(self):
        return self._w()

    def _guard_bound(self, tmax, scales):
        raise NotImplementedError

    def _held(self):
        """Context manager for one call of the stand-alone ops: the guard is evaluated once and its answer held until the call returns (effective_precision,
        enc_scales and _pack each ask; every answer walks the model's parameters -- 36 % of the host time of a drop-in frame with dropin.fuse_field)."""
        import contextlib

        @contextlib.contextmanager
        def hold():
            self._guard_hold = True
            outer = getattr(_held_keys, "memo", None)
            if outer is None:
                _held_keys.memo = {}
            try:
                yield
            finally:
                self._guard_hold = False
                self._guard_held = None
                self._weights_held = None
                if outer is None:
                    _held_keys.memo = None
        return hold()

    def _w(self):
        """self._weights(), looked up once per held call (a dozen nn.Module attribute walks, asked for by the guard, the packer and the source watch)."""
        held = self.__dict__.get("_weights_held")
        if held is not None:
            return held
        ws = self._weights()
        if self.__dict__.get("_guard_hold"):
            self._weights_held = ws
        return ws

    def _guard(self):
        held = self.__dict__.get("_guard_held")
        if held is not None:
            return held
        state = self._guard_now()
        if self.__dict__.get("_guard_hold"):
            self._guard_held = state
        return state

    def _guard_now(self):
        tables = self._guard_tables()
        key = tuple(_pkey(w) for w in self._guard_weights()) + tuple(_pkey(t) for t in tables)
        if getattr(self, "_guard_key", None) != key or PARANOID:
            tmax = [float(v) for v in torch.stack([t.detach().abs().max().float() for t in tables]).cpu().tolist()]
            scales = [_prescale_of(v) for v in tmax]
            bound = float(self._guard_bound(tmax, scales))
            wmax = float(torch.stack([w.detach().abs().max().float() for w in self._guard_weights()]).max())   # the weights are split into fp16 halves too
            self._guard_state = (scales + [1.0] * (3 - len(scales)), bound, wmax)
            self._guard_key = key
        return self._guard_state

    def enc_scales(self, prec=None):
        """[s_encoder, s_encoder_palette, s_encoder_clip] for the split-fp16 path (all 1.0 on the fp32 path)."""
        prec = self.effective_precision() if prec is None else prec
        return self._guard()[0] if prec in (1, 2) else [1.0, 1.0, 1.0]

    def _fp16_level(self):
        """1 = PNR_FIELD_F16X3 (split operands), 2 = PNR_FIELD_F16X2 (opt-in, NeRF field only: activations rounded once to fp16)."""
        return 2 if int(self.precision) == 2 and getattr(self, "supports_f16x2", False) else 1

    def effective_precision(self):
        """Matrix path of the stand-alone ops: an fp16 form only when the static bound rules an fp16 overflow out."""
        if int(self.precision) == 0:
            return 0
        return self._fp16_level() if self._guard()[1] < F16X3_SAFE_ACTIVATION and self._guard()[2] < F16X3_MAX_WEIGHT else 0

    def frame_precision(self):
        """(precision, watch) for the device-driven frame loops.  The static bound is a guarantee but pessimistic (products of L1 norms): trained
        weights often fail it without ever coming near fp16's range.  The frame loops therefore keep split-fp16 in that case and let the
        kernels WATCH the operands they split (watch = True, ~1 VALU per operand); a frame that reports an overflow is rendered again in exact
        fp32, and these weights stay on fp32 from then on (render_frame)."""
        if int(self.precision) == 0 or not self._guard()[2] < F16X3_MAX_WEIGHT:   # a weight itself beyond fp16's range: nothing to watch for
            return 0, False
        if self._guard()[1] < F16X3_SAFE_ACTIVATION:
            return self._fp16_level(), False
        if getattr(self, "_overflowed_key", None) == self._guard_key:
            return 0, False
        return self._fp16_level(), True

    def _note_overflow(self):
        import warnings
        self._overflowed_key = self._guard_key
        warnings.warn("fused field: an activation left fp16's range (> 65504) in the split-fp16 matrix path: the frame is rendered again on the exact "
                      "fp32 path, which these weights keep from now on")


TABLE_CHECK_STRIDE = 1 if PARANOID else 1021   # hash tables are checksummed on every 1021st word per frame (12 k scattered words of a 50 MB table: ~3 us; every 61st cost 18 us); weights in full


class _SourceWatch:
    """Blobs derived from parameters go stale SILENTLY when the parameters are rewritten through `.data` (torch_ema's copy_to / restore around
    an evaluation, nerf/utils.py:829-839, 959-961): neither identity nor version moves, so the keys above cannot tell.  The frame loops
    therefore checksum the sources on the device once per frame (pnr_checksum: one small launch, 8 bytes per source back with the frame's own
    read-back) and compare with the checksums taken when the blobs were built; on a mismatch the blobs are rebuilt and the frame is rendered
    again.  When torch CAN tell (a key changed) every blob of the object is rebuilt in that frame and its checksums become the reference, so a
    `.data` write can never hide behind an unrelated version bump.  Tables are sampled (TABLE_CHECK_STRIDE): a swap or a re-initialisation
    touches every row and is caught.  LIMITATION: a `.data` write that touches only PART of a table (loading or pruning some levels, a partial
    re-initialisation) will usually miss the sampled words and is NOT detected -- call invalidate_fused_caches(model) after such a write, or
    run with PNR_PARANOID_CACHE=1, which checks every word."""

    def _watched(self):
        """[(tensor, stride in 4-byte words)]: everything a cached blob of this object is derived from."""
        raise NotImplementedError

    def _watch_begin(self, slot=0):
        """slot: 0 / 1 -- a frame prepared while the previous one is still running (frame_prepare / frame_launch / frame_finish) takes the other read-back row."""
        srcs = self._watched()
        keys = tuple(_pkey(t) for t, _ in srcs)
        had = getattr(self, "_watch_keys", None)
        record = had != keys or getattr(self, "_watch_ref", None) is None
        if had is not None and had != keys:
            self.invalidate_caches()
        self._watch_keys = keys
        n, dev = len(srcs), srcs[0][0].device
        if getattr(self, "_watch_dev", None) is None or self._watch_dev.device != dev or self._watch_dev.shape[1] < n:
            self._watch_dev = torch.zeros(2, max(n, 24), dtype=torch.int64, device=dev)
            self._watch_host = torch.zeros(2, max(n, 24), dtype=torch.int64).pin_memory()
            self._watch_rows = [(self._watch_dev[k], self._watch_host[k]) for k in (0, 1)]     # (row views made once: a view per frame is host time)
        if getattr(self, "_watch_args_key", None) != keys:     # (the keys hold the data pointers: same keys, same argument arrays)
            self._watch_args = ((ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in srcs]), (ctypes.c_uint64 * n)(*[t.numel() * t.element_size() for t, _ in srcs]),
                                (_u32 * n)(*[int(st) for _, st in srcs]))
            self._watch_args_key = keys
        ptrs, sizes, strides = self._watch_args
        drow, hrow = self._watch_rows[slot]
        call("pnr_checksum", ptrs, sizes, strides, _u32(n), ptr(drow))
        hrow.copy_(drow, non_blocking=True)     # complete once the frame's own read-back is (same stream)
        return record, n, slot

    def _watch_end(self, state):
        """True: the frame just rendered used blobs that match their sources."""
        record, n, slot = state
        now = tuple(self._watch_rows[slot][1][:n].tolist())
        if record:
            self._watch_ref = now
            return True
        return now == self._watch_ref

    def _watch_failed(self):
        import warnings
        warnings.warn("fused field: parameters were rewritten behind torch's version counters (a `.data` write); the packed blobs were rebuilt and "
                      "the frame rendered again -- invalidate_fused_caches(model) after such writes avoids the double render")
        self.invalidate_caches()


def _set_near_far(a, nears, fars, aabb, min_near, N, dev):
    """nears / fars of a frame-args struct: given, or (aabb set, both None) left to the frame call's first launch (pnr_nerf_frame_args::aabb)."""
    a.aabb = None
    if nears is None or fars is None:
        if aabb is None:
            raise RuntimeError("render_frame needs nears / fars or the box they follow from (aabb)")
        if not (aabb.is_cuda and aabb.dtype == torch.float32 and aabb.is_contiguous() and aabb.numel() == 6):
            raise RuntimeError("aabb must be 6 contiguous fp32 values on the device")
        nears = torch.empty(N, dtype=torch.float32, device=dev)
        fars = torch.empty(N, dtype=torch.float32, device=dev)
        a.aabb, a.min_near = aabb.data_ptr(), float(min_near)
    a.nears, a.fars = nears.data_ptr(), fars.data_ptr()
    return nears, fars


def _set_finish(a, bg_color, N, mask):
    """Fill the finish / bg fields of a frame-args struct; returns True when the call will apply the epilogue."""
    a.finish, a.bg_map = 0, None
    if bg_color is None:
        return False
    if torch.is_tensor(bg_color):
        if bg_color.numel() == 3:
            vals = [float(v) for v in bg_color.flatten().tolist()]
        elif bg_color.shape[-1] == 3 and bg_color.numel() == 3 * N and bg_color.is_cuda and bg_color.dtype == torch.float32 and bg_color.is_contiguous():
            a.bg_map = bg_color.data_ptr()
            vals = [0.0, 0.0, 0.0]
        else:
            return False
    elif isinstance(bg_color, (int, float)):
        vals = [float(bg_color)] * 3
    else:
        vals = [float(v) for v in bg_color]
        if len(vals) != 3:
            return False
    for k in range(3):
        a.bg_color[k] = vals[k]
    a.finish = mask
    return True


class StaleFrame(RuntimeError):
    """frame_launch of a frame that was prepared before the object's blobs were rebuilt (invalidate_caches): prepare it again."""


class _FrameToken:
    """A frame between frame_prepare and frame_finish: its argument struct, outputs, host-side stats arrays and what a re-render needs."""
    __slots__ = ("a", "p", "out", "stats", "kms", "watch_state", "watch", "finished", "nears", "fars", "again", "keep", "stream", "depth_raw", "edit", "valid", "gen")

    def __init__(self, **kw):
        self.valid = None
        for k, v in kw.items():
            setattr(self, k, v)


def _frame_valid(fused, tok):
    """0: the finished frame's outputs stand; 1: its blobs did not match their sources (a `.data` write); 2: an operand left fp16's range under the watch."""
    if not fused._watch_end(tok.watch_state):
        return 1
    return 2 if (tok.watch and tok.stats[5]) else 0


class NeRFFieldFused(_PrecisionGuard, _SourceWatch):
    """Caches the MFMA-ordered weight blob of a NeRFNetwork and evaluates (sigma, rgb) for sample batches."""

    def __init__(self, model):
        self.model = model
        self.packed = None
        self.versions = None
        self.precision = 1  # PNR_FIELD_F16X3 (split-fp16 matrix path, ~2^-22 relative); 0 = PNR_FIELD_FP32 (exact fmaf chains); 2 = PNR_FIELD_F16X2 (opt-in: ~1e-5 on a colour)
        self.supports_f16x2 = True
        self.time_grid_kernel = False  # bench.py: HIP-event timing of the grid-encode launches inside the native frame loop
        self.table_half = False        # native loop: look the hash table up as fp16 with the reference's half interpolation (its --fp16 mode)
        m = model
        ok = (m.encoder.num_levels == 16 and m.encoder.level_dim == 2 and m.encoder.input_dim == 3 and m.hidden_dim == 64 and m.geo_feat_dim == 15
              and m.num_layers == 2 and m.num_layers_color == 3 and m.hidden_dim_color == 64 and m.encoder_dir.degree == 4)
        if not ok:
            raise RuntimeError("fused NeRF field kernel is specialised for the shipped architecture (hashgrid 16x2, 64-wide nets, SH degree 4)")

    def _weights(self):
        m = self.model
        return [_weight_of(m, "sigma_net", 0), _weight_of(m, "sigma_net", 1), _weight_of(m, "color_net", 0), _weight_of(m, "color_net", 1), _weight_of(m, "color_net", 2)]

    def invalidate_caches(self):
        self.versions = None
        self._emb_half = None
        self._guard_key = None
        self._watch_ref = None       # the next frame rebuilds every blob: its source checksums become the reference
        self._guard_held = self._weights_held = None   # (a held call that retries asks again)
        self._frame_plan = None      # the frame call's kept argument struct points into the blobs
        self._gen = self.__dict__.get("_gen", 0) + 1     # frames prepared before this point hold pointers into the old blobs (frame_launch refuses them)

    def _guard_tables(self):
        return [self.model.encoder.embeddings]

    def _watched(self):
        return [(w, 1) for w in self._w()] + [(self.model.encoder.embeddings, TABLE_CHECK_STRIDE)]

    def _guard_bound(self, tmax, scales):
        ws = self._weights()
        l1 = [float(v) for v in torch.stack([_l1(w).float() for w in ws]).cpu().tolist()]
        enc = tmax[0] * scales[0]
        h1 = l1[0] * enc                                   # sigma_net hidden (in prescaled units)
        geo = l1[1] * h1 / scales[0]                       # its 16 outputs, descaled
        c0 = l1[2] * max(1.6, geo)                         # colour net input = [SH (|Y| < 1.6 up to degree 4) ; geo]
        c1 = l1[3] * c0
        return max(enc, h1, geo, c0, c1)

    def _pack(self, prec=None):
        ws = self._w()
        prec = self.effective_precision() if prec is None else prec
        versions = tuple(_pkey(w) for w in ws) + (prec,)
        if self.packed is None or versions != self.versions or PARANOID:
            dev = ws[0].device
            if self.packed is None or self.packed.device != dev:
                self.packed = torch.empty(int(_lib.load().pnr_nerf_field_packed_bytes()) // 4, dtype=torch.float32, device=dev)
            ws = [require(w.detach().contiguous(), torch.float32, "weight") for w in ws]
            call("pnr_nerf_field_pack", *[ptr(w) for w in ws], ptr(self.packed), _int(prec))
            self.versions = versions
        return self.packed

    @torch.no_grad()
    def render_frame(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color=None, aabb=None, min_near=0.0):
        """One inference frame through the device-driven loop (pnr_nerf_render_frame).  Returns
        (weights_sum [N], depth [N], image [N,3], stats dict).  bg_color None: raw accumulations (bg mix and depth normalisation are
        the caller's); a number, 3 numbers or an [N,3] tensor: the call also applies run_cuda's epilogue (image + (1 - ws) bg,
        normalised depth) -- stats['finished'] says so.  aabb (device tensor of 6 floats) with nears = fars = None: the call computes
        near / far in its first launch (near_far_from_aabb's arithmetic); stats['nears'], stats['fars'] hold them."""
        with self._held():
            return self._render_frame(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)

    # The same frame in three steps (round 6; pnr_nerf_render_frame_submit / _finish): frame_prepare does everything in front of the library call (outputs, argument
    # struct, source checksums) and may run while the PREVIOUS frame is still on the device; frame_launch enqueues the frame and returns at once; frame_finish waits
    # for it and returns what render_frame returns.  A caller with a queue of frames (a video path, a rank's shard loop) runs
    #     tok = prepare(0); launch(tok);  for i in 1..: nxt = prepare(i); out = finish(tok); launch(nxt); tok = nxt; consume(out)
    # so that the host's work for frame i + 1 lies under frame i's kernels.  Two argument structs alternate (a frame's struct must stay as it was until its finish).
    @torch.no_grad()
    def frame_prepare(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color=None, aabb=None, min_near=0.0):
        with self._held():
            return self._frame_prepare(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)

    def frame_launch(self, tok):
        if tok.gen != self.__dict__.get("_gen", 0):
            raise StaleFrame("the frame was prepared before the packed blobs were rebuilt")
        tok.stream = stream_ptr()
        _lib.check(_lib.load().pnr_nerf_render_frame_submit(ctypes.byref(tok.a), tok.stream), "pnr_nerf_render_frame_submit")
        return tok

    def frame_wait(self, tok):
        """The library's finish call alone: waits for the frame (and enqueues what it still needs).  True: the frame's outputs are valid -- the caller may enqueue
        its next frame before it asks for frame_result(); False: the frame has to be rendered again (sources rewritten behind torch's counters, an fp16
        overflow), which frame_result() does -- ask for it BEFORE launching another frame."""
        _lib.check(_lib.load().pnr_nerf_render_frame_finish(ctypes.byref(tok.a), tok.stream), "pnr_nerf_render_frame_finish")
        tok.valid = _frame_valid(self, tok)
        return tok.valid == 0

    @torch.no_grad()
    def frame_result(self, tok):
        with self._held():
            return self._frame_post(tok)

    def frame_finish(self, tok):
        self.frame_wait(tok)
        return self.frame_result(tok)

    def _render_frame(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near):
        tok = self._frame_prepare(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)
        rc = _lib.load().pnr_nerf_render_frame(ctypes.byref(tok.a), stream_ptr())
        _lib.check(rc, "pnr_nerf_render_frame")
        return self._frame_post(tok)

    def _frame_prepare(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near):
        from . import raymarching
        m = self.model
        N = rays_o.shape[0]
        dev = rays_o.device
        lib = _lib.load()
        slot = self._slot = 1 - self.__dict__.get("_slot", 1)
        watch_state = self._watch_begin(slot)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        order = getattr(self, "ray_order", None)
        if order is not None and order.numel() != N:
            order = None
        enc = m.encoder
        # Everything of the argument struct that only changes when a parameter, a table or a setting does is filled once and kept (the "plan"): the key holds
        # the identity and version of every source (the watch's keys, formed above), the frame size and the settings read here.  A frame whose key matches
        # sets the per-frame fields only -- a third of the call's host time in front of the first launch went into re-deriving the same fifty values.
        plan_key = (self._watch_keys, N, dev, int(self.precision), bool(self.table_half), self.__dict__.get("_overflowed_key"), None if order is None else (id(order), order.data_ptr()),
                    float(m.bound), int(m.cascade), int(m.grid_size), float(m.density_scale), enc.num_levels, enc.per_level_scale, enc.base_resolution, enc.gridtype_id,
                    _pkey(enc.offsets))
        plans = self.__dict__.get("_frame_plan")
        if plans is None:
            plans = self._frame_plan = {}
        plan = plans.get(slot)
        if plan is None or plan[0] != plan_key or PARANOID:
            nbytes = int(lib.pnr_nerf_frame_workspace_bytes(N))
            if getattr(self, "_ws", None) is None or self._ws.numel() < nbytes or self._ws.device != dev:
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            emb = require(enc.embeddings.detach(), torch.float32, "embeddings")
            if self.table_half:   # the reference's --fp16 tables (`embeddings.to(torch.half)` per forward, gridencoder/grid.py:38): converted once per update here
                emb = _half_copy(self, "_emb_half", enc.embeddings)
            prec, watch = self.frame_precision()
            a = _lib.NerfFrameArgs()
            a.table_dtype = 1 if self.table_half else 0
            a.N = N
            a.bound, a.C, a.H = float(m.bound), int(m.cascade), int(m.grid_size)
            a.embeddings, a.offsets = emb.data_ptr(), enc.offsets.data_ptr()
            a.num_levels, a.S, a.base_resolution, a.gridtype = enc.num_levels, float(np.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id
            a.field_precision, a.watch_overflow = int(prec), int(watch)
            for k, v in enumerate(self.enc_scales(prec)):
                a.enc_scale[k] = v
            a.density_scale = float(m.density_scale)
            a.workspace, a.workspace_bytes = self._ws.data_ptr(), nbytes
            a.ray_order = order.data_ptr() if order is not None else None
            plan = plans[slot] = (plan_key, a, prec, watch, (emb, self._ws, order))     # (the tensors whose addresses the struct holds)
        a, prec, watch = plan[1], plan[2], plan[3]
        # The packed blob is NOT part of the plan: the stand-alone ops on this object (`self(x, d)` from network.forward) repack it in place for THEIR precision
        # (effective_precision() is fp32 where frame_precision() keeps split-fp16 with a watch).  _pack is a key compare when nothing changed.
        a.packed_weights = self._pack(prec).data_ptr()
        mip = raymarching.occupancy_mip(m.density_bitfield, m.cascade, m.grid_size, m.bound)
        stats = (ctypes.c_uint64 * 6)()
        a.rays_o, a.rays_d = rays_o.data_ptr(), rays_d.data_ptr()
        nears, fars = _set_near_far(a, nears, fars, aabb, min_near, N, dev)
        a.bitfield = m.density_bitfield.data_ptr()
        a.mip = mip.data_ptr() if mip is not None else None
        a.dt_gamma, a.max_steps, a.T_thresh = float(dt_gamma), int(max_steps), float(T_thresh)
        a.weights_sum, a.depth, a.image = ws.data_ptr(), depth.data_ptr(), image.data_ptr()
        a.stats = ctypes.cast(stats, ctypes.c_void_p)
        kms = (ctypes.c_float * 2)()
        a.kernel_ms = ctypes.cast(kms, ctypes.c_void_p) if self.time_grid_kernel else None
        finished = _set_finish(a, bg_color, N, 3)
        for t, name in ((rays_o, "rays_o"), (rays_d, "rays_d"), (nears, "nears"), (fars, "fars")):
            require(t, torch.float32, name)
        return _FrameToken(gen=self.__dict__.get("_gen", 0), a=a, out=(ws, depth, image), stats=stats, kms=kms, watch_state=watch_state, watch=watch, finished=finished, nears=nears, fars=fars,
                           again=(rays_o, rays_d, None if aabb is not None else nears, None if aabb is not None else fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near),
                           keep=(mip, bg_color))

    def _frame_post(self, tok):
        valid = tok.valid if tok.valid is not None else _frame_valid(self, tok)
        if valid == 1:
            self._watch_failed()
            return self._render_frame(*tok.again)
        stats, kms = tok.stats, tok.kms
        if valid == 2:
            self._note_overflow()
            return self._render_frame(*tok.again)
        ws, depth, image = tok.out
        return ws, depth, image, {"iterations": int(stats[0]), "rendered": int(stats[1]), "rows": int(stats[2]), "enqueued": int(stats[3]), "looks": int(stats[4]),
                                  "grid_ms": float(kms[0]), "grid_launches": int(kms[1]), "finished": tok.finished, "nears": tok.nears, "fars": tok.fars}

    @torch.no_grad()
    def __call__(self, x, d):
        m = self.model
        x01 = ((x + m.bound) / (2 * m.bound)).contiguous()  # same two roundings as GridEncoder.forward (gridencoder/grid.py:142)
        enc = grid_encode_raw(m.encoder, x01)
        B = x.shape[0]
        sigmas = torch.empty(B, dtype=torch.float32, device=x.device)
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=x.device)
        with self._held():
            call("pnr_nerf_field_forward", ptr(enc), ptr(require(d.contiguous(), torch.float32, "dirs")), ptr(self._pack()), _u32(B), ptr(sigmas),
                 ptr(rgbs), _int(self.effective_precision()), _f32(self.enc_scales()[0]), units=B)
        return sigmas, rgbs


class DensityFused(NeRFFieldFused):
    """sigma_net alone through the matrix cores (pnr_nerf_density_forward): sigma and the 15 geometry features of sample batches, no
    gradient.  Works for NeRFNetwork and PaletteNetwork (same encoder / sigma_net / color_net shapes; only the sigma_net part of the blob
    is read).  Users: the occupancy sweep (update_extra_state), density() under no_grad, PaletteNeRF training (geometry detached)."""

    def _guard_weights(self):   # only sigma_net runs here: colour-head updates (every training step) must not touch the guard or the blob
        m = self.model
        return [m.sigma_net[0].weight, m.sigma_net[1].weight]

    def _guard_bound(self, tmax, scales):
        l1 = [float(v) for v in torch.stack([_l1(w).float() for w in self._guard_weights()]).cpu().tolist()]
        enc = tmax[0] * scales[0]
        return max(enc, l1[0] * enc)

    def _pack(self, prec=None):
        prec = self.effective_precision() if prec is None else prec
        versions = tuple(_pkey(w) for w in self._guard_weights()) + (prec,)
        if self.packed is None or versions != self.versions or PARANOID:
            self.versions = None
            super()._pack(prec)
            self.versions = versions
        return self.packed

    @torch.no_grad()
    def __call__(self, x, scale=1.0, want_geo=True, enc=None):
        """enc: the encoder's raw level-major output at x when the caller has it already (the pair lookup of PaletteNeRF training)."""
        m = self.model
        if enc is None:
            x01 = ((x + m.bound) / (2 * m.bound)).contiguous()
            enc = grid_encode_raw(m.encoder, x01)
        B = x.shape[0]
        sigmas = torch.empty(B, dtype=torch.float32, device=x.device)
        geo = torch.empty(B, 15, dtype=torch.float32, device=x.device) if want_geo else None
        with self._held():
            call("pnr_nerf_density_forward", ptr(enc), ptr(self._pack()), _u32(B), ctypes.c_float(scale), ptr(sigmas), ptr(geo), _int(self.effective_precision()), _f32(self.enc_scales()[0]), units=B)
        return sigmas, geo


def density_fused(model):
    """The model's cached DensityFused (built on first use)."""
    d = getattr(model, "_density_fused", None)
    if d is None:
        d = model._density_fused = DensityFused(model)
    return d


class PaletteFieldFused(_PrecisionGuard, _SourceWatch):
    """Fused PaletteNeRF field + colour-basis composite (pnr_palette_field_forward).  Produces, per sample,
    sigma * density_scale, rgb and one packed aux row [direct 3 | view_dep 3 | omega nb | basis_rgb 3nb | unscaled 3nb | clip | pad]."""

    def __init__(self, model):
        self.model = model
        self.packed = None
        self.versions = None
        m = model
        ok = (m.encoder.num_levels == 16 and m.encoder.level_dim == 2 and m.hidden_dim == 64 and m.geo_feat_dim == 15 and m.num_layers == 2
              and m.num_layers_color == 3 and m.encoder_dir.degree == 4 and 1 <= m.num_basis <= _lib.MAX_BASIS and m.opt.clip_dim <= _lib.MAX_CLIP)
        if not ok:
            raise RuntimeError("fused palette field kernel is specialised for the shipped architecture")
        self.nb, self.pred_clip = int(m.num_basis), bool(m.opt.pred_clip)
        # without a clip head the reference composites clip_dim channels of zeros (palette/renderer.py:477,510): the map is zero whatever
        # happens, so those channels are left out of the packed aux row (52 -> 36 floats per sample for 4 bases) and returned as zeros
        self.clip_dim = int(m.opt.clip_dim) if self.pred_clip else 0
        self.precision = 1              # PNR_FIELD_F16X3; 0 = PNR_FIELD_FP32 (exact fmaf chains, pnr_palette_*'s fp32 matrix path); 2 = PNR_FIELD_F16X2 (opt-in)
        self.supports_f16x2 = True      # honoured by the 4-basis kernel without an edit head; the library runs every other shape as F16X3
        self.interleave_tables = True   # native loop: look both hash tables up through one interleaved copy (see _pair_table)
        self.table_half = False         # native loop: fp16 tables with the reference's half interpolation (its --fp16 mode; no clip head)
        self.aux_channels = int(_lib.load().pnr_palette_aux_channels(self.nb, self.clip_dim))

    def _weights(self):
        m = self.model
        ws = [_weight_of(m, *path) for path in (("sigma_net", 0), ("sigma_net", 1), ("diff_net", 0), ("diff_net", 1), ("diff_net", 2), ("color_net", 0), ("color_net", 1),
                                                ("color_net", 2), ("basis_net", 0), ("basis_net", 1), ("offsets_radiance_net",), ("omega_net", 0))]
        if self.pred_clip:
            ws += [_weight_of(m, "clip_net", 0), _weight_of(m, "clip_net", 1)]
        return ws

    def _tables(self):
        """The palette and the bias of offsets_radiance_net: packed into the blob next to the weights (the kernels read them from LDS)."""
        m = self.model
        return [m.basis_color, m.offsets_radiance_net.bias]

    def invalidate_caches(self):
        self.versions = None
        self._pair_key = self._triple_key = self._guard_key = None
        self._watch_ref = None       # the next frame rebuilds every blob: its source checksums become the reference
        self._guard_held = self._weights_held = None   # (a held call that retries asks again)
        self._frame_plan = None      # the frame call's kept argument struct points into the blobs
        self._gen = self.__dict__.get("_gen", 0) + 1     # frames prepared before this point hold pointers into the old blobs (frame_launch refuses them)

    def _watched(self):
        return [(w, 1) for w in self._w() + self._tables()] + [(t, TABLE_CHECK_STRIDE) for t in self._guard_tables()]

    def _guard_tables(self):
        m = self.model
        return [m.encoder.embeddings, m.encoder_palette.embeddings] + ([m.encoder_clip.embeddings] if self.pred_clip else [])

    def _guard_bound(self, tmax, scales):
        l1 = [float(v) for v in torch.stack([_l1(w).float() for w in self._weights()]).cpu().tolist()]
        s0, s1, d0, d1, d2, c0, c1, c2, b0, b1 = l1[:10]
        enc = tmax[0] * scales[0]
        h1 = s0 * enc
        geo = s1 * h1 / scales[0]
        dd0 = d0 * geo
        dd1 = d1 * dd0
        cc0 = c0 * max(1.6, geo)
        cc1 = c1 * cc0
        bin_ = max(tmax[1], 1.0) * scales[1]               # [enc_palette ; diffuse in (0,1)], prescaled together
        bb0 = b0 * bin_ / scales[1]                        # ELU output <= its input bound
        p = b1 * max(bb0, 1.0)
        sites = [enc, h1, geo, dd0, dd1, cc0, cc1, bin_, bb0, p]
        if self.pred_clip:
            ce = tmax[2] * scales[2]
            sites += [ce, l1[12] * ce]
        return max(sites)

    def _pack(self, prec=None):
        ws = self._w()
        prec = self.effective_precision() if prec is None else prec
        versions = tuple(_pkey(w) for w in ws + self._tables()) + (prec,)
        if self.packed is None or versions != self.versions or PARANOID:
            dev = ws[0].device
            lib = _lib.load()
            self.packed = torch.empty(int(lib.pnr_palette_field_packed_bytes(self.nb, self.clip_dim, int(self.pred_clip))), dtype=torch.uint8, device=dev)
            self._keep = [require(w.detach().contiguous(), torch.float32, "weight") for w in ws]
            pw = _lib.PaletteWeights()
            names = ["sigma0", "sigma1", "diff0", "diff1", "diff2", "color0", "color1", "color2", "basis0", "basis1", "offsets_radiance", "omega"]
            if self.pred_clip:
                names += ["clip0", "clip1"]
            for name, w in zip(names, self._keep):
                setattr(pw, name, w.data_ptr())
            self._keep_tables = [require(t.detach().float().contiguous(), torch.float32, "palette / bias") for t in self._tables()]
            pw.basis_color, pw.or_bias = self._keep_tables[0].data_ptr(), self._keep_tables[1].data_ptr()
            pw.num_basis, pw.clip_dim, pw.pred_clip, pw.precision = self.nb, self.clip_dim, int(self.pred_clip), int(prec)
            rc = lib.pnr_palette_field_pack(ctypes.byref(pw), ctypes.c_void_p(self.packed.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "pnr_palette_field_pack")
            self.versions = versions
        return self.packed

    def _edit_struct(self):
        """The model's RegionEdit / Stylizer state as a pnr_palette_edit (None when neither is set).  Rebuilt when the controllers' tensors
        change (object identity / version): a few small D2H copies per change, none per frame."""
        m = self.model
        edit, sty = getattr(m, "edit", None), getattr(m, "stylizer", None)
        if edit is None and sty is None:
            return None
        if sty is not None:
            key = ("s", id(sty), _pkey(sty.dI), _pkey(sty.dP), _pkey(sty.ddelta))
        else:
            tkey = lambda t: None if t is None or not torch.is_tensor(t) else (id(t), t.data_ptr(), t._version)
            key = ("e", id(edit), tkey(edit.delta_hsv), tkey(edit.mean_xyz), tkey(edit.mean_clip), float(edit.std_xyz), float(edit.std_clip), bool(edit.weight_mode))
        if getattr(self, "_edit_key", None) == key and not PARANOID:
            return self._edit
        e = _lib.PaletteEdit()
        nb = self.nb
        if sty is not None:    # palette/renderer.py:150-183
            e.mode = 2
            dI, dP, dd = sty.dI.detach().float().cpu(), sty.dP.detach().float().cpu().reshape(nb, 3), sty.ddelta.detach().float().cpu()
            for b in range(nb):
                e.dI[b] = float(dI[b])
                for k in range(3):
                    e.dP[b][k] = float(dP[b, k])
                    for j in range(3):
                        e.ddelta[b][k][j] = float(dd[b, k, j])
        else:                  # palette/renderer.py:84-147
            e.mode = 1
            dh = edit.delta_hsv.detach().float().cpu()
            for b in range(nb):
                for k in range(3):
                    e.delta_hsv[b][k] = float(dh[b, k])
            if edit.mean_xyz is not None:
                e.has_mean_xyz = 1
                for k, v in enumerate(edit.mean_xyz.detach().float().cpu().reshape(-1)[:3].tolist()):
                    e.mean_xyz[k] = v
            if edit.mean_clip is not None:
                mc = edit.mean_clip.detach().float().cpu().reshape(-1).tolist()
                if len(mc) != int(m.opt.clip_dim):
                    raise RuntimeError("RegionEdit.mean_clip must have clip_dim entries")
                e.has_mean_clip = len(mc)
                for k, v in enumerate(mc):
                    e.mean_clip[k] = v
            e.std_xyz, e.std_clip, e.weight_mode = float(edit.std_xyz), float(edit.std_clip), int(bool(edit.weight_mode))
        self._edit, self._edit_key = e, key
        return e

    @torch.no_grad()
    def _pair_table(self):
        """`encoder` and `encoder_palette` interleaved row by row ([rows, 4] fp32; one 16-byte gather then serves both lookups).
        A copy of both tables (2 x 50 MB for the shipped config), rebuilt when either changes."""
        m = self.model
        key = (_pkey(m.encoder.embeddings), _pkey(m.encoder_palette.embeddings), bool(self.table_half))
        a, b = m.encoder.embeddings.detach(), m.encoder_palette.embeddings.detach()
        if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape or a.shape[1] != 2:
            return None
        if getattr(self, "_pair_key", None) != key or PARANOID:
            if self.table_half:   # rows of 4 halves: (a.x, a.y, b.x, b.y)
                out = torch.cat([a.to(torch.float16), b.to(torch.float16)], dim=1).contiguous()
            else:
                out = torch.empty(a.shape[0], 4, dtype=torch.float32, device=a.device)
                call("pnr_interleave_tables", ptr(a.contiguous()), ptr(b.contiguous()), ctypes.c_uint64(a.shape[0]), ptr(out))
            self._pair, self._pair_key = out, key
        return self._pair

    @torch.no_grad()
    def _triple_table(self):
        """--pred_clip: the three tables interleaved row by row ([rows, 8] fp32 = encoder, encoder_palette, encoder_clip, 2 pad): one 32-byte
        row per corner serves all three lookups.  Rebuilt when any table changes."""
        m = self.model
        key = tuple(_pkey(e.embeddings) for e in (m.encoder, m.encoder_palette, m.encoder_clip))
        ts = [e.embeddings.detach() for e in (m.encoder, m.encoder_palette, m.encoder_clip)]
        if any(t.dtype != torch.float32 or t.shape != ts[0].shape or t.shape[1] != 2 for t in ts):
            return None
        if getattr(self, "_triple_key", None) != key or PARANOID:
            out = torch.empty(ts[0].shape[0], 8, dtype=torch.float32, device=ts[0].device)
            call("pnr_interleave_tables3", *[ptr(t.contiguous()) for t in ts], ctypes.c_uint64(ts[0].shape[0]), ptr(out))
            self._triple, self._triple_key = out, key
        return self._triple

    def render_frame(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color=None, aabb=None, min_near=0.0):
        """One PaletteNeRF inference frame through the device-driven loop (pnr_palette_render_frame).
        Returns (weights_sum [N], depth [N], image [N,3], aux_map [N, aux_channels], stats).  bg_color None: raw accumulations; a number, 3 numbers
        or an [N,3] tensor: the call's last launch also applies run_cuda's epilogue (palette/renderer.py:520-540: image and aux_map[:, 0:3] = direct_rgb
        blended with the background, depth normalised, the raw depth kept in stats['depth_raw']) -- stats['finished'] says so.  aabb with
        nears = fars = None: near / far computed by the call's first launch (stats['nears'], stats['fars'])."""
        with torch.no_grad(), self._held():
            return self._render_frame(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)

    # prepare / launch / finish: NeRFFieldFused explains (pnr_palette_render_frame_submit / _finish)
    def frame_prepare(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color=None, aabb=None, min_near=0.0):
        with torch.no_grad(), self._held():
            return self._frame_prepare(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)

    def frame_launch(self, tok):
        if tok.gen != self.__dict__.get("_gen", 0):
            raise StaleFrame("the frame was prepared before the packed blobs were rebuilt")
        tok.stream = stream_ptr()
        _lib.check(_lib.load().pnr_palette_render_frame_submit(ctypes.byref(tok.p), tok.stream), "pnr_palette_render_frame_submit")
        return tok

    def frame_wait(self, tok):
        _lib.check(_lib.load().pnr_palette_render_frame_finish(ctypes.byref(tok.p), tok.stream), "pnr_palette_render_frame_finish")
        tok.valid = _frame_valid(self, tok)
        return tok.valid == 0

    def frame_result(self, tok):
        with torch.no_grad(), self._held():
            return self._frame_post(tok)

    def frame_finish(self, tok):
        self.frame_wait(tok)
        return self.frame_result(tok)

    def _render_frame(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near):
        tok = self._frame_prepare(rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near)
        rc = _lib.load().pnr_palette_render_frame(ctypes.byref(tok.p), stream_ptr())
        _lib.check(rc, "pnr_palette_render_frame")
        return self._frame_post(tok)

    def _frame_prepare(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near):
        from . import raymarching
        m = self.model
        N = rays_o.shape[0]
        dev = rays_o.device
        lib = _lib.load()
        slot = self._slot = 1 - self.__dict__.get("_slot", 1)
        watch_state = self._watch_begin(slot)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        aux_map = torch.empty(N, self.aux_channels, dtype=torch.float32, device=dev)
        order = getattr(self, "ray_order", None)
        if order is not None and order.numel() != N:
            order = None
        enc = m.encoder
        # the kept argument struct (NeRFFieldFused._frame_prepare explains): key = identity and version of every source + frame size + the settings read below
        plan_key = (self._watch_keys, N, dev, int(self.precision), bool(self.table_half), bool(self.interleave_tables), self.__dict__.get("_overflowed_key"),
                    None if order is None else (id(order), order.data_ptr()), float(m.bound), int(m.cascade), int(m.grid_size), float(m.density_scale),
                    float(m.offsets_weight), float(m.view_dep_weight), enc.num_levels, enc.per_level_scale, enc.base_resolution, enc.gridtype_id,
                    tuple((_pkey(o.offsets), o.per_level_scale) for o in (enc, m.encoder_palette, m.encoder_clip)))
        plans = self.__dict__.get("_frame_plan")
        if plans is None:
            plans = self._frame_plan = {}
        plan = plans.get(slot)
        if plan is None or plan[0] != plan_key or PARANOID:
            nbytes = int(lib.pnr_palette_frame_workspace_bytes(N, self.nb, self.clip_dim, int(self.pred_clip)))
            if getattr(self, "_ws", None) is None or self._ws.numel() < nbytes or self._ws.device != dev:
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            # (once per set of offset buffers: torch.equal is a device comparison plus a host read)
            for other in (m.encoder_palette, m.encoder_clip):
                if not torch.equal(other.offsets, enc.offsets) or other.per_level_scale != enc.per_level_scale:
                    raise RuntimeError("the three hash grids must share one level layout")
            prec, watch = self.frame_precision()
            p = _lib.PaletteFrameArgs()
            a = p.base
            a.N = N
            a.bound, a.C, a.H = float(m.bound), int(m.cascade), int(m.grid_size)
            a.embeddings = require(enc.embeddings.detach(), torch.float32, "embeddings").data_ptr()
            a.offsets = enc.offsets.data_ptr()
            a.num_levels, a.S, a.base_resolution, a.gridtype = enc.num_levels, float(np.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id
            a.field_precision, a.watch_overflow = int(prec), int(watch)
            for k, v in enumerate(self.enc_scales(prec)):
                a.enc_scale[k] = v
            a.density_scale = float(m.density_scale)
            a.workspace, a.workspace_bytes = self._ws.data_ptr(), nbytes
            a.ray_order = order.data_ptr() if order is not None else None
            p.embeddings_palette = require(m.encoder_palette.embeddings.detach(), torch.float32, "embeddings").data_ptr()
            p.embeddings_clip = require(m.encoder_clip.embeddings.detach(), torch.float32, "embeddings").data_ptr() if self.pred_clip else None
            p.num_basis, p.clip_dim, p.pred_clip = self.nb, self.clip_dim, int(self.pred_clip)
            p.offsets_weight, p.view_dep_weight = float(m.offsets_weight), float(m.view_dep_weight)
            if self.table_half and self.pred_clip:
                raise RuntimeError("fp16 tables in the native PaletteNeRF loop need the interleaved pair table (no clip head)")
            pair = self._pair_table() if ((self.interleave_tables or self.table_half) and not self.pred_clip) else None
            p.embeddings_pair = pair.data_ptr() if pair is not None else None
            triple = self._triple_table() if (self.interleave_tables and self.pred_clip and not self.table_half) else None
            p.embeddings_triple = triple.data_ptr() if triple is not None else None
            a.table_dtype = 1 if self.table_half else 0
            plan = plans[slot] = (plan_key, p, prec, watch, (self._ws, order, pair, triple))     # (the tensors whose addresses the struct holds)
        p, prec, watch = plan[1], plan[2], plan[3]
        a = p.base
        a.packed_weights = self._pack(prec).data_ptr()     # every frame (NeRFFieldFused._frame_prepare explains): a stand-alone call may have packed ANOTHER tensor since
        mip = raymarching.occupancy_mip(m.density_bitfield, m.cascade, m.grid_size, m.bound)
        stats = (ctypes.c_uint64 * 6)()
        kms = (ctypes.c_float * 2)()
        a.rays_o, a.rays_d = rays_o.data_ptr(), rays_d.data_ptr()
        nears, fars = _set_near_far(a, nears, fars, aabb, min_near, N, dev)
        a.bitfield = m.density_bitfield.data_ptr()
        a.mip = mip.data_ptr() if mip is not None else None
        a.dt_gamma, a.max_steps, a.T_thresh = float(dt_gamma), int(max_steps), float(T_thresh)
        a.weights_sum, a.depth, a.image = ws.data_ptr(), depth.data_ptr(), image.data_ptr()
        a.stats = ctypes.cast(stats, ctypes.c_void_p)
        a.kernel_ms = ctypes.cast(kms, ctypes.c_void_p) if getattr(self, "time_grid_kernel", False) else None
        finished = _set_finish(a, bg_color, N, 7)
        depth_raw = torch.empty(N, dtype=torch.float32, device=dev) if finished else None
        a.depth_raw = depth_raw.data_ptr() if finished else None
        p.aux_map = aux_map.data_ptr()
        edit = self._edit_struct()
        p.edit = ctypes.cast(ctypes.pointer(edit), ctypes.c_void_p) if edit is not None else None
        for t, name in ((rays_o, "rays_o"), (rays_d, "rays_d"), (nears, "nears"), (fars, "fars")):
            require(t, torch.float32, name)
        return _FrameToken(gen=self.__dict__.get("_gen", 0), a=a, p=p, out=(ws, depth, image, aux_map), stats=stats, kms=kms, watch_state=watch_state, watch=watch, finished=finished, nears=nears, fars=fars,
                           depth_raw=depth_raw, edit=edit,
                           again=(rays_o, rays_d, None if aabb is not None else nears, None if aabb is not None else fars, dt_gamma, max_steps, T_thresh, bg_color, aabb, min_near),
                           keep=(mip, bg_color, self.packed))

    def _frame_post(self, tok):
        valid = tok.valid if tok.valid is not None else _frame_valid(self, tok)
        if valid == 1:
            self._watch_failed()
            return self._render_frame(*tok.again)
        stats, kms = tok.stats, tok.kms
        if valid == 2:
            self._note_overflow()
            return self._render_frame(*tok.again)
        ws, depth, image, aux_map = tok.out
        return ws, depth, image, aux_map, {"iterations": int(stats[0]), "rendered": int(stats[1]), "rows": int(stats[2]), "enqueued": int(stats[3]), "looks": int(stats[4]),
                                           "grid_ms": float(kms[0]), "grid_launches": int(kms[1]), "finished": tok.finished, "depth_raw": tok.depth_raw,
                                           "nears": tok.nears, "fars": tok.fars}

    @torch.no_grad()
    def network_forward(self, x, d):
        """PaletteNetwork.forward(x, d) (palette/network.py:156-190) as ONE launch behind the lookups: returns what the reference's method returns --
        (sigma [B] unscaled, clip_feat [B, clip_dim], omega [B, nb] normalised, offsets_radiance [B, 3 nb + 1], view_dep [B, 3], diffuse [B, 3]) --
        the last five as slices of one packed row (pnr_palette_edit.mode 3).  Inference only (no autograd graph)."""
        heads = _lib.PaletteEdit()
        heads.mode = 3
        sigmas, _, row = self(x, d, _edit=heads, _density_scale=1.0)
        nb, cd = self.nb, int(self.model.opt.clip_dim)
        omega = row[:, :nb]
        offsets_radiance = row[:, nb:4 * nb + 1]
        view_dep = row[:, 4 * nb + 1:4 * nb + 4]
        diffuse = row[:, 4 * nb + 4:4 * nb + 7]
        clip_feat = row[:, 4 * nb + 7:4 * nb + 7 + cd] if self.pred_clip else sigmas.new_zeros(sigmas.shape[0], cd)    # palette/network.py:179
        return sigmas, clip_feat, omega, offsets_radiance, view_dep, diffuse

    def __call__(self, x, d, deltas=None, _edit=None, _density_scale=None):
        """x [B,3] world positions, d [B,3] -> (sigmas [B] scaled by density_scale, rgbs [B,3], aux [B, aux_channels])."""
        m = self.model
        lib = _lib.load()
        B = x.shape[0]
        dev = x.device
        x01 = ((x + m.bound) / (2 * m.bound)).contiguous()
        pair = None
        if self.interleave_tables and not self.pred_clip and not self.table_half and pairable(m.encoder, m.encoder_palette):
            pair = self._pair_table()       # the native loop's interleaved copy (rebuilt when either table changes): one 16-byte gather serves both lookups
        if pair is not None:
            L = m.encoder.num_levels
            enc = torch.empty(L, B, 2, device=dev, dtype=torch.float32)
            enc_pal = torch.empty(L, B, 2, device=dev, dtype=torch.float32)
            e0 = m.encoder
            call("pnr_grid_encode_forward_pair", ptr(require(x01, torch.float32, "inputs")), ptr(pair), ptr(e0.offsets), ptr(enc), ptr(enc_pal), _u32(B), _u32(L),
                 _f32(np.log2(e0.per_level_scale)), _u32(e0.base_resolution), _u32(e0.gridtype_id), _int(int(e0.align_corners)), units=B)
        else:
            enc = grid_encode_raw(m.encoder, x01)
            enc_pal = grid_encode_raw(m.encoder_palette, x01)
        enc_clip = grid_encode_raw(m.encoder_clip, x01) if self.pred_clip else None
        sigmas = torch.empty(B, dtype=torch.float32, device=dev)
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=dev)
        aux = torch.empty(B, self.aux_channels, dtype=torch.float32, device=dev)
        a = _lib.PaletteFieldArgs()
        a.ctl, a.B = None, B
        a.enc, a.enc_palette = enc.data_ptr(), enc_pal.data_ptr()
        a.enc_clip = enc_clip.data_ptr() if enc_clip is not None else None
        a.level_stride = B
        a.dirs = require(d.contiguous(), torch.float32, "dirs").data_ptr()
        a.deltas = deltas.data_ptr() if deltas is not None else None
        a.packed = self._pack().data_ptr()
        a.num_basis, a.clip_dim, a.pred_clip = self.nb, self.clip_dim, int(self.pred_clip)
        a.density_scale, a.offsets_weight, a.view_dep_weight = float(m.density_scale if _density_scale is None else _density_scale), float(m.offsets_weight), float(m.view_dep_weight)
        a.aux_stride = self.aux_channels
        a.sigmas, a.rgbs, a.aux = sigmas.data_ptr(), rgbs.data_ptr(), aux.data_ptr()
        a.precision = int(self.effective_precision())
        if _edit is not None and a.precision == 2:
            a.precision = 1          # (the rounded-activation form exists for the composite kernels only)
        for k, v in enumerate(self.enc_scales()):
            a.enc_scale[k] = v
        edit = self._edit_struct() if _edit is None else _edit
        a.edit = ctypes.cast(ctypes.pointer(edit), ctypes.c_void_p) if edit is not None else None
        xw = require(x.contiguous(), torch.float32, "xyzs")
        a.xyzs = xw.data_ptr()
        rc = lib.pnr_palette_field_forward(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "pnr_palette_field_forward")
        return sigmas, rgbs, aux
