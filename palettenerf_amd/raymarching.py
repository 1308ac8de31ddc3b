"""Operator API of the ray-marching path -- same callables, argument order, defaults and return
conventions as the reference's raymarching/raymarching.py (cited per function), backed by the
gfx950 kernels behind the C ABI (include/pnr.h).  PyTorch is used for device memory, streams and
autograd plumbing only.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _lib
from ._torch_glue import call, ptr, require, to_cuda

_u32, _f32 = ctypes.c_uint32, ctypes.c_float
_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


USE_MIP = True  # stage the any/all occupancy mip in LDS for the march kernels (bit-identical results, far fewer global probes)
_MIP_ATTR = "_pnr_occupancy_mip"


def occupancy_mip(bitfield, C, H, bound):
    """Cached any/all brick mip of a density bitfield (pnr_build_occupancy_mip).  The cache entry lives ON the bitfield tensor object
    (an attribute), so it dies with the tensor: a later tensor that happens to get the same address from the caching allocator is a
    different object and starts without a mip.  The entry is keyed on the tensor's version counter (torch in-place ops, load_state_dict
    and this module's packbits() bump it), storage pointer and geometry; code that rewrites the bitfield behind torch's back (through
    `.data` or a raw pointer) must call invalidate_occupancy_mip(bitfield)."""
    if not USE_MIP or H % 4 != 0 or bitfield.data_ptr() % 8 != 0:
        return None
    nbytes = int(_lib.load().pnr_occupancy_mip_bytes(int(C), int(H)))
    if nbytes > 64 * 1024:
        return None
    key = (bitfield._version, bitfield.data_ptr(), str(bitfield.device), int(C), int(H), float(bound), bitfield.numel())
    ent = getattr(bitfield, _MIP_ATTR, None)
    if ent is not None and ent[0] == key:
        return ent[1]
    mip = torch.empty(nbytes // 4, dtype=torch.int32, device=bitfield.device)
    call("pnr_build_occupancy_mip", ptr(require(bitfield, torch.uint8, "density_bitfield")), _u32(C), _u32(H), _f32(bound), ptr(mip))
    setattr(bitfield, _MIP_ATTR, (key, mip))
    return mip


def note_bitfield_written(bitfield, C, H, bound, mip=None):
    """A raw kernel rewrote `bitfield` (the occupancy sweep): bump its version counter, which retires every mip cached for the old
    contents, and -- when the writer also rebuilt the mip -- install that one for the new contents."""
    torch.autograd.graph.increment_version(bitfield)
    if mip is not None:
        key = (bitfield._version, bitfield.data_ptr(), str(bitfield.device), int(C), int(H), float(bound), bitfield.numel())
        setattr(bitfield, _MIP_ATTR, (key, mip))


def invalidate_occupancy_mip(bitfield=None):
    """Drop the cached mip of `bitfield` (only needed after writes that bypass torch's version counter).  Without an argument this is a
    no-op kept for callers of the former process-wide cache."""
    if bitfield is not None and hasattr(bitfield, _MIP_ATTR):
        delattr(bitfield, _MIP_ATTR)


def _scratch(n, device):
    nbytes = int(_lib.load().pnr_scan_scratch_bytes(int(n)))
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)


# ---------------------------------------------------------------------------- utils
class _near_far_from_aabb(Function):
    """raymarching/raymarching.py:19-49"""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o = to_cuda(rays_o).contiguous().view(-1, 3)
        rays_d = to_cuda(rays_d).contiguous().view(-1, 3)
        aabb = to_cuda(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        call("pnr_near_far_from_aabb", ptr(require(rays_o, torch.float32, "rays_o")), ptr(require(rays_d, torch.float32, "rays_d")),
             ptr(require(aabb, torch.float32, "aabb")), _u32(N), _f32(min_near), ptr(nears), ptr(fars))
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    """raymarching/raymarching.py:52-80"""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        rays_o = to_cuda(rays_o).contiguous().view(-1, 3)
        rays_d = to_cuda(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        call("pnr_sph_from_ray", ptr(require(rays_o, torch.float32, "rays_o")), ptr(require(rays_d, torch.float32, "rays_d")),
             _f32(radius), _u32(N), ptr(coords))
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    """raymarching/raymarching.py:83-104"""

    @staticmethod
    def forward(ctx, coords):
        coords = to_cuda(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        call("pnr_morton3d", ptr(coords), _u32(N), ptr(indices))
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    """raymarching/raymarching.py:106-126"""

    @staticmethod
    def forward(ctx, indices):
        indices = to_cuda(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        call("pnr_morton3d_invert", ptr(indices), _u32(N), ptr(coords))
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    """raymarching/raymarching.py:129-155"""

    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        grid = to_cuda(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        call("pnr_packbits", ptr(require(grid, torch.float32, "grid")), _u32(N), _f32(thresh), ptr(require(bitfield, torch.uint8, "bitfield")))
        torch.autograd.graph.increment_version(bitfield)  # written by a raw kernel: invalidates the cached occupancy mip
        return bitfield


packbits = _packbits.apply


# ---------------------------------------------------------------------------- training
T_STORE_MAX = 1 << 26   # floats (256 MB): beyond that the training march walks twice as the reference does


class _march_rays_train(Function):
    """raymarching/raymarching.py:161-235.  Same outputs; sample offsets and the row order of `rays`
    are deterministic (prefix sum: row n == ray n) instead of atomics-ordered."""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o = to_cuda(rays_o).contiguous().view(-1, 3)
        rays_d = to_cuda(rays_d).contiguous().view(-1, 3)
        density_bitfield = to_cuda(density_bitfield).contiguous()
        dev = rays_o.device
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        sliced = force_all_rays or mean_count <= 0
        # The reference zero-fills all M = N * max_steps rows (134 MB for 4096 rays) and then keeps the first counter[0] of them
        # (raymarching.py:205-207, 223-229).  Only rows the caller can see need to be defined: with the slice, the rows in front of the
        # samples (a counter that did not start at zero) and the alignment tail are cleared below, everything else is written by the kernel;
        # without the slice (mean_count > 0) all M rows are returned, so they are zero-filled as in the reference.
        alloc = torch.empty if sliced else torch.zeros
        xyzs = alloc(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = alloc(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = alloc(M, 2, dtype=rays_o.dtype, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        if perturb:
            noises = torch.rand(N, dtype=rays_o.dtype, device=dev)
        else:
            noises = torch.zeros(N, dtype=rays_o.dtype, device=dev)
        scratch = _scratch(N, dev)
        mip = occupancy_mip(density_bitfield, C, H, bound)
        start = step_counter[:1].clone() if sliced else None   # where this call's rows begin (read back together with the new counter)
        # the counting pass keeps every sample's ray parameter (N * max_steps floats): the rows are then written without a second walk
        t_store = torch.empty(N * max_steps, dtype=torch.float32, device=dev) if N * max_steps <= T_STORE_MAX else None

        def launch():
            call("pnr_march_rays_train_mip", ptr(require(rays_o, torch.float32, "rays_o")), ptr(require(rays_d, torch.float32, "rays_d")),
                 ptr(require(density_bitfield, torch.uint8, "density_bitfield")), _f32(bound), _f32(dt_gamma), _u32(max_steps), _u32(N),
                 _u32(C), _u32(H), _u32(M), ptr(require(nears, torch.float32, "nears")), ptr(require(fars, torch.float32, "fars")),
                 ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(require(step_counter, torch.int32, "step_counter")), ptr(noises),
                 ptr(scratch), ptr(mip), ptr(t_store))

        launch()
        if sliced:
            first, m = torch.cat([start, step_counter[:1]]).tolist()     # ONE host read (the reference's `.item()`, raymarching.py:224)
            if m > M:   # rays were dropped for lack of room (only possible when the counter did not start at zero): their rows must read as zeros
                step_counter[0] = first
                step_counter[1] -= N
                for buf in (xyzs, dirs, deltas):
                    buf.zero_()
                launch()
            m_al = m + (align - m % align) if align > 0 else m
            for buf in (xyzs, dirs, deltas):
                if first > 0:
                    buf[:min(first, M)].zero_()
                if m_al > m:
                    buf[min(m, M):min(m_al, M)].zero_()
            xyzs, dirs, deltas = xyzs[:m_al], dirs[:m_al], deltas[:m_al]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    """raymarching/raymarching.py:238-291"""

    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        call("pnr_composite_rays_train_forward", ptr(require(sigmas, torch.float32, "sigmas")), ptr(require(rgbs, torch.float32, "rgbs")),
             ptr(require(deltas, torch.float32, "deltas")), ptr(require(rays, torch.int32, "rays")), _u32(M), _u32(N), _f32(T_thresh),
             ptr(weights_sum), ptr(depth), ptr(image))
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        ctx.set_materialize_grads(False)   # depth's gradient is never read: no zero tensor is made for it
        return weights_sum, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is ignored, as in the reference (raymarching.py:275)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        if grad_weights_sum is None and grad_image is None:
            return None, None, None, None, None
        grad_weights_sum = torch.zeros_like(weights_sum) if grad_weights_sum is None else grad_weights_sum.contiguous()
        grad_image = torch.zeros_like(image) if grad_image is None else grad_image.contiguous()
        grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        call("pnr_composite_rays_train_backward", ptr(require(grad_weights_sum, torch.float32, "grad_weights_sum")),
             ptr(require(grad_image, torch.float32, "grad_image")), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(weights_sum),
             ptr(image), _u32(M), _u32(N), _f32(T_thresh), ptr(grad_sigmas), ptr(grad_rgbs))
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


class _composite_rays_flex_train(Function):
    """raymarching/raymarching.py:294-341"""

    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, input, deltas, rays, T_thresh=1e-4):
        sigmas, input, deltas = sigmas.contiguous(), input.contiguous(), deltas.contiguous()
        M, N, n_channel = sigmas.shape[0], rays.shape[0], input.shape[-1]
        output = torch.empty(N, n_channel, dtype=sigmas.dtype, device=sigmas.device)
        call("pnr_composite_rays_flex_train_forward", ptr(require(sigmas, torch.float32, "sigmas")),
             ptr(require(input, torch.float32, "input")), ptr(require(deltas, torch.float32, "deltas")),
             ptr(require(rays, torch.int32, "rays")), _u32(M), _u32(N), _u32(n_channel), _f32(T_thresh), ptr(output))
        ctx.save_for_backward(sigmas, input, deltas, rays, output)
        ctx.dims = [M, N, n_channel, T_thresh]
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        sigmas, input, deltas, rays, output = ctx.saved_tensors
        M, N, n_channel, T_thresh = ctx.dims
        grad_input = torch.zeros_like(input)
        call("pnr_composite_rays_flex_train_backward", ptr(require(grad_output, torch.float32, "grad_output")), ptr(sigmas), ptr(input),
             ptr(deltas), ptr(rays), ptr(output), _u32(M), _u32(N), _u32(n_channel), _f32(T_thresh), ptr(grad_input))
        return None, grad_input, None, None, None


composite_rays_flex_train = _composite_rays_flex_train.apply


# ---------------------------------------------------------------------------- inference
class _march_rays(Function):
    """raymarching/raymarching.py:347-398 (including the always-pad alignment, :381-382)"""

    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                perturb=False, dt_gamma=0, max_steps=1024):
        rays_o = to_cuda(rays_o).contiguous().view(-1, 3)
        rays_d = to_cuda(rays_d).contiguous().view(-1, 3)
        dev = rays_o.device
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        # the reference zero-fills the three buffers (raymarching.py:384-386: three launches per call); here the march kernel clears the slots
        # it does not fill and the alignment rows itself (pnr_march_rays_fill): same contents, one launch
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = torch.empty(M, 2, dtype=rays_o.dtype, device=dev)
        noises = torch.rand(n_alive, dtype=rays_o.dtype, device=dev) if perturb else None  # NULL = zeros (no perturbation)
        mip = occupancy_mip(density_bitfield, C, H, bound)
        call("pnr_march_rays_fill", _u32(n_alive), _u32(n_step), ptr(require(rays_alive, torch.int32, "rays_alive")),
             ptr(require(rays_t, torch.float32, "rays_t")), ptr(require(rays_o, torch.float32, "rays_o")),
             ptr(require(rays_d, torch.float32, "rays_d")), _f32(bound), _f32(dt_gamma), _u32(max_steps), _u32(C), _u32(H),
             ptr(require(density_bitfield, torch.uint8, "density_bitfield")), ptr(require(near, torch.float32, "near")),
             ptr(require(far, torch.float32, "far")), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), ptr(mip), _u32(M))
        return xyzs, dirs, deltas


_march_rays_now = _march_rays.apply


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False, dt_gamma=0, max_steps=1024):
    """raymarching/raymarching.py:347-398."""
    if _flex_queues.q.shared is not None or _flex_queues.q.armed:
        _flex_queues.q.flush()
    return _march_rays_now(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align, perturb, dt_gamma, max_steps)


class _composite_rays(Function):
    """raymarching/raymarching.py:401-423 (in place; returns an empty tuple)"""

    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        call("pnr_composite_rays", _u32(n_alive), _u32(n_step), _f32(T_thresh), ptr(require(rays_alive, torch.int32, "rays_alive")),
             ptr(require(rays_t, torch.float32, "rays_t")), ptr(require(sigmas, torch.float32, "sigmas")),
             ptr(require(rgbs, torch.float32, "rgbs")), ptr(require(deltas, torch.float32, "deltas")),
             ptr(require(weights_sum, torch.float32, "weights_sum")), ptr(require(depth, torch.float32, "depth")),
             ptr(require(image, torch.float32, "image")))
        return tuple()


_composite_rays_now = _composite_rays.apply


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching/raymarching.py:401-423 (in place; returns an empty tuple).  Queued flex composites (defer_flex_composites) are issued first: this call is the
    writer of what they read."""
    if _flex_queues.q.shared is not None or _flex_queues.q.armed:
        _flex_queues.q.flush()
    return _composite_rays_now(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)


class _composite_rays_flex(Function):
    """raymarching/raymarching.py:425-447 (in place on `output`)"""

    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh=1e-2):
        sigmas, input, deltas = sigmas.contiguous(), input.contiguous(), deltas.contiguous()
        call("pnr_composite_rays_flex", _u32(n_alive), _u32(n_step), _u32(n_channel), _f32(T_thresh),
             ptr(require(rays_alive, torch.int32, "rays_alive")), ptr(require(rays_t, torch.float32, "rays_t")),
             ptr(require(sigmas, torch.float32, "sigmas")), ptr(require(input, torch.float32, "input")),
             ptr(require(deltas, torch.float32, "deltas")), ptr(require(weights_sum, torch.float32, "weights_sum")),
             ptr(require(output, torch.float32, "output")))
        return tuple()


_composite_rays_flex_now = _composite_rays_flex.apply


def composite_rays_flex_multi(n_alive, n_step, rays_alive, rays_t, sigmas, deltas, weights_sum, maps, T_thresh=1e-2):
    """SURVEY 8(b)'s multi-map variant: every (n_channel, input, output) of `maps` composited by ONE launch (pnr_composite_rays_flex_multi) -- what the
    six / seven `composite_rays_flex` calls of one march iteration of PaletteRenderer.run_cuda compute (palette/renderer.py:508-516), each output bit for
    bit the single call's.  More than PNR_FLEX_MAX_MAPS maps are issued in groups."""
    sigmas, deltas = require(sigmas.contiguous(), torch.float32, "sigmas"), require(deltas.contiguous(), torch.float32, "deltas")
    require(rays_alive, torch.int32, "rays_alive"), require(weights_sum, torch.float32, "weights_sum")
    keep = []
    for g in range(0, len(maps), _lib.FLEX_MAX_MAPS):
        group = maps[g:g + _lib.FLEX_MAX_MAPS]
        arr = (_lib.FlexMap * len(group))()
        for i, (n_channel, inp, out) in enumerate(group):
            inp = require(inp.contiguous(), torch.float32, "input")
            keep.append(inp)
            arr[i].n_channel, arr[i].input, arr[i].output = int(n_channel), inp.data_ptr(), require(out, torch.float32, "output").data_ptr()
        call("pnr_composite_rays_flex_multi", int(n_alive), int(n_step), float(T_thresh), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(deltas), ptr(weights_sum),
             ctypes.cast(arr, ctypes.c_void_p), len(group))
    return tuple()


class _FlexQueue:
    """Opt-in deferral of composite_rays_flex (dropin.fuse_field switches it on; defer_flex_composites(False) off).  The reference's PaletteNeRF loop calls
    composite_rays_flex six or seven times per march iteration and THEN composite_rays (palette/renderer.py:508-519).  composite_rays_flex reads sigmas / input /
    deltas / rays_alive / weights_sum and writes only its own `output` (raymarching.cu:1114-1185), so the calls of an iteration commute with each other and may be
    issued at any point before the next writer of what they read -- composite_rays (weights_sum, rays_alive).  With deferral on, a call is queued (the tensors are
    held) and the queue is issued as ONE pnr_composite_rays_flex_multi launch in front of the next composite_rays / march_rays / compact_alive of this module, or
    by flush_flex_composites() -- call that before reading a flex output that no composite_rays follows.  Calls whose shared arguments differ from the queue's
    flush it first.  Results are bit for bit those of the immediate calls."""

    def __init__(self):
        self.on = False
        self.armed = False     # one-shot: on until the next flush (dropin.fuse_field arms it behind every fused PaletteNetwork.forward of an inference iteration)
        self.shared = None     # (n_alive, n_step, rays_alive, rays_t, sigmas, deltas, weights_sum, T_thresh)
        self.maps = []

    def push(self, n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh):
        sh = self.shared
        if sh is not None and not (sh[0] == n_alive and sh[1] == n_step and sh[2] is rays_alive and sh[4] is sigmas and sh[5] is deltas and sh[6] is weights_sum
                                   and sh[7] == T_thresh):
            self.flush()
        if self.shared is None:
            self.shared = (n_alive, n_step, rays_alive, rays_t, sigmas, deltas, weights_sum, T_thresh)
        self.maps.append((n_channel, input, output))

    def flush(self):
        self.armed = False
        if self.shared is None:
            return
        (n_alive, n_step, rays_alive, rays_t, sigmas, deltas, weights_sum, T_thresh), maps = self.shared, self.maps
        self.shared, self.maps = None, []
        if len(maps) == 1:
            _composite_rays_flex_now(n_alive, n_step, maps[0][0], rays_alive, rays_t, sigmas, maps[0][1], deltas, weights_sum, maps[0][2], T_thresh)
        else:
            composite_rays_flex_multi(n_alive, n_step, rays_alive, rays_t, sigmas, deltas, weights_sum, maps, T_thresh)


class _FlexQueues(__import__("threading").local):
    """One queue per host thread: a thread's deferred composites must go out on that thread's stream, in front of that thread's next composite_rays."""

    def __init__(self):
        self.q = _FlexQueue()


_flex_queues = _FlexQueues()


def defer_flex_composites(on=True):
    """Switch the deferral of composite_rays_flex on or off (_FlexQueue explains); returns the previous setting."""
    was = _flex_queues.q.on
    if not on:
        _flex_queues.q.flush()
    _flex_queues.q.on = bool(on)
    return was


def flush_flex_composites():
    _flex_queues.q.flush()


def arm_flex_deferral():
    """Defer the composite_rays_flex calls from here up to the next composite_rays / march_rays / compact_alive / flush_flex_composites (one march iteration of
    the reference's loop), then fall back to immediate calls."""
    _flex_queues.q.armed = True


def composite_rays_flex(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh=1e-2):
    """raymarching/raymarching.py:425-447 (in place on `output`; returns an empty tuple)."""
    if (_flex_queues.q.on or _flex_queues.q.armed) and not torch.is_autocast_enabled() and sigmas.dtype == torch.float32 and input.dtype == torch.float32:
        _flex_queues.q.push(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh)
        return tuple()
    return _composite_rays_flex_now(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh)


class _spread_ray_to_sample(Function):
    """raymarching/raymarching.py:451-473"""

    @staticmethod
    @_fwd32
    def forward(ctx, input, rays, output):
        input = input.contiguous()
        N, M, n_channel = input.shape[0], output.shape[0], input.shape[-1]
        call("pnr_spread_ray_to_sample", ptr(require(input, torch.float32, "input")), ptr(require(rays, torch.int32, "rays")), _u32(M),
             _u32(N), _u32(n_channel), ptr(require(output, torch.float32, "output")))
        return tuple()


spread_ray_to_sample = _spread_ray_to_sample.apply


# ---------------------------------------------------------------------------- MI355X-first additions
def compact_alive(rays_alive, n_alive=None, out=None, count=None):
    """Device-side, order-preserving replacement of `rays_alive[rays_alive >= 0]`
    (nerf/renderer.py:376).  Returns (compacted ids buffer, device int32[1] count); no host sync."""
    if _flex_queues.q.shared is not None or _flex_queues.q.armed:
        _flex_queues.q.flush()
    n = rays_alive.shape[0] if n_alive is None else n_alive
    if out is None:
        out = torch.empty_like(rays_alive)
    if count is None:
        count = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    scratch = _scratch(max(n, 1), rays_alive.device)
    call("pnr_compact_alive", _u32(n), ptr(require(rays_alive, torch.int32, "rays_alive")), ptr(out), ptr(count), ptr(scratch))
    return out, count
