"""Host-side mirror of the reference's renderer classes for the occupancy-march path:
`NeRFRenderer.run_cuda` (nerf/renderer.py:258-393) and `PaletteRenderer.run_cuda`
(palette/renderer.py:296-552), with the same constructor arguments, buffers (state_dict names),
keyword arguments and result dictionaries.

Two execution modes for inference frames:
  * "compat"  -- the reference's host-driven loop, step for step (one host sync per iteration for
                 the boolean-mask compaction); used for parity against golden frames.
  * "device"  -- same arithmetic and the same n_step schedule, but the alive list is compacted on
                 the GPU (wave64 ballot + prefix sum) and n_alive only crosses to the host once per
                 iteration through a pinned 4-byte read-back of the already-computed count.
The pure-PyTorch (`cuda_ray=False`) renderer of the reference is dead code there
 and is not mirrored here.
"""
import ctypes
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, raymarching
from ._torch_glue import call, ptr, require
from .palette_utils import hsv_to_rgb, palette_train_shade, rgb_to_hsv
from .train_loss import RawTrain, TrainResults


def default_opt(**kw):
    """The subset of main_palette.py's argparse namespace the hot path reads (main_palette.py:16-101)."""
    opt = SimpleNamespace(num_basis=4, clip_dim=16, pred_clip=False, use_initialization_from_rgbxy=False, test=True,
                          color_space="srgb", smooth_sigma_xyz=0.005, smooth_sigma_color=0.2, smooth_sigma_clip=0.0)
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF sampling of the piecewise-constant density `weights` over `bins` (nerf/renderer.py:12-45, NeRF's hierarchical sampler):
    bins [B, T], weights [B, T-1] -> [B, n_samples].  det: stratified midpoints instead of uniform random numbers."""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    lead = list(cdf.shape[:-1])
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples).to(weights.device).expand(lead + [n_samples])
    else:
        u = torch.rand(lead + [n_samples]).to(weights.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    span = cdf_hi - cdf_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return bin_lo + (u - cdf_lo) / span * (bin_hi - bin_lo)


class _MarchState:
    """Per-frame state of the inference loop (nerf/renderer.py:344-350)."""

    def __init__(self, N, nears, device):
        self.N = N
        self.weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        self.depth = torch.zeros(N, dtype=torch.float32, device=device)
        self.image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        self.rays_alive = torch.arange(N, dtype=torch.int32, device=device)
        self.rays_t = nears.clone()
        self.n_samples = 0  # evaluated rows (incl. padding), host-side bookkeeping only
        self.rendered = torch.zeros(1, dtype=torch.int64, device=device)  # march-emitted samples (delta > 0), device-side


class _LazyScalar:
    """A number the device (or an in-flight copy) still owes the host: resolved -- one wait -- the first time somebody asks for it."""

    def __init__(self, fetch):
        self._fetch, self._value = fetch, None

    def get(self):
        if self._fetch is not None:
            self._value, self._fetch = self._fetch(), None
        return self._value


def _lazy_property(name):
    slot = "_lazy_" + name

    def get(self):
        v = self.__dict__.get(slot, 0)
        return v.get() if isinstance(v, _LazyScalar) else v

    def put(self, value):
        self.__dict__[slot] = value

    return property(get, put)


class _OccupancyMaintenance:
    """Producer of the density bitfield the march consumes: NeRFRenderer.mark_untrained_grid / update_extra_state
    (nerf/renderer.py:395-561) as calls into the device-resident sweep of csrc/occupancy.hip.  The reference's methods are host Python
    over torch ops (a `.item()` for the mean, `nonzero`, boolean-mask scatters); here the host only draws the random numbers (as the
    reference does, with torch's generator) and enqueues: no step waits for the device.  `mean_density` and `mean_count` stay plain
    numbers to their readers (the checkpoint writer, march_rays_train) but are fetched only when read."""

    mean_density = _lazy_property("mean_density")
    mean_count = _lazy_property("mean_count")
    occupancy_chunk = 1 << 20      # samples in flight per lookup / sigma_net launch pair of the fused sweep (144 B of workspace each)

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells no training camera sees (or that a camera sees closer than min_near) get density -1 and are never sampled.  Returns the
        number of cells marked as a 0-dim device tensor (the reference prints it)."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        poses = torch.as_tensor(poses).to(device=dev, dtype=torch.float32).contiguous()
        fx, fy, cx, cy = [float(v) for v in intrinsic]
        marked = torch.empty((), dtype=torch.int32, device=dev)
        call("pnr_mark_untrained_grid", ptr(poses), ctypes.c_uint32(poses.shape[0]), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx), ctypes.c_float(cy),
             ctypes.c_uint32(self.cascade), ctypes.c_uint32(self.grid_size), ctypes.c_float(self.bound), ctypes.c_float(self.min_near),
             ctypes.c_int(int(bool(getattr(self, "filter_close_point", False)))), ptr(require(self.density_grid, torch.float32, "density_grid")), ptr(marked))
        return marked

    def _fused_sweep_ok(self):
        """The shipped field (16 x 2 hash grid, fp32 table, sigma_net 32 -> 64 -> 16) evaluated by its own density(): the whole sweep is one
        C-ABI call.  Anything else -- another architecture, a density() replaced on the instance or overridden by a subclass -- goes through its density() between the
        point and scatter kernels."""
        enc, net = getattr(self, "encoder", None), getattr(self, "sigma_net", None)
        if "density" in self.__dict__ or enc is None or net is None or getattr(self, "occupancy_generic", False):
            return False
        # a subclass that overrides density() at class level (the reference always calls self.density(), nerf/renderer.py:499) must be asked:
        # the fused sweep is taken only when the method IS one this library declared fusable (network.py sets `_pnr_fused_density` on its own)
        if not getattr(getattr(type(self), "density", None), "_pnr_fused_density", False):
            return False
        emb = getattr(enc, "embeddings", None)
        return (emb is not None and emb.is_cuda and emb.dtype == torch.float32 and getattr(enc, "num_levels", 0) == 16 and getattr(enc, "level_dim", 0) == 2
                and getattr(enc, "input_dim", 0) == 3 and not getattr(enc, "align_corners", False) and len(net) == 2
                and tuple(net[0].weight.shape) == (64, 32) and tuple(net[1].weight.shape) == (16, 64) and net[0].weight.dtype == torch.float32)

    def _occupancy_workspace(self, chunk):
        need = int(_lib.load().pnr_occupancy_workspace_bytes(self.cascade, self.grid_size, int(chunk)))
        ws = self.__dict__.get("_occ_ws")
        if ws is None or ws.numel() < need or ws.device != self.density_grid.device:
            ws = self.__dict__["_occ_ws"] = torch.empty(need, dtype=torch.uint8, device=self.density_grid.device)
        return ws

    def _sigma_blob(self):
        """sigma_net as the exact-fp32 blob of the matrix-core kernels (the sigma_net part of the NeRF field blob; no colour weights are
        passed, so only that part is written).  Packed anew for every sweep -- one tiny launch -- so it cannot go stale whichever way the
        weights were written (optimiser kernels, `.data` swaps of an EMA)."""
        w0, w1 = self.sigma_net[0].weight, self.sigma_net[1].weight
        blob = self.__dict__.get("_occ_blob")
        if blob is None or blob.device != w0.device:
            blob = self.__dict__["_occ_blob"] = torch.empty(int(_lib.load().pnr_nerf_field_packed_bytes()) // 4, dtype=torch.float32, device=w0.device)
        a, b = require(w0.detach().contiguous(), torch.float32, "sigma_net.0"), require(w1.detach().contiguous(), torch.float32, "sigma_net.1")
        call("pnr_nerf_field_pack", ptr(a), ptr(b), None, None, None, ptr(blob), ctypes.c_int(0))
        return blob

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, noise=None, coords=None, occ_rand=None):
        """EMA-max update of density_grid from the field, repack the bitfield (+ its mip), refresh mean_density / mean_count
        (nerf/renderer.py:467-561).  noise / coords / occ_rand: the sweep's random numbers when the caller wants to supply them
        (include/pnr.h, pnr_occupancy_args); drawn here with torch's generator otherwise, as the reference draws them."""
        if not self.cuda_ray:
            return
        G, C = self.grid_size, self.cascade
        dev = self.density_grid.device
        # mean_count first: its source, the last <= 16 step counters, is complete once the training steps already enqueued have run -- the
        # copy is queued in FRONT of the sweep, so whoever reads mean_count next waits for those steps only, not for the sweep
        total_step = min(16, self.local_step)
        if total_step > 0:
            host = torch.empty(total_step, dtype=torch.int32).pin_memory()
            host.copy_(self.step_counter[:total_step, 0], non_blocking=True)
            done = torch.cuda.Event()
            done.record()

            def fetch(host=host, done=done, total_step=total_step):
                done.synchronize()
                return int(int(host.sum()) / total_step)
            self.mean_count = _LazyScalar(fetch)
        self.local_step = 0

        a = _lib.OccupancyArgs()
        a.C, a.H, a.bound = C, G, float(self.bound)
        grid = require(self.density_grid, torch.float32, "density_grid")
        bits = require(self.density_bitfield, torch.uint8, "density_bitfield")
        a.density_grid, a.density_bitfield = grid.data_ptr(), bits.data_ptr()
        a.density_scale, a.decay, a.density_thresh = float(self.density_scale), float(decay), float(self.density_thresh)
        if self.iter_density < 16:      # every cell while the field is young
            a.mode, a.n_partial = 0, 0
            if noise is None:
                noise = torch.rand(C, G ** 3, 3, device=dev)
        else:                           # afterwards G^3/4 uniform cells + G^3/4 currently occupied ones per cascade
            n = G ** 3 // 4
            a.mode, a.n_partial = 1, n
            if coords is None:
                coords = torch.randint(0, G, (C, n, 3), device=dev, dtype=torch.int32)
            if occ_rand is None:
                occ_rand = torch.randint(0, 2 ** 31 - 1, (C, n), device=dev, dtype=torch.int32)
            if noise is None:
                noise = torch.rand(C, 2 * n, 3, device=dev)
            coords, occ_rand = require(coords, torch.int32, "coords"), require(occ_rand, torch.int32, "occ_rand")
            a.coords, a.occ_rand = coords.data_ptr(), occ_rand.data_ptr()
        a.noise = require(noise, torch.float32, "noise").data_ptr()
        state = torch.empty(2, dtype=torch.float32, device=dev)
        a.state = state.data_ptr()
        mip_bytes = int(_lib.load().pnr_occupancy_mip_bytes(C, G))
        mip = torch.empty(mip_bytes // 4, dtype=torch.int32, device=dev) if raymarching.USE_MIP and G % 4 == 0 and mip_bytes <= 64 * 1024 and bits.data_ptr() % 8 == 0 else None
        a.mip = mip.data_ptr() if mip is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib = _lib.load()
        if self._fused_sweep_ok():
            enc = self.encoder
            ws = self._occupancy_workspace(self.occupancy_chunk)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            a.embeddings, a.offsets = enc.embeddings.detach().data_ptr(), enc.offsets.data_ptr()
            a.num_levels, a.S, a.base_resolution, a.gridtype = enc.num_levels, float(math.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id
            a.packed_sigma_net = self._sigma_blob().data_ptr()
            _lib.check(lib.pnr_occupancy_update(ctypes.byref(a), stream), "pnr_occupancy_update")
        else:
            ws = self._occupancy_workspace(0)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            _lib.check(lib.pnr_occupancy_begin(ctypes.byref(a), stream), "pnr_occupancy_begin")
            total = int(lib.pnr_occupancy_samples(ctypes.byref(a)))
            step = max(int(S), 1) ** 3
            for first in range(0, total, step):
                count = min(step, total - first)
                pts = torch.empty(count, 4, dtype=torch.float32, device=dev)
                _lib.check(lib.pnr_occupancy_points(ctypes.byref(a), first, count, ctypes.c_void_p(pts.data_ptr()), stream), "pnr_occupancy_points")
                sig = self.density(pts[:, :3].contiguous())["sigma"].reshape(-1).detach().float().contiguous()
                _lib.check(lib.pnr_occupancy_scatter(ctypes.byref(a), ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(sig.data_ptr()), count, stream), "pnr_occupancy_scatter")
            _lib.check(lib.pnr_occupancy_commit(ctypes.byref(a), stream), "pnr_occupancy_commit")
        raymarching.note_bitfield_written(bits, C, G, self.bound, mip)
        self.mean_density = _LazyScalar(lambda state=state: float(state[0]))
        self.iter_density += 1


class _RendererBase(nn.Module):
    def _init_march_state(self, bound, cuda_ray, density_scale, min_near, density_thresh, bg_radius):
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        self.march_mode = "compat"
        self.count_rendered = False
        aabb_train = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb_train)
        self.register_buffer("aabb_infer", aabb_train.clone())
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    def invalidate_fused_caches(self):
        """Forget the blobs the fused kernels derive from the parameters (fused.invalidate_fused_caches).  load_state_dict and
        initialize_palette call it; call it after writing parameters through `.data` (an EMA swap, nerf/utils.py:829-839)."""
        from .fused import invalidate_fused_caches
        invalidate_fused_caches(self)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_fused_caches()
        return out

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ shared inference loop
    def _infer_loop(self, rays_o, rays_d, nears, fars, perturb, dt_gamma, max_steps, shade):
        """while step < max_steps: march -> shade(...) -> compaction (nerf/renderer.py:354-380).
        `shade(st, n_alive, n_step, xyzs, dirs, deltas)` evaluates the field and runs the composites
        (composite_rays last: it is the one that mutates rays_alive / rays_t)."""
        N = rays_o.shape[0]
        st = _MarchState(N, nears, rays_o.device)
        device_mode = self.march_mode == "device"
        if device_mode:
            spare = torch.empty_like(st.rays_alive)
            count = torch.empty(1, dtype=torch.int32, device=rays_o.device)
            host_count = torch.empty(1, dtype=torch.int32).pin_memory()
        n_alive = N
        step = 0
        while step < max_steps:
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, st.rays_alive, st.rays_t, rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128,
                                                        perturb if step == 0 else False, dt_gamma, max_steps)
            st.n_samples += xyzs.shape[0]
            if self.count_rendered:
                st.rendered += (deltas[:, 0] > 0).sum()
            shade(st, n_alive, n_step, xyzs, dirs, deltas)
            if device_mode:
                out, _ = raymarching.compact_alive(st.rays_alive, n_alive, out=spare, count=count)
                spare, st.rays_alive = st.rays_alive, out
                host_count.copy_(count, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                n_alive = int(host_count.item())
            else:
                st.rays_alive = st.rays_alive[st.rays_alive >= 0]
                n_alive = st.rays_alive.shape[0]
            step += n_step
        return st

    # One inference frame of the device-driven loop in three steps (march_mode "native" only; round 6).  A caller with a queue of frames -- a video path, a rank's
    # shard loop -- overlaps its own work for frame i + 1 with frame i's kernels:
    #     h = m.render_prepare(rays(0), ...); m.render_launch(h)
    #     for i in 1..: nxt = m.render_prepare(rays(i), ...); out = m.render_finish(h); m.render_launch(nxt); h = nxt; consume(out)
    # render_prepare allocates the outputs and fills the argument struct (it may run while the previous frame is on the device), render_launch enqueues the frame
    # and returns at once (pnr_*_render_frame_submit), render_finish waits and returns exactly what render() returns.  Parameters must not change in between.
    def render_prepare(self, rays_o, rays_d, **kwargs):
        if not (self.cuda_ray and not self.training and self.march_mode == "native"):
            raise RuntimeError("render_prepare / render_launch / render_finish exist for march_mode = 'native' inference frames")
        with torch.no_grad():
            pending = self.run_cuda(rays_o, rays_d, _phase="prepare", **kwargs)
        pending.redo = lambda: self.render_prepare(rays_o, rays_d, **kwargs)
        return pending

    def render_launch(self, pending):
        """Enqueue a prepared frame; returns the pending frame to finish (a NEW one when the model's blobs were rebuilt since render_prepare -- a frame in
        front of it found its sources rewritten -- and the frame had to be prepared again)."""
        from .fused import StaleFrame
        try:
            pending.fused.frame_launch(pending.tok)
        except StaleFrame:
            pending = pending.redo()
            pending.fused.frame_launch(pending.tok)
        return pending

    def render_finish(self, pending):
        with torch.no_grad():
            return pending.complete(pending.fused.frame_finish(pending.tok))

    def render_wait(self, pending):
        """render_finish in two halves: wait for the frame (True: its outputs are valid and render_result() only builds the result dict -- the caller may
        render_launch its next frame first; False: ask for render_result() before launching anything: the frame is rendered again)."""
        return pending.fused.frame_wait(pending.tok)

    def render_result(self, pending):
        with torch.no_grad():
            return pending.complete(pending.fused.frame_result(pending.tok))

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """nerf/renderer.py:564-603 / palette/renderer.py:554-573 -- never staged when cuda_ray."""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        if not hasattr(self, "run"):
            raise ValueError("Pure pytorch version is not available")   # palette/renderer.py:292-294
        # Uniform-sampling path (BASELINE configs[0]).  The reference's dispatcher crashes here by accident (nerf/renderer.py:591 reads a
        # key run() never returns, :601 passes rays_gt into num_steps); run() itself works and is what this mirrors.  Staged rendering
        # follows nerf/renderer.py:577-589: batches of max_ray_batch rays, B == 1.
        if not staged or self.training:
            return self.run(rays_o, rays_d, **kwargs)
        B, N = rays_o.shape[:2]
        out = {"depth": torch.empty(B, N, device=rays_o.device), "image": torch.empty(B, N, 3, device=rays_o.device),
               "weights_sum": torch.empty(B * N, device=rays_o.device)}
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                r = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                out["depth"][b:b + 1, head:tail] = r["depth"]
                out["image"][b:b + 1, head:tail] = r["image"]
                out["weights_sum"][b * N + head:b * N + tail] = r["weights_sum"]
        return out


class PendingFrame:
    """A native-loop frame between render_prepare and render_finish (fused.py: frame_prepare / frame_launch / frame_finish)."""
    __slots__ = ("fused", "tok", "complete", "redo")

    def __init__(self, fused, tok, complete):
        self.fused, self.tok, self.complete, self.redo = fused, tok, complete, None


def _zero_map(owner, name, shape, like):
    """An all-zero map the reference allocates afresh per frame (nerf/renderer.py:386 rgb_norm, palette/renderer.py:441 clip_feat without a clip head):
    the native frame path hands out one cached tensor per shape instead (a fill launch and its host time per frame otherwise) -- as long as nobody has
    written into it: torch counts in-place writes (`_version`), a touched map is replaced."""
    key = "_zero_" + name
    cached = owner.__dict__.get(key)
    if cached is not None and cached[0].shape == shape and cached[0].device == like.device and cached[0]._version == cached[1]:
        return cached[0]
    z = torch.zeros(*shape, dtype=torch.float32, device=like.device)
    owner.__dict__[key] = (z, z._version)
    return z


class NeRFRenderer(_OccupancyMaintenance, _RendererBase):
    """nerf/renderer.py:61-125"""

    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1, filter_close_point=False):
        super().__init__()
        self.filter_close_point = filter_close_point
        self._init_march_state(bound, cuda_ray, density_scale, min_near, density_thresh, bg_radius)

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        """nerf/renderer.py:127-255 -- the path without an occupancy grid: num_steps samples spread evenly over [near, far] of the box,
        optionally upsample_steps more drawn from the coarse weights (sample_pdf), alpha compositing in torch.  Encoders and near/far are
        the HIP ops; depth is normalised to [0, 1] over [near, far] (unlike run_cuda's)."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        lo, hi = aabb[:3], aabb[3:]

        def points(z):  # [N, T] -> [N, T, 3], clipped to the box
            return torch.min(torch.max(rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1), lo), hi)

        # fused_field: one fused MFMA field launch per query gives sigma and rgb of every point; rgb of the points the reference does not
        # send through its colour head (mask below) is zeroed afterwards, which is what its masked colour query returns for them
        fast = bool(getattr(self, "fused_field", False)) and not self.training and not torch.is_grad_enabled() and not torch.is_autocast_enabled()

        def query(xyzs, T):
            if fast:
                sigma, rgb = self(xyzs.reshape(-1, 3), rays_d.view(-1, 1, 3).expand(N, T, 3).reshape(-1, 3))
                return {"sigma": sigma.view(N, T, 1), "rgb": rgb.view(N, T, 3)}
            return {k: v.view(N, T, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}

        def ray_weights(z, sigma):  # transmittance-weighted opacities; the last interval is one uniform step long
            deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
            trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1), dim=-1)[..., :-1]
            return deltas, alphas * trans

        z_vals = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand(N, num_steps)
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist
        xyzs = points(z_vals)
        fields = query(xyzs, num_steps)
        if upsample_steps > 0:
            with torch.no_grad():
                deltas, weights = ray_weights(z_vals, fields["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(z_mid, weights[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = points(new_z)
            new_fields = query(new_xyzs, upsample_steps)   # only the new points go through the field again
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, 1, order.unsqueeze(-1).expand_as(xyzs))
            for k in fields:
                both = torch.cat([fields[k], new_fields[k]], dim=1)
                fields[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))
        _, weights = ray_weights(z_vals, fields["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: v.reshape(-1, v.shape[-1]) for k, v in fields.items()}
        mask = weights > 1e-4   # the colour head only runs where a sample matters (hard-coded in the reference too)
        if fast:
            rgbs = fields["rgb"] * mask.unsqueeze(-1)
        else:
            rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **flat).view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        depth = torch.sum(weights * ((z_vals - nears) / (fars - nears)).clamp(0, 1), dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        if self.bg_radius > 0:
            bg_color = self.background(raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius), rays_d.reshape(-1, 3))
        elif bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": weights_sum}

    def run_cuda(self, rays_o, rays_d, rays_gt=None, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=1e-4, **kwargs):
        """nerf/renderer.py:258-393"""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        if rays_gt is not None:
            rays_gt = rays_gt.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        aabb = self.aabb_train if self.training else self.aabb_infer
        # the device-driven frame computes near / far inside its own first launch (same arithmetic, one launch and one operator call less per frame)
        native_frame = not self.training and self.march_mode == "native" and rays_o.is_cuda and aabb.is_cuda
        nears, fars = (None, None) if native_frame else raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d)
        elif bg_color is None:
            bg_color = 1
        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            if rays_gt is not None:
                gt_rgbs = torch.zeros_like(xyzs)
                raymarching.spread_ray_to_sample(rays_gt, rays, gt_rgbs)
                rgb_norm = ((gt_rgbs - rgbs) ** 2).sum(-1, keepdim=True).repeat(1, 3)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            if rays_gt is not None:
                _, _, rgb_norm_map = raymarching.composite_rays_train(sigmas, rgb_norm, deltas, rays, T_thresh)
                rgb_norm_map = rgb_norm_map.mean(dim=-1).view(*prefix)
            else:   # the reference composites an all-zero rgb_norm here (nerf/renderer.py:304-309, 327): the map is exactly zero and so is its gradient
                rgb_norm_map = torch.zeros(*prefix, dtype=image.dtype, device=image.device)
            # image and depth (nerf/renderer.py:328-332: background blend, depth normalisation) are formed on first access; train_loss()
            # computes them, the loss and the gradients from results.raw in one launch each way instead
            results = TrainResults(RawTrain(weights_sum, depth, image, None, nears, fars, bg_color, tuple(prefix), 0, 0))
            results["rgb_norm"] = rgb_norm_map
            results["weights_sum"] = weights_sum
            return results
        elif self.march_mode == "native":
            # device-driven loop: same schedule and arithmetic, no per-iteration host sync (pnr_nerf_render_frame)
            if getattr(self, "_fused", None) is None:
                from .fused import NeRFFieldFused
                self._fused = NeRFFieldFused(self)
            if perturb:
                raise RuntimeError("march_mode='native' covers inference without perturbation")
            # under fp16 autocast (the reference's -O mode) the loop looks the hash table up as fp16 with the reference's half interpolation
            # (gridencoder/grid.py:36-39), the field stays on its fp32-accurate matrix path; outputs are fp32 as the reference's are
            was_half = self._fused.table_half
            self._fused.table_half = was_half or torch.is_autocast_enabled()
            frame_args = (rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh)
            frame_kw = dict(bg_color=bg_color, aabb=aabb if native_frame else None, min_near=self.min_near)
            try:
                if kwargs.get("_phase") == "prepare":   # render_prepare(): everything in front of the library call, now; the frame itself in render_launch / render_finish
                    tok = self._fused.frame_prepare(*frame_args, **frame_kw)
                else:
                    ret = self._fused.render_frame(*frame_args, **frame_kw)
            finally:
                self._fused.table_half = was_half

            def complete(ret):
                weights_sum, depth_acc, image_acc, stats = ret
                nears, fars = stats["nears"], stats["fars"]
                if stats["finished"]:   # the frame call applied the epilogue below itself (same fp32 operations, one launch less each)
                    image, depth = image_acc, depth_acc
                else:
                    image = image_acc + (1 - weights_sum).unsqueeze(-1) * bg_color
                    depth = torch.clamp(depth_acc - nears, min=0) / (fars - nears)
                image = image.view(*prefix, 3)
                depth = depth.view(*prefix)
                results["n_samples"] = stats["rows"]
                results["rendered"] = torch.tensor([stats["rendered"]], dtype=torch.int64)   # host tensor: the count came back with the control block
                results["iterations"], results["host_looks"] = stats["iterations"], stats["looks"]
                results["grid_ms"], results["grid_launches"] = stats["grid_ms"], stats["grid_launches"]
                results["depth"] = depth
                results["image"] = image
                results["rgb_norm"] = _zero_map(self, "rgb_norm", tuple(prefix), image)
                results["weights_sum"] = weights_sum
                return results

            if kwargs.get("_phase") == "prepare":
                return PendingFrame(self._fused, tok, complete)
            return complete(ret)
        else:
            def shade(st, n_alive, n_step, xyzs, dirs, deltas):
                sigmas, rgbs = self(xyzs, dirs)
                sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, st.rays_alive, st.rays_t, sigmas, rgbs, deltas, st.weights_sum, st.depth,
                                           st.image, T_thresh)

            st = self._infer_loop(rays_o, rays_d, nears, fars, perturb, dt_gamma, max_steps, shade)
            weights_sum = st.weights_sum
            image = st.image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(st.depth - nears, min=0) / (fars - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            rgb_norm_map = torch.zeros_like(image[..., 0])
            results["n_samples"] = st.n_samples
            results["rendered"] = st.rendered
        results["depth"] = depth
        results["image"] = image
        results["rgb_norm"] = rgb_norm_map
        results["weights_sum"] = weights_sum
        return results


class RegionEdit(nn.Module):
    """Regional appearance editing controller (palette/renderer.py:84-147): hue shift and
    saturation/value scaling per palette basis in HSV space, blended by a spatial / semantic
    Gaussian window."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.mean_xyz = None
        self.mean_clip = None
        self.std_xyz = 1
        self.std_clip = 1
        self.weight_mode = False
        self.delta_hsv = torch.zeros(self.opt.num_basis, 3)
        self.delta_hsv[..., 1:3] = 1

    def update_cent(self, mean_xyz=None, mean_clip=None):
        self.mean_xyz = None if mean_xyz is None else mean_xyz[None, ...]
        self.mean_clip = None if mean_clip is None else mean_clip[None, ...]

    def update_std(self, std_xyz=None, std_clip=None):
        if std_xyz is not None:
            self.std_xyz = std_xyz
        if std_clip is not None:
            self.std_clip = std_clip

    def update_delta_hsv(self, rgb_orig, rgb_new):
        if rgb_orig.device != self.delta_hsv.device:
            self.delta_hsv = self.delta_hsv.type_as(rgb_orig)
        nb = self.opt.num_basis
        hsv_all = rgb_to_hsv(torch.cat([rgb_orig, rgb_new], dim=0))
        hsv_orig, hsv_new = hsv_all[:nb], hsv_all[nb:]
        self.delta_hsv[:, 0] = torch.fmod((hsv_new[:, 0] - hsv_orig[:, 0] + 360), 360)
        self.delta_hsv[:, 1] = (hsv_new[:, 1] / hsv_orig[:, 1] + 1e-9)
        self.delta_hsv[:, 2] = (hsv_new[:, 2] / hsv_orig[:, 2] + 1e-9)

    def forward(self, rgbs, xyz=None, clip_feat=None):
        hsv = rgb_to_hsv(rgbs)
        if rgbs.device != self.delta_hsv.device:
            self.delta_hsv = self.delta_hsv.type_as(rgbs)
        weight = torch.ones_like(rgbs[..., 0:1, 0])
        if xyz is not None and self.mean_xyz is not None:
            weight *= torch.exp(-((xyz - self.mean_xyz) ** 2.).sum(dim=-1, keepdim=True) / self.std_xyz)
        if clip_feat is not None and self.mean_clip is not None:
            weight *= torch.exp(-((clip_feat - self.mean_clip) ** 2.).sum(dim=-1, keepdim=True) / self.std_clip)
        hsv_new = hsv.clone()
        hsv_new[..., 0] = torch.fmod((hsv[..., 0] + self.delta_hsv[..., 0] + 360), 360)
        hsv_new[..., 1] = torch.clip((hsv[..., 1] * self.delta_hsv[..., 1]), 0)
        hsv_new[..., 2] = torch.clip((hsv[..., 2] * self.delta_hsv[..., 2]), 0)
        rgb_new = hsv_to_rgb(hsv_new)
        if self.weight_mode:
            return weight[..., None].repeat(1, self.opt.num_basis, 3)
        return torch.lerp(rgbs, rgb_new, weight[..., None])


class Stylizer(nn.Module):
    """User-guided photorealistic style transfer head (palette/renderer.py:150-183): per-basis intensity shift dI, palette shift dP and a
    3x3 transform of the offsets per basis (kept near a rotation by ARAP_loss), optimised by the GUI and then used by run_cuda in place of
    the plain colour-basis composite.  Same parameter names and shapes as the reference's."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nb = opt.num_basis
        self.dI = nn.Parameter(torch.zeros(nb, dtype=torch.float32))
        self.dP = nn.Parameter(torch.zeros(1, nb, 3, dtype=torch.float32))
        self.ddelta = nn.Parameter(torch.eye(3, dtype=torch.float32)[None].repeat(nb, 1, 1))

    def ARAP_loss(self):
        eye = torch.eye(3, dtype=torch.float32, device=self.ddelta.device)[None]
        return ((torch.bmm(self.ddelta, self.ddelta.transpose(1, 2)) - eye) ** 2).sum()

    def forward(self, radiance, omega, palette, offsets, view_dep=None):
        nb = self.opt.num_basis
        lead = offsets.shape[:-2]
        intensity = (F.softplus(radiance.reshape(-1, 1, 1)).repeat(1, nb, 1) + self.dI[None, :, None]).clamp(0)
        colour = palette.reshape(-1, nb, 3) + self.dP + torch.einsum("npi,pij->npj", offsets.reshape(-1, nb, 3), self.ddelta)
        rgbs = (omega.reshape(-1, nb, 1) * (intensity * colour).clamp(0, 1)).sum(dim=-2)
        if view_dep is not None:
            rgbs = rgbs + view_dep.detach()
        return rgbs.reshape(*lead, 3)


class PaletteRenderer(_RendererBase):
    """palette/renderer.py:186-245"""

    def __init__(self, opt, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.opt = opt
        self.num_basis = opt.num_basis
        self.freeze_basis_color = opt.use_initialization_from_rgbxy
        self.require_smooth_loss = False
        self.color_weight = 0
        self.edit = None
        self.stylizer = None
        self.view_dep_weight = 1
        self.offsets_weight = 1
        self._init_march_state(bound, cuda_ray, density_scale, min_near, density_thresh, bg_radius)
        if opt.test or not opt.use_initialization_from_rgbxy:
            self.basis_color = nn.Parameter(torch.zeros([self.num_basis, 3]) + 0.5, requires_grad=True)
        else:
            self.basis_color = None

    def initialize_palette(self, color_list=None, hist_weights=None):
        """palette/renderer.py:247-268; with --color_space linear the extracted (sRGB) palette is moved to linear colour first (:256-259,
        srgb_to_linear of nerf/utils.py:39-40)."""
        if color_list is None:
            if self.basis_color is None:
                self.basis_color = nn.Parameter(torch.zeros([self.num_basis, 3]) + 0.5, requires_grad=True)
        else:
            dev = self.aabb_train.device
            colors = torch.as_tensor(color_list, dtype=torch.float32, device=dev).reshape(self.num_basis, 3).clone()
            space = getattr(self.opt, "color_space", "srgb")
            if space == "linear":
                colors = torch.where(colors < 0.04045, colors / 12.92, ((colors + 0.055) / 1.055) ** 2.4)
            self.basis_color = nn.Parameter(colors, requires_grad=True)
        self.invalidate_fused_caches()
        self.basis_color_origin = nn.Parameter(self.basis_color.data.clone(), requires_grad=False)
        if hist_weights is not None:
            hw = torch.as_tensor(hist_weights).float().permute(3, 0, 1, 2).unsqueeze(0)
            self.hist_weights = nn.Parameter(hw, requires_grad=False)

    def forward(self, x, d):
        raise NotImplementedError()

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4,
                 gui_mode=False, **kwargs):
        """palette/renderer.py:296-552"""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        nb, clip_dim = self.num_basis, self.opt.clip_dim
        aabb = self.aabb_train if self.training else self.aabb_infer
        # under fp16 autocast only the native loop takes the fused path (fp16 tables, fp32-accurate field); it has no clip-head variant there
        autocast_ok = not torch.is_autocast_enabled() or (self.march_mode == "native" and not perturb and not self.opt.pred_clip)
        # RegionEdit and the Stylizer run inside the fused field kernel's epilogue (pnr_palette_edit): editing costs no extra launch
        use_fused = not self.training and bool(getattr(self, "fused_field", False)) and autocast_ok
        native = use_fused and self.march_mode == "native" and not perturb
        # the device-driven frame computes near / far inside its own first launch (same arithmetic, one launch and one operator call less per frame)
        native_near_far = native and rays_o.is_cuda and aabb.is_cuda
        nears, fars = (None, None) if native_near_far else raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d)
        elif bg_color is None:
            bg_color = 1
        results = {}

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
            M = xyzs.shape[0]
            # geometry is frozen here (sigma detached below, geo_feat inside the network): encoder + sigma_net as the fused density kernel
            sigmas, clip_feat, omega, offsets_radiance, view_dep, diffuse = self(xyzs, dirs, frozen_density=bool(getattr(self, "fused_train_density", True)))
            offsets, radiance = offsets_radiance[..., :-1], offsets_radiance[..., -1:]
            sigmas = (self.density_scale * sigmas).detach()  # palette/renderer.py:333-334
            fused_shade = bool(getattr(self, "fused_train_shade", True)) and xyzs.is_cuda and nb <= 16
            radiance = radiance.reshape(M, 1, 1)
            offsets = offsets.reshape(M, nb, 3)
            omega = omega.reshape(M, nb, 1)
            view_dep = view_dep.reshape(M, 3)
            diffuse = diffuse.reshape(M, 3)
            clip_feat = clip_feat.reshape(M, clip_dim)
            basis_color = self.basis_color[None, :, :].clamp(0, 1)
            if self.freeze_basis_color:
                basis_color = basis_color.detach()
            if not fused_shade:
                final_color = F.softplus(radiance) * (basis_color + offsets)
                basis_rgb = omega * final_color
                rgbs = basis_rgb.sum(dim=-2) + view_dep.detach()
                direct_rgb = diffuse + view_dep
                omega_sparsity = omega[..., 0].sum(dim=-1, keepdim=True) / ((omega[..., 0] ** 2).sum(dim=-1, keepdim=True) + 1e-6) - 1
                offsets_norm = (offsets ** 2).sum(dim=-1).sum(dim=-1, keepdim=True)
                view_dep_norm = (view_dep ** 2).sum(dim=-1, keepdim=True)
            if self.require_smooth_loss:
                xyzs_diff = (xyzs + torch.rand_like(xyzs) * self.bound * 0.03).clamp(-self.bound, self.bound)
                _, clip_feat_diff, omega_diff, _, _, diffuse_diff = self(xyzs_diff, dirs, frozen_density=bool(getattr(self, "fused_train_density", True)))
                omega_diff = omega_diff.reshape(M, nb, 1)
                diffuse_diff = diffuse_diff.reshape(M, 3)
                xyzs_weight = (xyzs - xyzs_diff).norm(dim=-1, keepdim=True) ** 2 / self.bound ** 2 / self.opt.smooth_sigma_xyz
                rgb_weight = (diffuse - diffuse_diff).norm(dim=-1, keepdim=True) ** 2 / self.opt.smooth_sigma_color
                if self.opt.pred_clip and self.opt.smooth_sigma_clip > 0:
                    clip_weight = (clip_feat - clip_feat_diff).norm(dim=-1, keepdim=True) / self.opt.smooth_sigma_clip
                else:
                    clip_weight = 0
                smooth_weight = torch.exp(-xyzs_weight - rgb_weight - clip_weight).detach()
                smooth_norm = ((omega_diff - omega)[..., 0] ** 2).sum(dim=-1, keepdim=True) * smooth_weight
                if self.opt.pred_clip:
                    smooth_norm += ((clip_feat_diff - clip_feat) ** 2).sum(dim=-1, keepdim=True) * smooth_weight
            else:
                smooth_norm = None

            # 13 + clip_dim + nb channels in ONE flex composite (palette/renderer.py:384-386)
            if fused_shade:  # the whole colour-basis composite + the 33-column row in one HIP launch each way (pnr_palette_train_shade_*)
                # without a clip head the row keeps its clip_dim columns, zero, as the reference's torch.zeros clip_feat does
                rgbs, all_buffer = palette_train_shade(omega.reshape(M, nb), offsets_radiance.reshape(M, 3 * nb + 1), view_dep, diffuse,
                                                       clip_feat if self.opt.pred_clip else None, smooth_norm,
                                                       self.basis_color.detach() if self.freeze_basis_color else self.basis_color, clip_dim)
            else:
                if smooth_norm is None:
                    smooth_norm = torch.zeros_like(omega_sparsity)
                all_buffer = torch.cat([omega_sparsity, view_dep_norm, offsets_norm, smooth_norm, view_dep, direct_rgb, diffuse, clip_feat,
                                        omega[..., 0]], dim=-1)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            all_map = raymarching.composite_rays_flex_train(sigmas, all_buffer, deltas, rays, T_thresh)
            # image, depth and direct_rgb (palette/renderer.py:387-391,399: background blend, depth normalisation) are formed on first access;
            # train_loss() computes them, the loss and every gradient from results.raw in one launch each way instead
            results = TrainResults(RawTrain(weights_sum, depth, image, all_map, nears, fars, bg_color, tuple(prefix), nb, clip_dim))
            results["weights_sum"] = weights_sum
            results["omega_sparsity"] = all_map[..., 0:1].view(*prefix)
            results["view_dep_norm"] = all_map[..., 1:2].view(*prefix)
            results["offsets_norm"] = all_map[..., 2:3].view(*prefix)
            results["smooth_norm"] = all_map[..., 3:4].view(*prefix)
            results["view_dep_rgb"] = all_map[..., 4:7].view(*prefix, 3)
            results["diffuse_rgb"] = all_map[..., 10:13].view(*prefix, 3)
            results["clip_feat"] = all_map[..., 13:13 + clip_dim].view(*prefix, clip_dim)
            results["basis_acc"] = all_map[..., 13 + clip_dim:13 + clip_dim + nb].view(*prefix, nb)
            return results

        f32 = dict(dtype=torch.float32, device=device)
        if use_fused and self.stylizer is not None and not gui_mode:
            raise RuntimeError("the Stylizer renders in gui_mode only (palette/renderer.py:481-488 defines no basis maps for it)")
        if use_fused:
            if getattr(self, "_fused", None) is None:
                from .fused import PaletteFieldFused
                self._fused = PaletteFieldFused(self)
            if not native:   # the device-driven loop brings its own (uninitialised, fully written) aux map
                aux_map = torch.zeros(N, self._fused.aux_channels, **f32)
        if not use_fused:    # the reference's six maps (palette/renderer.py:436-441); the fused paths composite one packed aux row instead
            view_dep_rgb_map = torch.zeros(N, 3, **f32)
            direct_rgb_map = torch.zeros(N, 3, **f32)
            basis_rgb_map = torch.zeros(N, 3 * nb, **f32)
            unscaled_basis_rgb_map = torch.zeros(N, 3 * nb, **f32)
            basis_acc_map = torch.zeros(N, nb, **f32)
        if not use_fused or self._fused.clip_dim != clip_dim:
            clip_feat_map = _zero_map(self, "clip_feat", (N, clip_dim), rays_o) if native else torch.zeros(N, clip_dim, **f32)

        def shade_fused(st, n_alive, n_step, xyzs, dirs, deltas):
            # one fused field launch + ONE flex composite over the packed aux row instead of ~40 launches and 6 flex composites
            sigmas, rgbs, aux = self._fused(xyzs, dirs, deltas)
            raymarching.composite_rays_flex(n_alive, n_step, self._fused.aux_channels, st.rays_alive, st.rays_t, sigmas, aux, deltas, st.weights_sum,
                                            aux_map, T_thresh)
            raymarching.composite_rays(n_alive, n_step, st.rays_alive, st.rays_t, sigmas, rgbs, deltas, st.weights_sum, st.depth, st.image, T_thresh)

        def shade(st, n_alive, n_step, xyzs, dirs, deltas):
            M = xyzs.shape[0]
            sigmas, clip_feat, omega, offsets_radiance, view_dep, diffuse = self(xyzs, dirs)
            offsets, radiance = offsets_radiance[..., :-1], offsets_radiance[..., -1:]
            radiance = radiance.reshape(M, 1, 1)
            offsets = offsets.reshape(M, nb, 3)
            omega = omega.reshape(M, nb, 1)
            view_dep = view_dep.reshape(M, 3)
            diffuse = diffuse.reshape(M, 3)
            clip_feat = clip_feat.reshape(M, clip_dim)
            basis_color = self.basis_color[None, :, :].clamp(0, 1)
            basis_rgb = unscaled_basis_rgb = None
            if self.stylizer is not None:
                rgbs = self.stylizer(radiance, omega, basis_color, offsets, view_dep)
            else:
                # the palette colour-basis composite (palette/renderer.py:482-494)
                final_color = F.softplus(radiance) * (basis_color + self.offsets_weight * offsets)
                unscaled_basis_rgb = basis_color + offsets
                if self.edit is not None:
                    final_color = self.edit(final_color, xyzs, clip_feat)
                basis_rgb = omega * final_color
                rgbs = basis_rgb.sum(dim=-2) + self.view_dep_weight * view_dep
            sigmas = self.density_scale * sigmas
            a = (n_alive, n_step)
            if not gui_mode:
                direct_rgb = diffuse + view_dep
                fl = raymarching.composite_rays_flex
                fl(*a, 3, st.rays_alive, st.rays_t, sigmas, direct_rgb, deltas, st.weights_sum, direct_rgb_map, T_thresh)
                fl(*a, 3, st.rays_alive, st.rays_t, sigmas, view_dep, deltas, st.weights_sum, view_dep_rgb_map, T_thresh)
                fl(*a, nb, st.rays_alive, st.rays_t, sigmas, omega, deltas, st.weights_sum, basis_acc_map, T_thresh)
                fl(*a, nb * 3, st.rays_alive, st.rays_t, sigmas, basis_rgb.reshape(M, nb * 3), deltas, st.weights_sum, basis_rgb_map, T_thresh)
                fl(*a, nb * 3, st.rays_alive, st.rays_t, sigmas, unscaled_basis_rgb.reshape(M, nb * 3), deltas, st.weights_sum,
                   unscaled_basis_rgb_map, T_thresh)
            raymarching.composite_rays_flex(*a, clip_dim, st.rays_alive, st.rays_t, sigmas, clip_feat, deltas, st.weights_sum, clip_feat_map,
                                            T_thresh)
            # must come last: the only composite that mutates rays_alive / rays_t / weights_sum (palette/renderer.py:517-519)
            raymarching.composite_rays(*a, st.rays_alive, st.rays_t, sigmas, rgbs, deltas, st.weights_sum, st.depth, st.image, T_thresh)

        def tail(st, aux_map, finished, stats, nears, fars):
            nonlocal clip_feat_map, direct_rgb_map, view_dep_rgb_map, basis_acc_map, basis_rgb_map, unscaled_basis_rgb_map
            if use_fused:  # unpack the composited aux row into the reference's maps
                direct_rgb_map, view_dep_rgb_map = aux_map[:, 0:3], aux_map[:, 3:6]
                basis_acc_map = aux_map[:, 6:6 + nb]
                basis_rgb_map = aux_map[:, 6 + nb:6 + 4 * nb]
                unscaled_basis_rgb_map = aux_map[:, 6 + 4 * nb:6 + 7 * nb]
                if self._fused.clip_dim == clip_dim:
                    clip_feat_map = aux_map[:, 6 + 7 * nb:6 + 7 * nb + clip_dim]
            weights_sum = st.weights_sum
            if finished:
                image, depth, depth_origin = st.image, st.depth, stats["depth_raw"]
            else:
                image = st.image + (1 - weights_sum).unsqueeze(-1) * bg_color
                depth_origin = st.depth.clone()
                depth = torch.clamp(st.depth - nears, min=0) / (fars - nears)
            results["depth"] = depth.view(*prefix)
            results["depth_origin"] = depth_origin.view(*prefix)
            results["image"] = image.view(*prefix, 3)
            results["weights_sum"] = weights_sum
            results["clip_feat"] = clip_feat_map.reshape(*prefix, clip_dim)
            results["n_samples"] = st.n_samples
            results["rendered"] = st.rendered
            if not gui_mode:
                results["direct_rgb"] = (direct_rgb_map if finished else direct_rgb_map + (1 - weights_sum).unsqueeze(-1) * bg_color).reshape(*prefix, 3)
                results["view_dep_rgb"] = view_dep_rgb_map.reshape(*prefix, 3)
                results["basis_rgb"] = basis_rgb_map.reshape(*prefix, nb * 3)
                results["unscaled_basis_rgb"] = unscaled_basis_rgb_map.reshape(*prefix, nb * 3)
                results["basis_acc"] = basis_acc_map.reshape(*prefix, nb)
            return results

        if native:  # device-driven loop: same schedule and arithmetic, no per-iteration host sync (pnr_palette_render_frame)
            was_half = self._fused.table_half
            self._fused.table_half = was_half or torch.is_autocast_enabled()     # -O mode: fp16 tables with the reference's half interpolation
            frame_args = (rays_o, rays_d, nears, fars, dt_gamma, max_steps, T_thresh)
            frame_kw = dict(bg_color=bg_color, aabb=aabb if native_near_far else None, min_near=self.min_near)
            try:
                if kwargs.get("_phase") == "prepare":   # render_prepare(): the frame itself goes out in render_launch / render_finish
                    tok = self._fused.frame_prepare(*frame_args, **frame_kw)
                else:
                    ret = self._fused.render_frame(*frame_args, **frame_kw)
            finally:
                self._fused.table_half = was_half

            def complete(ret):
                ws_n, depth_n, image_n, aux_n, stats = ret
                st = _MarchState.__new__(_MarchState)
                st.weights_sum, st.depth, st.image, st.n_samples = ws_n, depth_n, image_n, stats["rows"]
                st.rendered = torch.tensor([stats["rendered"]], dtype=torch.int64)   # host tensor: the count came back with the control block
                results["iterations"], results["grid_ms"], results["grid_launches"] = stats["iterations"], stats["grid_ms"], stats["grid_launches"]
                results["host_looks"] = stats["looks"]
                # finished: the frame call's last launch applied the epilogue itself (same fp32 operations; eleven launches less)
                return tail(st, aux_n, stats["finished"], stats, stats["nears"], stats["fars"])

            if kwargs.get("_phase") == "prepare":
                return PendingFrame(self._fused, tok, lambda ret: complete(ret))
            return complete(ret)
        st = self._infer_loop(rays_o, rays_d, nears, fars, perturb, dt_gamma, max_steps, shade_fused if use_fused else shade)
        return tail(st, aux_map if use_fused else None, False, None, nears, fars)
