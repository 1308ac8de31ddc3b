"""Several frames of a camera path in flight at once (video rendering: the reference's test loop renders its 120 poses one after
another, nerf/utils.py:716-740, palette/utils.py:993-1078; the frames are independent).

A frame of the device-driven loop is a chain of ~100 dependent launches whose march launches end in a long, thin tail (a few waves
with the longest rays) and whose host side has a gap between two frames.  A second frame running on its own stream fills both:
measured on MI355X, 800x800 NeRF 4.13 -> 3.09 ms per frame with two frames in flight, 2.88 with three.  The
price is latency: each frame takes longer from submission to completion.

`pnr_*_render_frame` keeps its per-frame state (pinned control block, iteration prediction) per host thread and device, and a
fused-field object owns one workspace: every frame in flight therefore needs its own host thread, its own fused-field object and
its own stream.  `clone_for_concurrent_frames` makes such a handle on the SAME weights; `FramesInFlight` runs the threads.
"""
import copy
import threading

import torch

_FUSED_SETTINGS = ("precision", "table_half", "interleave_tables", "ray_order", "time_grid_kernel")


def clone_for_concurrent_frames(model):
    """A second handle on `model`'s weights for another host thread: parameters, buffers and sub-modules are shared (nothing is
    copied), the fused-field object -- workspace, packed-weight blob, interleaved tables -- is its own.  Plain attributes (edit,
    stylizer, offsets_weight, density_scale, ...) are snapshots of the moment: FramesInFlight.render re-syncs them from the model
    before every run (sync_twin), and the model keeps a weak list of its twins so that invalidate_fused_caches reaches their blobs."""
    import weakref
    fused = getattr(model, "_fused", None)
    if fused is None:
        raise RuntimeError("clone_for_concurrent_frames: the model has no fused field (fused_field = True and one frame rendered, or _fused set)")
    twin = copy.copy(model)                 # shallow: the same Parameter / buffer tensors
    twin.__dict__.pop("_fused_twins", None)
    twin._fused = type(fused)(twin)
    for name in _FUSED_SETTINGS:
        if hasattr(fused, name):
            setattr(twin._fused, name, getattr(fused, name))
    model.__dict__.setdefault("_fused_twins", []).append(weakref.ref(twin))
    return twin


def sync_twin(model, twin):
    """Bring a twin's plain attributes up to date with the model's (everything but its own fused-field object and its private caches):
    `model.edit = RegionEdit(...)`, a changed offsets_weight or density_scale after the clone would otherwise be shadowed by the snapshot."""
    keep = {k: twin.__dict__[k] for k in ("_fused", "_density_fused", "_occ_ws", "_occ_blob") if k in twin.__dict__}
    for k, v in model.__dict__.items():
        if k not in ("_fused", "_density_fused", "_occ_ws", "_occ_blob", "_fused_twins"):
            twin.__dict__[k] = v
    for k in list(twin.__dict__):
        if k not in model.__dict__ and k not in keep:
            del twin.__dict__[k]
    twin.__dict__.update(keep)
    src = getattr(model, "_fused", None)
    if src is not None:
        for name in _FUSED_SETTINGS:
            if hasattr(src, name):
                setattr(twin._fused, name, getattr(src, name))


class _Worker(threading.Thread):
    """One render thread of a FramesInFlight pool, alive until close(): the frame calls keep a pinned control block, their timing events and
    the previous frame's iteration count per HOST THREAD (csrc/frame.hip), so a thread that survives from one render() to the next starts its
    frames with a warm prediction and allocates nothing (round 2 made new threads per call: a hipHostMalloc and cold predictions every time)."""

    def __init__(self, name):
        super().__init__(name=name, daemon=True)
        import queue
        self.jobs = queue.SimpleQueue()
        self.done = queue.SimpleQueue()
        self.start()

    def run(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            try:
                job()
                self.done.put(None)
            except BaseException as e:   # noqa: BLE001 -- handed to the caller
                self.done.put(e)


class FramesInFlight:
    """`n` frames in flight: frame i of a sequence goes to handle i % n, each handle renders on its own host thread and stream.  The
    threads live as long as the object (close() or garbage collection ends them)."""

    def __init__(self, model, n, device=None, shared_stream=False):
        """shared_stream: every handle enqueues into ONE stream -- the frames run back to back, their kernels never overlap (each launch has the
        chip to itself, as with one frame at a time), but the host's gap between two frames (read-back, Python, the first launches of the next
        frame: ~0.15 ms of 3.8 for the 800x800 NeRF frame) is gone: while one thread waits for its frame's control block, the other has already
        queued the next frame behind it.  Default: a stream per handle (kernels of different frames overlap: more throughput, shared launches)."""
        if n < 1:
            raise ValueError("FramesInFlight: n >= 1")
        self.device = device if device is not None else next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("FramesInFlight needs the HIP path (a CUDA/HIP device); there is no CPU fallback")
        self.models = [model] + [clone_for_concurrent_frames(model) for _ in range(n - 1)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(1 if shared_stream else n)] * (n if shared_stream else 1)
        self.workers = [_Worker(f"pnr-frame-{k}") for k in range(n)]

    def close(self):
        for w in self.workers:
            if w.is_alive():
                w.jobs.put(None)
        for w in self.workers:
            w.join(timeout=5.0)
        self.workers = []

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass

    def render(self, rays_of, n_frames, consume=None, before=None, abort=None, **kwargs):
        """Render frames 0 .. n_frames-1; `rays_of(i)` -> (rays_o, rays_d) resident on the device; `consume(i, results)` is called on
        the rendering thread (inside its stream context) as soon as frame i is complete (default: keep the results); `before(i, model)`
        right in front of the frame.  Returns the list of results (or of consume's return values) in frame order.  Exceptions of a
        worker are re-raised here; `abort(error)` is called first, on the failing thread -- pass dist.OrderedGather.abort when consume()
        takes turns through one, or the other threads wait for a frame that will never be submitted."""
        if not self.workers:
            raise RuntimeError("FramesInFlight: closed")
        out = [None] * n_frames
        for twin in self.models[1:]:
            sync_twin(self.models[0], twin)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))   # the inputs may have been produced on the caller's stream

        def work(k):
            try:
                with torch.cuda.stream(self.streams[k]), torch.no_grad():
                    self.streams[k].wait_event(ready)
                    for i in range(k, n_frames, len(self.models)):
                        ro, rd = rays_of(i)
                        if before is not None:
                            before(i, self.models[k])
                        r = self.models[k].render(ro, rd, **kwargs)
                        out[i] = consume(i, r) if consume is not None else r
                    self.streams[k].synchronize()
            except BaseException as e:   # noqa: BLE001
                if abort is not None:
                    abort(e)
                raise

        for k, w in enumerate(self.workers):
            w.jobs.put(lambda k=k: work(k))
        errors = [e for e in (w.done.get() for w in self.workers) if e is not None]
        if errors:
            raise errors[0]
        return out


def render_queue(model, rays_of, n_frames, consume=None, **kwargs):
    """Frames 0 .. n_frames-1 ONE AT A TIME on the device, the host's work for frame i + 1 done while frame i runs (round 6: model.render_prepare /
    render_launch / render_finish over pnr_*_render_frame_submit / _finish).  One host thread, one stream, one workspace; a frame's kernels never share
    the chip with another frame's (FramesInFlight overlaps frames; this only removes the host's turnaround between them -- what a rank rendering its
    shard of every frame of a video needs: 0.15 of a 2.1 ms shard frame).  `rays_of(i)` -> (rays_o, rays_d) on the device; `consume(i, results)` is called
    when frame i is complete, while frame i + 1 is already running.  Returns the list of results (or of consume's return values)."""
    out = [None] * n_frames
    if n_frames == 0:
        return out
    with torch.no_grad():
        ro, rd = rays_of(0)
        cur = model.render_launch(model.render_prepare(ro, rd, **kwargs))
        for i in range(n_frames):
            nxt = None
            if i + 1 < n_frames:
                ro, rd = rays_of(i + 1)
                nxt = model.render_prepare(ro, rd, **kwargs)     # outputs, argument struct, source checksums: under frame i's kernels
            done = cur
            if model.render_wait(done):
                if nxt is not None:
                    cur = model.render_launch(nxt)               # frame i + 1 is on the device before frame i's result dict is even built
                r = model.render_result(done)
            else:                                                # (sources rewritten behind torch's counters, an fp16 overflow: the frame is rendered again first)
                r = model.render_result(done)
                if nxt is not None:
                    cur = model.render_launch(nxt)
            out[i] = consume(i, r) if consume is not None else r
    return out


def render_path(model, poses, intrinsics, H, W, frames_in_flight=2, linear_to_srgb=False, to_host=True, **render_kwargs):
    """The inner loop of the reference's `Trainer.test` over a camera path (nerf/utils.py:704-731, palette/utils.py:993-1044) without the file
    writer: for every pose  get_rays (:133-147) -> model.render -> [linear_to_srgb for linear-colour scenes] -> `(pred * 255).astype(np.uint8)`
    for colour and depth.  Rays are generated on the device (pnr_get_rays), the bytes are formed on the device (pnr_image_to_uint8: a frame
    crosses PCIe as 4 bytes per pixel), and `frames_in_flight` poses are rendered concurrently.
    poses: [n, 4, 4] camera-to-world (host or device).  Returns (rgb uint8 [n, H, W, 3], depth uint8 [n, H, W]): numpy arrays, or device
    tensors with to_host=False."""
    from . import rays as prays
    poses = torch.as_tensor(poses, dtype=torch.float32)
    n = poses.shape[0]
    device = next(model.parameters()).device
    poses = poses.to(device)
    fif = FramesInFlight(model, max(1, min(int(frames_in_flight), n)), device)

    def rays_of(i):
        ro, rd = prays.rays_from_indices(poses[i:i + 1], intrinsics, H, W, None)
        return ro, rd

    def consume(i, r):
        rgb = prays.image_to_uint8(r["image"].reshape(H, W, 3), linear_to_srgb)
        dep = prays.image_to_uint8(r["depth"].reshape(H, W), False)
        return (rgb.cpu(), dep.cpu()) if to_host else (rgb, dep)

    frames = fif.render(rays_of, n, consume=consume, **render_kwargs)
    if to_host:
        import numpy as np
        return np.stack([f[0].numpy() for f in frames]), np.stack([f[1].numpy() for f in frames])
    return torch.stack([f[0] for f in frames]), torch.stack([f[1] for f in frames])
