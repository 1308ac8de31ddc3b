"""Ray sharding of one frame across the GPUs of a node and the per-frame all-gather of the per-ray
outputs.  One process per GPU, torch.distributed ("nccl" == RCCL over xGMI
on ROCm; "gloo" on CPU for the tests).

Partitioning: the frame is cut into TILE x TILE pixel tiles assigned round-robin to ranks (ray cost
varies by >100x between empty-space and object rays; contiguous stripes would leave most ranks
idle).  Every rank renders its rays with the unmodified single-GPU loop -- there is no collective
on the data path of the march -- then ONE all_gather_into_tensor per frame moves the packed
[n_max, K] fp32 buffer (K = 5 for rgb+depth+alpha).  Payload at 800x800, K=5: 1.6 MB per rank;
latency-bound, not bandwidth-bound, on 7 x 153 GB/s xGMI links.  The tile map is static, so the
un-tiling is a precomputed gather index: no masks, no host synchronisation per frame.
"""
import torch
import torch.distributed as dist

TILE = 32


def tile_assignment(H, W, world_size, tile=TILE):
    """int64 [H*W] -> owning rank of every pixel (row-major), tiles dealt round-robin."""
    ty = torch.arange(H) // tile
    tx = torch.arange(W) // tile
    n_tx = (W + tile - 1) // tile
    tile_id = ty[:, None] * n_tx + tx[None, :]
    return (tile_id % world_size).reshape(-1)


def shard_indices(H, W, rank, world_size, tile=TILE):
    """Pixel indices (ascending) rendered by `rank`, and the padded per-rank length."""
    owner = tile_assignment(H, W, world_size, tile)
    counts = torch.bincount(owner, minlength=world_size)
    idx = torch.nonzero(owner == rank, as_tuple=False).reshape(-1)
    return idx, int(counts.max())


class FrameGatherer:
    """Static plan of one frame's all-gather: send/receive buffers and the pixel -> gathered-row index."""

    def __init__(self, H, W, K, device, group=None, tile=TILE, slots=2):
        self.H, self.W, self.K, self.group = H, W, K, group
        self.slots = int(slots)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        owner = tile_assignment(H, W, self.world, tile)
        counts = torch.bincount(owner, minlength=self.world)
        self.n_max = int(counts.max())
        # position of pixel p inside its owner's (ascending) index list
        local_pos = torch.empty(H * W, dtype=torch.int64)
        for r in range(self.world):
            sel = torch.nonzero(owner == r, as_tuple=False).reshape(-1)
            local_pos[sel] = torch.arange(sel.numel())
            if r == self.rank:
                self.idx = sel
        self.gather_index = (owner * self.n_max + local_pos).to(device)
        # `slots` buffer pairs (two by default): the all-gather of frame k may still be in flight while frame k+1 is packed (start / finish
        # below); with F frames in flight (pipeline.FramesInFlight + OrderedGather) up to F + 1 gathers are outstanding
        self.send = [torch.zeros(self.n_max, K, dtype=torch.float32, device=device) for _ in range(self.slots)]
        self.recv = [torch.empty(self.world * self.n_max, K, dtype=torch.float32, device=device) for _ in range(self.slots)]
        self._turn = 0
        # Re-use of a buffer pair is safe by construction, whoever calls from whichever thread and stream: a slot is `busy` from start() to
        # finish() (start() refuses a busy slot; OrderedGather.submit waits for it), and the packing of the next frame into it is ordered
        # behind the last reader of its receive rows (an event recorded after finish()'s index_select) and behind the previous collective.
        self._busy = [False] * self.slots
        self._read = [None] * self.slots
        self._work = [None] * self.slots

    def next_slot_free(self):
        return not self._busy[self._turn]

    def start(self, parts):
        """Pack this rank's rows and launch the all-gather WITHOUT making the compute stream wait for it: the collective runs on the
        communicator's stream while the next frame is rendered.  Returns a handle for finish().  At most `slots` gathers outstanding:
        starting one more before the oldest has been finish()ed is an error (its rows would be overwritten unread)."""
        n = self.idx.numel()
        slot = self._turn
        if self._busy[slot]:
            raise RuntimeError(f"FrameGatherer.start: all {self.slots} buffer pairs hold a gather that has not been finish()ed")
        self._turn = (self._turn + 1) % self.slots
        self._busy[slot] = True
        if self._work[slot] is not None:
            self._work[slot].wait()          # (already complete: its finish() ran) orders this stream behind the collective that read send[slot]
        if self._read[slot] is not None:
            torch.cuda.current_stream(self.send[slot].device).wait_event(self._read[slot])   # the index_select that read recv[slot], maybe on another stream
        torch.cat(parts, dim=1, out=self.send[slot][:n])
        work = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)
        self._work[slot] = work
        return work, slot

    def finish(self, handle):
        """Wait for a started all-gather and return the [H*W, K] frame (every rank gets the full frame)."""
        work, slot = handle
        work.wait()
        out = self.recv[slot].index_select(0, self.gather_index)
        if out.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(out.device))
            self._read[slot] = ev
        self._busy[slot] = False
        return out

    def __call__(self, parts):
        """parts: tensors [n_local, k_i] (sum k_i == K), rows ordered like self.idx.  Returns the [H*W, K] frame on every rank."""
        return self.finish(self.start(parts))


class OrderedGather:
    """Frames rendered by several host threads (pipeline.FramesInFlight), ONE communicator: a collective must be issued in the same order on
    every rank, so the gather of frame i is started only after the gathers of frames 0 .. i-1 -- whichever thread rendered them.  submit() is
    called on the thread (and stream) that rendered frame i; it blocks only while an earlier frame's gather has not been started yet.  No
    thread ever waits for a later frame, so the turnstile cannot deadlock as long as every frame index is submitted exactly once."""

    def __init__(self, gatherer, first_frame=0):
        import threading
        self.gatherer = gatherer
        self._next = int(first_frame)
        self._cv = threading.Condition()
        self._aborted = None

    def submit(self, i, parts):
        with self._cv:
            # frame i's turn, and a free buffer pair: with F render threads and F + 1 pairs the pair of frame i is the one frame i - F - 1 used,
            # whose finish() its thread calls right after submitting frame i - 1 -- so this wait is short and cannot deadlock
            while self._aborted is None and (self._next != i or not self.gatherer.next_slot_free()):
                self._cv.wait(timeout=1.0)
            if self._aborted is not None:
                raise RuntimeError("OrderedGather: aborted because another frame's worker failed") from self._aborted
            handle = self.gatherer.start(parts)
            self._next += 1
            self._cv.notify_all()
        return handle

    def finish(self, handle):
        out = self.gatherer.finish(handle)
        with self._cv:
            self._cv.notify_all()        # a buffer pair became free
        return out

    def abort(self, error=None):
        """A worker failed: frames it would have submitted never come, so every thread waiting for its turn must give up (the peers of a
        multi-rank job then fail in the collective's timeout instead of hanging in a join)."""
        with self._cv:
            self._aborted = error if error is not None else RuntimeError("aborted")
            self._cv.notify_all()


def gather_frame(local, idx, n_max, H, W, group=None):
    """One-shot form of FrameGatherer (builds the plan every call; tests and small scripts)."""
    g = FrameGatherer(H, W, local.shape[1], local.device, group)
    assert g.n_max == n_max and torch.equal(g.idx, idx.cpu())
    return g([local])
