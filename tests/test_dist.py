"""N > 1 path on CPU: ray-tile sharding and the per-frame all-gather under gloo, world_size 2 and 3."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from palettenerf_amd import dist as pdist


def test_tile_assignment_is_a_partition_and_balanced():
    for H, W, world in ((800, 800, 8), (100, 75, 3), (33, 31, 2), (16, 16, 4)):
        owner = pdist.tile_assignment(H, W, world)
        assert owner.shape == (H * W,) and int(owner.min()) >= 0 and int(owner.max()) < world
        seen = torch.zeros(H * W, dtype=torch.int32)
        n_max = None
        for r in range(world):
            idx, n_max = pdist.shard_indices(H, W, r, world)
            assert torch.all(idx[1:] > idx[:-1])  # ascending: keeps the alive-list compaction order meaningful
            seen[idx] += 1
        assert torch.all(seen == 1)
    counts = torch.bincount(pdist.tile_assignment(800, 800, 8), minlength=8)
    assert counts.max() / counts.min() < 1.3  # 625 tiles of 32x32 over 8 ranks


def _worker(rank, world, port, H, W, K):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx, n_max = pdist.shard_indices(H, W, rank, world)
        full_ref = torch.arange(H * W * K, dtype=torch.float32).reshape(H * W, K) * 0.5 + 1.0
        frame = pdist.gather_frame(full_ref[idx].clone(), idx, n_max, H, W)
        assert torch.equal(frame, full_ref), f"rank {rank}: assembled frame differs"
        g = pdist.FrameGatherer(H, W, K, torch.device("cpu"))
        for rep in range(2):  # the static plan is reusable frame after frame
            ref = full_ref + rep
            frame = g([ref[g.idx, :3].clone(), ref[g.idx, 3:4].clone(), ref[g.idx, 4:].clone()])
            assert torch.equal(frame, ref), f"rank {rank}: FrameGatherer frame differs"
        # pipelined use (bench.py): the all-gather of frame k is finished only after frame k+1 has been packed and started
        refs = [full_ref * (rep + 2) for rep in range(4)]
        pending, done = None, []
        for ref in refs:
            handle = g.start([ref[g.idx].clone()])
            if pending is not None:
                done.append(g.finish(pending))
            pending = handle
        done.append(g.finish(pending))
        for ref, frame in zip(refs, done):
            assert torch.equal(frame, ref), f"rank {rank}: pipelined frame differs"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 70, 50), (3, 64, 96)])
def test_gather_frame_gloo(world, H, W):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, H, W, 5), nprocs=world, join=True)


def _ordered_worker(rank, world, port, H, W, K, F, n_frames):
    """Frames produced by F threads per rank in scrambled completion order; the gathers must still pair up frame by frame across ranks."""
    import threading
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = pdist.FrameGatherer(H, W, K, torch.device("cpu"), slots=F + 1)
        og = pdist.OrderedGather(g)
        base = torch.arange(H * W * K, dtype=torch.float32).reshape(H * W, K)
        out, errors = [None] * n_frames, []

        def work(k):
            try:
                prev = None
                for i in range(k, n_frames, F):
                    time.sleep(0.002 * ((i * 7 + rank * 3 + k) % 5))       # "render": threads and ranks finish their frames in different orders
                    h = og.submit(i, [(base + i)[g.idx].clone()])
                    if prev is not None:
                        out[prev[0]] = og.finish(prev[1])
                    prev = (i, h)
                if prev is not None:
                    out[prev[0]] = og.finish(prev[1])
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(F)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
            assert not t.is_alive(), "turnstile deadlocked"
        assert not errors, errors
        for i in range(n_frames):
            assert torch.equal(out[i], base + i), f"rank {rank}: frame {i} was gathered from another frame's rows"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 2), (2, 3)])
def test_ordered_gather_with_several_render_threads_gloo(world, F):
    """dist.OrderedGather: F render threads per rank share one communicator; frame i's all-gather is issued as the i-th collective on every rank
    whatever order the threads finish in, so every rank assembles frame i from frame i's shards (and nothing deadlocks)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ordered_worker, args=(world, port, 40, 36, 3, F, 12), nprocs=world, join=True)


def _slot_reuse_worker(rank, world, port, H, W, K, F, n_frames):
    """The round-2 advisor's scenario: the consumer of a frame is SLOW to finish() (the buffer pair of frame i - F - 1 is still being read
    when frame i is ready).  A pair is handed out again only after its finish(): submit() waits, start() alone refuses."""
    import threading
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = pdist.FrameGatherer(H, W, K, torch.device("cpu"), slots=F + 1)
        og = pdist.OrderedGather(g)
        base = torch.arange(H * W * K, dtype=torch.float32).reshape(H * W, K)
        out, errors = [None] * n_frames, []

        def work(k):
            try:
                prev = None
                for i in range(k, n_frames, F):
                    h = og.submit(i, [(base + i)[g.idx].clone()])
                    if prev is not None:
                        time.sleep(0.01 * ((prev[0] + rank) % 3))        # a lagging consumer: the other threads race ahead to the pair it still owns
                        out[prev[0]] = og.finish(prev[1])
                    prev = (i, h)
                if prev is not None:
                    out[prev[0]] = og.finish(prev[1])
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                og.abort(e)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(F)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
            assert not t.is_alive(), "deadlock"
        assert not errors, errors
        for i in range(n_frames):
            assert torch.equal(out[i], base + i), f"rank {rank}: frame {i} holds another frame's rows"
        # the plan alone: F + 1 gathers outstanding is the limit, one more is refused instead of overwriting unread rows
        hs = [g.start([(base + 100 + j)[g.idx].clone()]) for j in range(F + 1)]
        with pytest.raises(RuntimeError, match="not been finish"):
            g.start([base[g.idx].clone()])
        for j, h in enumerate(hs):
            assert torch.equal(g.finish(h), base + 100 + j)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,F", [(2, 2), (2, 3)])
def test_gather_buffers_are_not_reused_before_their_consumer_is_done_gloo(world, F):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_slot_reuse_worker, args=(world, port, 40, 36, 3, F, 14), nprocs=world, join=True)


def test_ordered_gather_abort_releases_the_waiting_threads():
    """A render thread that dies never submits its frames; abort() makes every thread waiting for its turn raise instead of waiting for ever
    (pipeline.FramesInFlight calls it from the failing worker).  No communicator needed: nothing is started."""
    import threading

    class NoGather:
        def next_slot_free(self):
            return True

        def start(self, parts):
            raise AssertionError("frame 1 must never start: frame 0 was not submitted")

    og = pdist.OrderedGather(NoGather())
    caught = []

    def waiter():
        try:
            og.submit(1, [])          # frame 0 belongs to the thread that failed
        except RuntimeError as e:
            caught.append(e)

    t = threading.Thread(target=waiter)
    t.start()
    og.abort(ValueError("render of frame 0 failed"))
    t.join(timeout=10)
    assert not t.is_alive() and len(caught) == 1 and isinstance(caught[0].__cause__, ValueError)
