#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (BASELINE.json: rendered samples/sec + ms/frame @800x800).

A "step" is one inference frame of the occupancy-march path over a batch of synthetic rays:
configs[1] = NeRF-synthetic-lego geometry, `-m nerf` inference, 800x800, scene S0 (SURVEY.md 8d /
Appendix B), seeded random-init field, inputs resident in HBM when the timed region starts.

N > 1: rays are sharded over ranks in interleaved 32x32 pixel tiles (no collective on the march data path); every step ends
with ONE all_gather_into_tensor (RCCL) of the packed per-ray (rgb, depth, alpha) rows, after which every rank holds the full
output.  --scaling weak (default): a step renders N views of the scene (800x800 each, azimuths spread over the orbit), i.e.
per-GPU work is fixed -- each rank gets 1/N of the tiles of EVERY view, which also balances the load.  --scaling strong: a step
is ONE 800x800 frame split N ways (latency-bound below ~2.5 ms/frame: see DESIGN.md section 4).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GRID_BYTES_PER_SAMPLE_FP32 = 12 + 16 * 8 * 2 * 4 + 32 * 4  # 1164 B: xyz + 16 levels x 8 corners x 2 x fp32 + 32 outputs
GRID_BYTES_PER_SAMPLE_FP16 = 12 + 16 * 8 * 2 * 2 + 32 * 2  # 588 B
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--model", choices=["nerf", "palette"], default="nerf")
    ap.add_argument("--density-scale", type=float, default=100.0, help="S0-opaque (trained-scene-like early termination); ~0 = translucent")
    ap.add_argument("--fp16", action="store_true", help="the reference's -O mode: autocast, half hash tables")
    ap.add_argument("--mode", choices=["compat", "device", "fused", "native"], default=None)
    ap.add_argument("--field-precision", choices=["f16x3", "fp32"], default="f16x3",
                    help="matrix path of the fused field: split-fp16 (3 MFMAs per product, ~2^-22 relative) or exact fp32 MFMA")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--ray-order", choices=["tile8", "tile4", "tile16", "morton", "rowmajor"], default="tile8", help="initial order of the alive list in the native loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-interleave", action="store_true", help="-m palette: separate hash-table lookups instead of the interleaved copy (A/B)")
    ap.add_argument("--pred-clip", action="store_true", help="-m palette with the clip-feature head (main_palette.py --pred_clip): third hash table + clip_net")
    ap.add_argument("--half-tables", action="store_true", help="native loop with fp16 hash tables and the reference's half interpolation (its --fp16 tables); MLP unchanged")
    ap.add_argument("--scene", choices=["s0", "s1"], default="s0", help="s0: dense 8^3 bricks (the headline scene); s1: sparse 4^3 bricks, the occupied box ~94 %% air")
    ap.add_argument("--dt-gamma", type=float, default=0.0, help="march step growth (0 = the lego config; 1/128 = the LLFF / 360 configs)")
    ap.add_argument("--cpu-crop", type=int, default=480, help="side of the centre crop timed on the CPU oracle")
    return ap.parse_args()


def build_model(args, device):
    from palettenerf_amd import network, raymarching, renderer, scene
    if args.model == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=args.density_scale, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(pred_clip=bool(getattr(args, "pred_clip", False))), bound=2, cuda_ray=True, density_scale=args.density_scale, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(device).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid() if args.scene == "s0" else scene.sparse_density_grid()).to(device))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.count_rendered = True
    return m


def cpu_baseline(args):
    """The oracle ("port") timed on this host, 1 thread, on a bounded centre crop of the same frame."""
    import numpy as np
    import oracle
    from oracle.facade import make_oracle_modules
    from palettenerf_amd import network, renderer, scene
    import palettenerf_amd.gridencoder as pge
    import palettenerf_amd.shencoder as psh
    torch.set_num_threads(1)
    rm, ge, sh, pu = make_oracle_modules()
    saved = (renderer.raymarching, pge.GridEncoder, psh.SHEncoder)
    renderer.raymarching, pge.GridEncoder, psh.SHEncoder = rm, ge.GridEncoder, sh.SHEncoder
    try:
        if args.model == "nerf":
            m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=args.density_scale, min_near=0.2)
        else:
            m = network.PaletteNetwork(renderer.default_opt(pred_clip=bool(getattr(args, "pred_clip", False))), bound=2, cuda_ray=True, density_scale=args.density_scale, min_near=0.2)
        scene.seed_field_(m, 0)
        grid = scene.brick_density_grid() if args.scene == "s0" else scene.sparse_density_grid()
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(oracle.packbits(grid, 0.5)))
        m.eval()
        m.count_rendered = True
        H = W = args.res
        pose = torch.from_numpy(scene.lookat_pose())[None]
        ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
        c = args.cpu_crop
        ys = torch.arange(H // 2 - c // 2, H // 2 + c // 2)
        idx = (ys[:, None] * W + ys[None, :]).reshape(-1)
        ro, rd = ro[:, idx].contiguous(), rd[:, idx].contiguous()
        t0 = time.perf_counter()
        with torch.no_grad():
            r = m.render(ro, rd, perturb=False, dt_gamma=args.dt_gamma, max_steps=1024, T_thresh=1e-4, **({"gui_mode": False} if args.model == "palette" else {}))
        dt = time.perf_counter() - t0
        n = int(r["rendered"].item())
        # BASELINE configs[0] beside it: the reference's CPU-runnable case, NeRFRenderer.run at 400x400 with --num_steps 512 --upsample_steps 0
        # (main_nerf.py:31-32), a quarter of one max_ray_batch of 4096 rays (160 such pieces make the frame), 8 threads for the torch part as the reference's scripts set
        uniform = None
        if args.model == "nerf":
            n_thr = min(8, os.cpu_count() or 1)       # the reference's scripts pin OMP_NUM_THREADS=8 (scripts/run_blender.sh:47)
            torch.set_num_threads(n_thr)
            mu = network.NeRFNetwork(bound=2, cuda_ray=False, density_scale=args.density_scale, min_near=0.2)
            scene.seed_field_(mu, 0)
            mu.eval()
            ro0, rd0 = scene.get_rays(pose, scene.intrinsics_from_fov(400, 400), 400, 400)
            mid = 400 * 200 - 512
            t0 = time.perf_counter()
            with torch.no_grad():
                mu.run(ro0[:, mid:mid + 1024].contiguous(), rd0[:, mid:mid + 1024].contiguous(), num_steps=512, upsample_steps=0, perturb=False)
            du = time.perf_counter() - t0
            uniform = {"value": 1024 * 512 / du, "unit": "evaluated samples/s", "cores": n_thr, "ms_per_400x400_frame_extrapolated": du * 160 * 1e3,
                       "sample": f"1024 rays x 512 uniform samples (1/160) of a 400x400 frame ({du:.1f} s; C oracle encoders 1 thread + torch MLP on {n_thr} threads; host has {os.cpu_count()} cores -- all of them made this piece slower)"}
            torch.set_num_threads(1)
    finally:
        renderer.raymarching, pge.GridEncoder, psh.SHEncoder = saved
    return {"uniform_path_config0": uniform, "value": n / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"centre {c}x{c} crop of the {H}x{W} frame ({idx.numel()} rays, {n} rendered samples, {dt:.1f} s; -m {args.model}, C oracle ops + torch CPU MLP, 1 thread; "
                      f"host has {os.cpu_count()} cores)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("PNR_BENCH_FORCE_DIST") == "1"  # the latter exercises the RCCL path on a 1-GPU box
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from palettenerf_amd import _torch_glue, dist as pdist, scene
    m = build_model(args, device)
    mode = args.mode or "native"     # --fp16: the native loop under autocast looks the tables up as fp16 (the reference's -O tables), the field stays fp32-accurate
    m.march_mode = "device" if mode == "fused" else mode
    if mode in ("fused", "native") and args.model == "nerf" and (not args.fp16 or mode == "native"):
        from palettenerf_amd.fused import NeRFFieldFused
        m.fused_field = True
        m._fused = NeRFFieldFused(m)
        m._fused.precision = 0 if args.field_precision == "fp32" else 1
    if mode in ("fused", "native") and args.model == "palette" and (not args.fp16 or mode == "native"):
        m.fused_field = True
        if args.no_interleave:
            from palettenerf_amd.fused import PaletteFieldFused
            m._fused = PaletteFieldFused(m)
            m._fused.interleave_tables = False
    H = W = args.res
    n_views = world if args.scaling == "weak" else 1
    import numpy as np
    poses = torch.from_numpy(np.stack([scene.lookat_pose(azimuth_deg=45.0 + 360.0 * v / n_views) for v in range(n_views)]))
    ro, rd = scene.get_rays(poses, scene.intrinsics_from_fov(H, W), H, W)          # [n_views, H*W, 3]
    ro, rd = ro.reshape(1, n_views * H * W, 3), rd.reshape(1, n_views * H * W, 3)  # the views stacked vertically: one (n_views*H) x W image
    VH = n_views * H
    idx, n_max = pdist.shard_indices(VH, W, rank, world)
    ro, rd = ro[:, idx].contiguous().to(device), rd[:, idx].contiguous().to(device)
    kw = dict(perturb=False, dt_gamma=args.dt_gamma, max_steps=1024, T_thresh=1e-4)
    if args.model == "palette":
        kw["gui_mode"] = False

    gatherer = pdist.FrameGatherer(VH, W, 5, device) if use_dist else None
    pending = []   # all-gather of the previous frame, still in flight

    def frame():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=args.fp16):
            r = m.render(ro, rd, **kw)
        if use_dist:  # one all-gather of the packed (rgb, depth, alpha) rows; every rank ends up with the full frame.  It is started here
            # and completed after the NEXT frame has been rendered (or at the end of the timed region): communication overlaps compute
            handle = gatherer.start([r["image"][0], r["depth"][0][:, None], r["weights_sum"][:, None]])
            full = gatherer.finish(pending.pop()) if pending else None
            pending.append(handle)
            return r, full
        return r, None

    for _ in range(args.warmup):
        frame()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    prof = _torch_glue.profile_kernels(["pnr_grid_encode_forward"]) if rank == 0 else None
    rendered = torch.zeros(1, dtype=torch.int64, device=device)
    rendered_host = 0
    rows = 0
    native_ms, native_launches = 0.0, 0
    native_rows = 0
    timed_native = m.march_mode == "native" and rank == 0
    if timed_native and getattr(m, "_fused", None) is None:
        from palettenerf_amd.fused import PaletteFieldFused
        m._fused = PaletteFieldFused(m)
    if m.march_mode == "native" and args.ray_order != "rowmajor":
        from palettenerf_amd.fused import tile_ray_order
        if getattr(m, "_fused", None) is None:
            from palettenerf_amd.fused import PaletteFieldFused
            m._fused = PaletteFieldFused(m)
        m._fused.table_half = bool(args.half_tables)
        m._fused.ray_order = tile_ray_order(idx, W, {"tile8": 8, "tile4": 4, "tile16": 16, "morton": 0}[args.ray_order]).to(device)   # idx: row-major pixel ids (of the stacked views) this rank renders
        for _ in range(2):
            frame()                                               # re-warm with the final ordering
        torch.cuda.synchronize()
    if m.march_mode == "native":   # untimed, on every rank (a frame holds a collective): one frame with the in-library HIP-event timing on,
        fused = getattr(m, "_fused", None)   # so that the events exist before the timed region
        if fused is not None:
            fused.time_grid_kernel = True
        frame()
        if fused is not None:
            fused.time_grid_kernel = False
        torch.cuda.synchronize()
    if use_dist:
        if pending:
            gatherer.finish(pending.pop())
        dist.barrier()
        torch.cuda.synchronize()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # diagnostics only (no sync inside the loop)
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        # HIP events around every grid-encode launch cost ~6 us each (two per iteration): instrument the launches of the
        # FIRST timed step only, so the measurement lives inside the timed region without distorting it
        if timed_native:
            m._fused.time_grid_kernel = i == 0
        r, _full = frame()
        if r["rendered"].is_cuda:
            rendered += r["rendered"]
        else:
            rendered_host += int(r["rendered"])   # native loop: the count is already on the host (it came back with the control block)
        rows += r["n_samples"]
        if timed_native and i == 0:
            native_ms, native_launches, native_rows = r.get("grid_ms", 0.0), r.get("grid_launches", 0), r["n_samples"]
        step_ev[i + 1].record()
    if pending:
        _full = gatherer.finish(pending.pop())   # the last frame's all-gather completes inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rendered += rendered_host
    _torch_glue.profile_kernels(None)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rendered, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rendered = int(rendered.item())

    if rank == 0:
        per_sample = GRID_BYTES_PER_SAMPLE_FP16 if args.fp16 else GRID_BYTES_PER_SAMPLE_FP32
        if args.half_tables or (args.fp16 and m.march_mode == "native"):
            per_sample = 12 + 16 * 8 * 2 * 2 + 32 * 4   # half rows gathered, fp32 encoder output written
        n_tables = 1 if args.model == "nerf" else (3 if args.pred_clip else 2)  # palette: encoder + encoder_palette (+ encoder_clip with --pred-clip)
        launches = prof["pnr_grid_encode_forward"]
        k_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in launches)
        k_units = sum(u for _, _, u in launches)
        n_launches = len(launches)
        kernel_name = "k_grid_fwd (pnr_grid_encode_forward)"
        if m.march_mode == "native":  # events recorded inside pnr_nerf_render_frame around every k_frame_grid launch
            k_ms, k_units, n_launches = native_ms, native_rows * n_tables, native_launches
            kernel_name = "k_frame_grid (device-driven frame loop)" if args.model == "nerf" else "k_frame_grid_pair (device-driven frame loop, encoder + encoder_palette interleaved)"
        achieved = (k_units * per_sample) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None  # HBM-side bytes per launch from the committed PMC passes of this exact workload (profiles/r01_traffic.json)
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if m.march_mode == "native" and H == 800 and args.density_scale == 100.0 and not args.fp16 and not args.half_tables and world == 1 and os.path.exists(tpath):
            t = json.load(open(tpath)).get(args.model)
            if t:
                pair = args.model == "palette"   # the interleaved pair kernel is ONE launch for both tables
                traffic = t["traffic_bytes_per_launch"] / (1 if pair else n_tables)
        raw_steps = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
        slowest = max(range(args.steps), key=lambda i: raw_steps[i])
        per_step = sorted(raw_steps)
        out = {
            "metric": "rendered_samples_per_sec", "value": total_rendered / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1], "slowest_step": slowest}, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f16" if args.fp16 else "f32", "data": "synthetic",
            "config": {"workload": f"configs[{1 if args.model == 'nerf' else 2}]: NeRF-synthetic lego geometry (scene {args.scene.upper()}), -m {args.model} inference, {H}x{W}, {n_views} view(s)/step",
                       "rays_per_step": n_views * H * W, "rendered_samples_per_step": total_rendered // args.steps,
                       "evaluated_rows_per_step_rank0": rows // args.steps, "density_scale": args.density_scale, "dt_gamma": args.dt_gamma, "march_mode": m.march_mode, "fused_field": bool(getattr(m, "fused_field", False)), "field_precision": args.field_precision, "half_tables": bool(args.half_tables or (args.fp16 and m.march_mode == "native")), "ray_order": args.ray_order,
                       "parallelism": f"32x32 ray tiles of {n_views} view(s) round-robin over {world} GPUs + one all_gather/step" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches": n_launches,
                         "avg_launch_ms": k_ms / max(1, n_launches), "avg_rows_per_launch": k_units / max(1, n_launches),
                         "algorithmic_bytes_per_row": per_sample,
                         "algorithmic_bytes_per_launch": per_sample * k_units / max(1, n_launches)},
        }
        if achieved > HBM_PEAK_GBS:
            out["roofline"]["note"] = ("algorithmic bytes per second exceed the HBM peak: table rows are re-used out of L2 / Infinity Cache "
                                       "(the HBM-side bytes per launch are in `traffic`)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        try:   # RCCL prints its version banner through C stdio, which is flushed at exit: push it out first so the JSON line is the last one
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
