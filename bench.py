#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (BASELINE.json: rendered samples/sec + ms/frame @800x800; PSNR vs reference).

A "step" is one inference frame of the occupancy-march path over a batch of synthetic rays, inputs resident in HBM when the
timed region starts.  Step i renders pose i of a camera path (the camera MOVES: 3 degrees of azimuth per step on the lego orbit,
one pose of the 120-pose ellipse per step on the garden path), so the frame loop's iteration prediction, the occupancy mip and
the tile-ordered alive list all face changing frames.

Workloads (--workload):
  lego          configs[1]: NeRF-synthetic-lego geometry (scene S0), `-m nerf` inference, 800x800, dt_gamma 0        (default at --gpus 1)
  lego_palette  configs[2]: the same frame through the PaletteNeRF model (`-m palette`)
  garden        configs[4]: Mip-360-garden-like scene S2, `-m palette` video render, 1297x840, dt_gamma 1/128, the 120-pose
                ellipse path of scripts/llff2nerf.py:104-106                                                     (default at --gpus N > 1)
N > 1 (`--gpus N`): one process per GPU.  Started under torch.distributed.run (the driver's way) the ranks come from the
environment; started plainly, this script spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself --
before anything touches the GPU -- and relays its output.  Rays are sharded over ranks in interleaved 32x32 pixel tiles (no
collective on the march data path); every step ends with ONE all_gather_into_tensor (RCCL) of the packed per-ray rows the
reference's video writer consumes (rgb, depth, alpha; + view-dependent colour, basis images and basis weights for the palette
model).  Defaults at N > 1 (round 5): `--workload garden --scaling strong` -- north_star's question, configs[4]: ONE garden frame's rays split N ways;
the line then carries every rank's own shard render ms, the all-gather alone (HIP events) and the same frame rendered whole by one GPU inside
the same job (`config.strong_speedup_vs_single_gpu_in_this_job`: the driver's N = 1 run is configs[1], another workload), and `extra.weak` holds the
weak-scaling leg of rounds 1-4 (N lego views per step, per-GPU work fixed).  Explicit --workload / --model / --scaling override the defaults;
`--scaling weak`: a step renders N views; `--scaling strong`: ONE frame split N ways.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRID_BYTES_PER_SAMPLE_FP32 = 12 + 16 * 8 * 2 * 4 + 32 * 4  # 1164 B: xyz + 16 levels x 8 corners x 2 x fp32 + 32 outputs
GRID_BYTES_PER_SAMPLE_FP16 = 12 + 16 * 8 * 2 * 2 + 32 * 2  # 588 B
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    "lego": dict(config=1, model="nerf", scene="s0", dt_gamma=0.0, poses=120,
                 label="configs[1]: NeRF-synthetic lego geometry (scene S0), -m nerf inference"),
    "lego_palette": dict(config=2, model="palette", scene="s0", dt_gamma=0.0, poses=120,
                         label="configs[2]: NeRF-synthetic lego geometry (scene S0), -m palette inference"),
    "garden": dict(config=4, model="palette", scene="s2", dt_gamma=1.0 / 128, poses=120,
                   label="configs[4]: Mip-360 garden-like scene S2, -m palette video render (120-pose ellipse path)"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--res", type=int, default=800, help="lego workloads: frame side")
    ap.add_argument("--model", choices=["nerf", "palette"], default=None, help="shorthand: --model palette == --workload lego_palette")
    ap.add_argument("--density-scale", type=float, default=100.0, help="opaque, trained-scene-like early termination; ~0 = translucent")
    ap.add_argument("--fp16", action="store_true", help="the reference's -O mode: autocast, half hash tables")
    ap.add_argument("--mode", choices=["compat", "device", "fused", "native"], default=None)
    ap.add_argument("--field-precision", choices=["f16x3", "fp32", "f16x2"], default="f16x3",
                    help="matrix path of the fused field: split-fp16 (3 MFMAs per product, ~2^-22 relative) or exact fp32 MFMA")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: weak at --gpus 1 (a step = one frame), STRONG at --gpus N > 1 (north_star: the rays of ONE frame sharded over the GPUs)")
    ap.add_argument("--dist-default", action="store_true",
                    help="take the N > 1 defaults (--workload garden --scaling strong = configs[4]) whatever the rank count: with PNR_BENCH_FORCE_DIST=1 "
                         "this is the N > 1 line over a one-rank RCCL communicator (tests)")
    ap.add_argument("--ray-order", choices=["tile8", "tile4", "tile16", "morton", "rowmajor"], default="tile8", help="initial order of the alive list in the native loop")
    ap.add_argument("--static-pose", action="store_true", help="every step renders pose 0 (round-1 behaviour; A/B against the moving camera)")
    ap.add_argument("--pose-step-deg", type=float, default=3.0, help="lego orbit: degrees of azimuth per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra driver-observed legs (exact-fp32 field, palette model)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (FETCH_SIZE / WRITE_SIZE of the lookup kernel)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--one-call-per-frame", action="store_true",
                    help="the timed frames as m.render() calls (the host turns around between two frames: rounds 1-5's loop) instead of the prepare / launch / finish queue")
    ap.add_argument("--shared-stream", action="store_true",
                    help="with --main-frames-in-flight F > 1: all F handles enqueue into one stream (frames back to back without the host's gap, kernels never overlap)")
    ap.add_argument("--main-frames-in-flight", type=int, default=1, metavar="F",
                    help="render the TIMED steps themselves with F frames in flight (any --gpus N: the ranks' all-gathers are issued in frame order through one "
                         "communicator).  Default 1: a step is one frame at a time, ms_per_step is a frame's latency and the roofline launches run alone")
    ap.add_argument("--frames-in-flight", type=int, nargs="*", default=[2, 3], metavar="F",
                    help="extra leg (1 GPU): the same camera path with F frames in flight, one host thread + stream + fused-field object each (video throughput)")
    ap.add_argument("--shard-emulation", type=int, default=0, metavar="S",
                    help="1 GPU: also render each of the S tile shards of pose 0 on its own and report max / mean shard ms (load balance of the S-GPU split)")
    ap.add_argument("--emulate-shard", default=None, metavar="R/N",
                    help="1 GPU, no communicator: render only rank R's share of an N-way tile split of every frame (what rank R of `--gpus N --scaling strong` renders); "
                         "the PMC child passes of an N > 1 run use it so that roofline.traffic is per launch of the SAME shard-sized launches as roofline.achieved")
    ap.add_argument("--no-interleave", action="store_true", help="-m palette: separate hash-table lookups instead of the interleaved copy (A/B)")
    ap.add_argument("--pred-clip", action="store_true", help="-m palette with the clip-feature head (main_palette.py --pred_clip): third hash table + clip_net")
    ap.add_argument("--num-basis", type=int, default=4)
    ap.add_argument("--half-tables", action="store_true", help="native loop with fp16 hash tables and the reference's half interpolation (its --fp16 tables); MLP unchanged")
    ap.add_argument("--scene", choices=["s0", "s1", "s2"], default=None, help="override the workload's scene (s1: sparse 4^3 bricks, the occupied box ~94 %% air)")
    ap.add_argument("--dt-gamma", type=float, default=None, help="override the workload's march step growth")
    ap.add_argument("--cpu-crop", type=int, default=400, help="side of the centre crop rendered by the CPU oracle (baseline + PSNR)")
    args = ap.parse_args(argv)
    # N > 1 answers north_star's multi-GPU question by default: configs[4] -- ONE garden video frame (1297 x 840, -m palette), its rays sharded over
    # the GPUs in 32 x 32 tiles, one RCCL all-gather of the per-ray rows per frame (strong scaling; the reference's dormant collective sites:
    # palette/utils.py:802-817).  N = 1 stays configs[1] (the configuration the metric is quoted on).  Explicit --workload / --model / --scaling win.
    many = args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.dist_default
    args.defaulted_for_ranks = bool(many and args.workload is None and args.model is None and args.scaling is None)
    if args.workload is None:
        args.workload = "lego_palette" if args.model == "palette" else ("garden" if args.defaulted_for_ranks else "lego")
    if args.scaling is None:
        args.scaling = "strong" if (many and args.workload == "garden") else "weak"
    wl = dict(WORKLOADS[args.workload])
    if args.scene is not None:
        wl["scene"] = args.scene
    if args.dt_gamma is not None:
        wl["dt_gamma"] = args.dt_gamma
    if args.workload == "garden":
        from palettenerf_amd import scene as _scene
        wl["H"], wl["W"] = _scene.GARDEN_H, _scene.GARDEN_W
    else:
        wl["H"] = wl["W"] = args.res
    args.wl = wl
    args.model = wl["model"]
    args.shard = None
    if args.emulate_shard:
        r, n = (int(v) for v in args.emulate_shard.split("/"))
        if not 0 <= r < n:
            ap.error("--emulate-shard R/N needs 0 <= R < N")
        args.shard = (r, n)
    return args


def core_argv(args):
    """The flags that define the headline workload (what a child pass of this script must repeat)."""
    a = ["--workload", args.workload, "--scaling", args.scaling, "--res", str(args.res), "--density-scale", repr(float(args.density_scale)), "--field-precision", args.field_precision,
         "--ray-order", args.ray_order, "--pose-step-deg", repr(float(args.pose_step_deg)), "--num-basis", str(args.num_basis)]
    for flag, on in (("--fp16", args.fp16), ("--one-call-per-frame", args.one_call_per_frame), ("--static-pose", args.static_pose), ("--no-interleave", args.no_interleave), ("--pred-clip", args.pred_clip),
                     ("--half-tables", args.half_tables)):
        if on:
            a.append(flag)
    if args.mode:
        a += ["--mode", args.mode]
    if args.scene:
        a += ["--scene", args.scene]
    if args.dt_gamma is not None:
        a += ["--dt-gamma", repr(float(args.dt_gamma))]
    return a


def spawn_ranks(args, argv, script=None):
    """`bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process (this process has not
    touched the GPU and never will), relay its output, exit with its code.  `script`: what the ranks run (tests; default this file)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def world_from_env(args, argv):
    """(world, rank, local_rank), or SystemExit after having run the ranks as children."""
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            import torch   # device_count() does not initialise the GPU
            if torch.cuda.device_count() < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
            raise SystemExit(spawn_ranks(args, argv))
        return 1, 0, 0
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


# ------------------------------------------------------------------------------------------------ scene / model / rays
def density_grid_of(name):
    from palettenerf_amd import scene
    return {"s0": scene.brick_density_grid, "s1": scene.sparse_density_grid, "s2": scene.garden_density_grid}[name]()


def make_model(args, model_kind, cuda_ray=True):
    from palettenerf_amd import network, renderer
    if model_kind == "nerf":
        return network.NeRFNetwork(bound=2, cuda_ray=cuda_ray, density_scale=args.density_scale, min_near=0.2)
    opt = renderer.default_opt(pred_clip=bool(args.pred_clip), num_basis=int(args.num_basis))
    return network.PaletteNetwork(opt, bound=2, cuda_ray=cuda_ray, density_scale=args.density_scale, min_near=0.2)


def build_model(args, device, model_kind=None, field_precision=None):
    import torch
    from palettenerf_amd import raymarching, scene
    model_kind = model_kind or args.model
    m = make_model(args, model_kind)
    scene.seed_field_(m, 0)
    m = m.to(device).eval()
    m.density_grid.copy_(torch.from_numpy(density_grid_of(args.wl["scene"])).to(device))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.count_rendered = True
    mode = args.mode or "native"     # --fp16: the native loop under autocast looks the tables up as fp16 (the reference's -O tables), the field stays fp32-accurate
    m.march_mode = "device" if mode == "fused" else mode
    prec = field_precision or args.field_precision
    if mode in ("fused", "native") and (not args.fp16 or mode == "native"):
        from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
        m.fused_field = True
        m._fused = NeRFFieldFused(m) if model_kind == "nerf" else PaletteFieldFused(m)
        m._fused.precision = {"fp32": 0, "f16x3": 1, "f16x2": 2}[prec]   # f16x2: NeRF field only (the PaletteNeRF field runs it as f16x3)
        m._fused.table_half = bool(args.half_tables)
        if model_kind == "palette" and args.no_interleave:
            m._fused.interleave_tables = False
    return m


def pose_of(args, step, view=0, n_views=1):
    from palettenerf_amd import scene
    if args.static_pose:
        step = 0
    if args.workload == "garden":
        return scene.garden_orbit_pose(step + (view * scene.GARDEN_POSES) // n_views)
    return scene.lookat_pose(azimuth_deg=45.0 + args.pose_step_deg * step + 360.0 * view / n_views)


def intrinsics_of(args):
    from palettenerf_amd import scene
    wl = args.wl
    return scene.garden_intrinsics(wl["H"], wl["W"]) if args.workload == "garden" else scene.intrinsics_from_fov(wl["H"], wl["W"])


class RayBank:
    """This rank's rays of every distinct step of the camera path, generated on the device (pnr_get_rays) BEFORE the timed
    region: the benchmark's inputs are resident in HBM, as the brief asks.  Views of a step are stacked vertically into one
    (n_views * H) x W image whose 32x32 tiles are dealt round-robin to the ranks."""

    def __init__(self, args, n_views, idx, device):
        self.args, self.n_views, self.device = args, n_views, device
        self.idx = idx.to(device)
        self.n_steps = 1 if args.static_pose else args.wl["poses"]
        self.cache = {}

    def get(self, step):
        import numpy as np
        import torch
        from palettenerf_amd import rays
        k = step % self.n_steps
        if k not in self.cache:
            H, W = self.args.wl["H"], self.args.wl["W"]
            poses = torch.from_numpy(np.stack([pose_of(self.args, k, v, self.n_views) for v in range(self.n_views)])).to(self.device)
            ro, rd = rays.rays_from_indices(poses, intrinsics_of(self.args), H, W, None)        # [n_views, H*W, 3]
            ro = ro.reshape(1, -1, 3)[:, self.idx].contiguous()
            rd = rd.reshape(1, -1, 3)[:, self.idx].contiguous()
            self.cache[k] = (ro, rd)
        return self.cache[k]


# ------------------------------------------------------------------------------------------------ CPU legs (oracle = checker + baseline)
def cpu_baseline(args, crop_rays):
    """The oracle ("port": C restatement of the kernels + torch CPU MLPs under this repo's mirror of the reference's renderer) on a
    bounded centre crop of pose 0 of the same workload: once on 1 thread, once on all cores (OpenMP over rays / samples in the C ops,
    torch threads for the MLPs).  Returns the baseline record and the crop's oracle image / weights (the PSNR reference)."""
    import torch
    import oracle
    from oracle import orc
    from oracle.facade import make_oracle_modules
    from palettenerf_amd import renderer, scene
    import palettenerf_amd.gridencoder as pge
    import palettenerf_amd.shencoder as psh
    ro, rd = crop_rays
    rm, ge, sh, pu = make_oracle_modules()
    saved = (renderer.raymarching, pge.GridEncoder, psh.SHEncoder)
    renderer.raymarching, pge.GridEncoder, psh.SHEncoder = rm, ge.GridEncoder, sh.SHEncoder
    cores = os.cpu_count() or 1
    # "all cores": OpenMP threads over rays / samples in the C ops; the torch MLP batches (<= 160 k rows x 64) stop scaling long before a 256-core
    # host is full (oversubscribed GEMM threads made this leg 15x SLOWER than one thread), so torch gets at most 32 threads
    omp_threads, torch_threads = min(cores, 64), min(cores, 32)
    legs = {}
    try:
        m = make_model(args, args.model)
        scene.seed_field_(m, 0)
        grid = density_grid_of(args.wl["scene"])
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(oracle.packbits(grid, 0.5)))
        m.eval()
        m.count_rendered = True
        kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
        if args.model == "palette":
            kw["gui_mode"] = False
        ref = None
        for name, variant, threads, tthreads in (("1_thread", "", 1, 1), ("all_cores", "omp", omp_threads, torch_threads)):
            prev = orc.use_variant(variant)
            orc.set_threads(threads)
            torch.set_num_threads(tthreads)
            t0 = time.perf_counter()
            with torch.no_grad():
                r = m.render(ro, rd, **kw)
            dt = time.perf_counter() - t0
            orc.use_variant(prev)
            n = int(r["rendered"].item())
            legs[name] = {"value": n / dt, "unit": "samples/s", "cores": threads, "torch_threads": tthreads, "host_cores": cores, "seconds": dt}
            if ref is None:
                ref = r
            else:   # the C ops are bit-identical on any thread count (tests/test_oracle.py); torch's CPU GEMMs block differently per thread count
                legs[name]["max_abs_rgb_vs_1_thread"] = float((r["image"] - ref["image"]).abs().max())
        torch.set_num_threads(1)
    finally:
        renderer.raymarching, pge.GridEncoder, psh.SHEncoder = saved
    # BASELINE configs[0] beside it -- the reference's CPU-runnable case, "pure-PyTorch raymarching on CPU (no CUDA ext)": NeRFRenderer.run at 400x400 with
    # --num_steps 512 --upsample_steps 0 (main_nerf.py:31-32), hash-grid and SH encoders written with torch ops (oracle/torch_encoders.py; the reference ships no
    # torch hash grid), nn.Linear fp32, torch threads = min(host cores, 64).  A bounded piece: 512 rays of one max_ray_batch of 4096 (320 such pieces make the frame).
    uniform = None
    if args.model == "nerf":
        from oracle.torch_encoders import TorchGridEncoder, TorchSHEncoder
        saved2 = (renderer.raymarching, pge.GridEncoder, psh.SHEncoder)
        renderer.raymarching, pge.GridEncoder, psh.SHEncoder = rm, TorchGridEncoder, TorchSHEncoder    # (run() takes near / far from the ray-box op: the C oracle's)
        n_thr = min(cores, 64)
        try:
            torch.set_num_threads(n_thr)
            mu = make_model(args, "nerf", cuda_ray=False)
            scene.seed_field_(mu, 0)
            mu.eval()
            pose = torch.from_numpy(scene.lookat_pose())[None]
            ro0, rd0 = scene.get_rays(pose, scene.intrinsics_from_fov(400, 400), 400, 400)
            mid, n_rays = 400 * 200 - 256, 512
            with torch.no_grad():
                mu.run(ro0[:, mid:mid + 32].contiguous(), rd0[:, mid:mid + 32].contiguous(), num_steps=512, upsample_steps=0, perturb=False)   # warm the thread pool
                t0 = time.perf_counter()
                mu.run(ro0[:, mid:mid + n_rays].contiguous(), rd0[:, mid:mid + n_rays].contiguous(), num_steps=512, upsample_steps=0, perturb=False)
            du = time.perf_counter() - t0
            uniform = {"value": n_rays * 512 / du, "unit": "evaluated samples/s", "cores": n_thr, "host_cores": cores, "ms_per_400x400_frame_extrapolated": du * (160000 / n_rays) * 1e3,
                       "sample": f"{n_rays} rays x 512 uniform samples (1/{160000 // n_rays}) of a 400x400 frame ({du:.1f} s; pure-torch hash grid + SH + nn.Linear, fp32, {n_thr} torch threads)"}
        finally:
            renderer.raymarching, pge.GridEncoder, psh.SHEncoder = saved2
            torch.set_num_threads(1)
    # The reference's OWN kernels on this GPU, beside the CPU port: raymarching.cu and shencoder.cu of /root/reference compiled unmodified for gfx950
    # (oracle/ref_build.py: build_hip -> oracle/_ref/ref_*.so, built where the reference checkout is, loaded here) under the per-op loop -- the
    # reference's march / composite / SH kernels, its Python's allocations (three zero-fills per march call), torch MLPs, host-side compaction.
    # The hash grid is this repository's in both runs (gridencoder.cu does not compile for HIP).  A baseline, like the CPU figures: never `value`.
    ref_gpu = None
    try:
        from oracle import ref_ops
        if ref_ops.available() and torch.cuda.is_available():
            gdev = torch.device("cuda", torch.cuda.current_device())
            cargs = argparse.Namespace(**vars(args))
            cargs.mode, cargs.fp16, cargs.half_tables = "compat", False, False
            H_, W_ = args.wl["H"], args.wl["W"]
            pose = torch.from_numpy(pose_of(args, 0)[None])
            fro, frd = scene.get_rays(pose, intrinsics_of(args), H_, W_)
            fro, frd = fro.to(gdev), frd.to(gdev)
            gkw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
            if args.model == "palette":
                gkw["gui_mode"] = False

            def run(n_frames):
                mg = build_model(cargs, gdev, args.model)
                with torch.no_grad():
                    r = mg.render(fro, frd, **gkw)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n_frames):
                        r = mg.render(fro, frd, **gkw)
                    torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n_frames * 1e3, r

            ours_ms, ours_r = run(3)
            with ref_ops.swapped_in():
                ref_ms, ref_r = run(3)
            ref_gpu = {"ms_per_frame": ref_ms, "ms_per_frame_this_repository_same_loop": ours_ms, "frames": 3,
                       "value": int(ref_r["rendered"].sum()) / (ref_ms * 1e-3), "unit": "samples/s", "kind": "reference",
                       "rendered_equal": int(ref_r["rendered"].sum()) == int(ours_r["rendered"].sum()),
                       "max_abs_rgb_between_the_two": float((ref_r["image"] - ours_r["image"]).abs().max()),
                       "what": f"{H_}x{W_} -m {args.model} frame, pose 0, per-op loop: the reference's raymarching.cu / shencoder.cu kernels (compiled for gfx950 by torch's hipify + hipcc) "
                               "against this repository's per-op kernels under the very same loop; hash grid and MLPs identical in both"}
    except Exception as e:      # noqa: BLE001 -- a baseline that cannot run is reported, never fatal
        ref_gpu = {"error": repr(e)}
    one, allc = legs["1_thread"], legs["all_cores"]
    n = int(ref["rendered"].item())
    what = (f"centre {args.cpu_crop}x{args.cpu_crop} crop of pose 0 of the {args.wl['H']}x{args.wl['W']} frame ({ro.shape[1]} rays, {n} rendered samples; -m {args.model}, "
            "C oracle ops + torch CPU MLP")
    rec = {"value": allc["value"], "unit": "samples/s", "cores": allc["cores"], "host_cores": cores, "kind": "port",
           "sample": what + f", {allc['seconds']:.1f} s on {allc['cores']} OpenMP threads + {allc['torch_threads']} torch threads of a {cores}-core host)",
           "one_thread": {"value": one["value"], "unit": "samples/s", "cores": 1, "seconds": one["seconds"], "sample": what + ", 1 thread)"},
           "all_cores_max_abs_rgb_vs_1_thread": allc.get("max_abs_rgb_vs_1_thread"), "uniform_path_config0": uniform,
           "reference_kernels_mi355x": ref_gpu}
    return rec, ref


def crop_indices(H, W, c):
    import torch
    c = min(c, H, W)
    ys = torch.arange(H // 2 - c // 2, H // 2 - c // 2 + c)
    xs = torch.arange(W // 2 - c // 2, W // 2 - c // 2 + c)
    return (ys[:, None] * W + xs[None, :]).reshape(-1)


# ------------------------------------------------------------------------------------------------ the GPU side
def gather_parts(args, r, nb):
    """Per-ray rows the reference's video writer consumes (nerf/utils.py:716-740; palette/utils.py:993-1078): rgb, depth, alpha
    (+ view_dep_rgb, basis_rgb, basis_acc for the palette model)."""
    parts = [r["image"][0], r["depth"][0][:, None], r["weights_sum"][:, None]]
    if args.model == "palette":
        parts += [r["view_dep_rgb"][0], r["basis_rgb"][0], r["basis_acc"][0]]
    return parts


def timed_frames(m, bank, kw, steps, fp16, first_step=0):
    """`steps` frames back to back on one model without gather; returns (ms per step, rendered per step)."""
    import gc
    import torch
    rendered = 0
    for i in range(steps):
        bank.get(first_step + i)
    torch.cuda.synchronize()
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()      # see main(): no interpreter GC pass inside a timed region
    try:
        t0 = time.perf_counter()
        for i in range(steps):
            ro, rd = bank.get(first_step + i)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
                r = m.render(ro, rd, **kw)
            rendered += int(r["rendered"].sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    return dt / steps * 1e3, rendered // steps


def timed_frames_median(m, bank, kw, steps, first_step=0):
    """Per-frame device time between events recorded behind each frame (no sync inside the loop), median and mean over `steps` frames.  The shard
    emulation uses the median: a 2.8 ms shard frame is ~90 launches deep, and one host hiccup among five frames (another tenant's process on the
    box's CPU) moves a mean by a third."""
    import gc
    import torch
    for i in range(steps):
        bank.get(first_step + i)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    rendered = 0
    torch.cuda.synchronize()
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        evs[0].record()
        for i in range(steps):
            ro, rd = bank.get(first_step + i)
            with torch.no_grad():
                r = m.render(ro, rd, **kw)
            rendered += int(r["rendered"].sum())
            evs[i + 1].record()
        torch.cuda.synchronize()
    finally:
        if gc_was_on:
            gc.enable()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return ms[steps // 2], sum(ms) / steps, rendered // steps


def timed_frames_queue(m, bank, kw, steps, first_step=0):
    """`steps` frames through pipeline.render_queue (model.render_prepare / render_launch / render_finish over pnr_*_render_frame_submit / _finish): one frame
    on the device at a time, the host's work for frame i + 1 done under frame i's kernels.  Per-frame time = host time between two consecutive frames'
    completions (a completion is the return of render_finish: the frame's read-back has arrived); median and mean over `steps` frames, rendered per frame."""
    import gc
    import torch
    from palettenerf_amd.pipeline import render_queue
    for i in range(steps + 1):
        bank.get(first_step + i)
    done, rendered = [], [0]
    torch.cuda.synchronize()
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()

    def consume(i, r):
        done.append(time.perf_counter())
        rendered[0] += int(r["rendered"].sum())

    try:
        render_queue(m, lambda i: bank.get(first_step + i), steps + 1, consume=consume, **kw)
        torch.cuda.synchronize()
    finally:
        if gc_was_on:
            gc.enable()
    ms = sorted((done[i + 1] - done[i]) * 1e3 for i in range(steps))
    return ms[steps // 2], sum(ms) / steps, rendered[0] // (steps + 1)


# ------------------------------------------------------------------------------------------------ configs[3]: the training step
def make_training_step(model_kind, rays, device, fp16=False, torch_adam=False, torch_loss=False):
    """configs[3]-shaped training step (main_palette.py:223 / palette/utils.py:481 on LLFF-like input): `rays` random rays per step from a
    forward-facing 17-camera rig over the slab scene, dt_gamma 1/128, march_rays_train -> field -> composite_rays_train (+ the flex composite
    of the palette model) -> PaletteTrainer.train_step's loss with main_palette.py's default weights (palette/utils.py:483-600: colour MSE,
    direct-colour MSE, sparsity / offsets / view-dependence regularisers, palette anchor; the NeRF model: colour MSE) -> backward (composite,
    MLPs, grid_encode backward) -> Adam.  torch_loss: the same loss written with torch on the result dict (as the reference's trainer does)
    instead of palettenerf_amd.train_loss.  Returns (model, step(i) -> None)."""
    import numpy as np
    import torch
    from palettenerf_amd import network, raymarching, renderer, scene
    if model_kind == "palette":
        m = network.PaletteNetwork(renderer.default_opt(test=False), bound=2, cuda_ray=True, min_near=0.02)
    else:
        m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.02)
    scene.seed_field_(m, 0)
    m = m.to(device).train()
    m.density_grid.copy_(torch.from_numpy(scene.slab_density_grid()).to(device))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = 756, 1008          # LLFF fern images_4
    g = torch.Generator().manual_seed(0)
    poses = []
    for i in range(17):       # cameras on a 0.3-radius disc at z = 1.5 looking down -z
        a = 2 * np.pi * i / 17
        p = np.eye(4, dtype=np.float32)
        p[:3, 0], p[:3, 1], p[:3, 2] = [1, 0, 0], [0, -1, 0], [0, 0, -1]
        p[:3, 3] = [0.3 * np.cos(a), 0.3 * np.sin(a), 1.5]
        poses.append(p)
    ro_all, rd_all = scene.get_rays(torch.from_numpy(np.stack(poses)), scene.intrinsics_from_fov(H, W, 0.9), H, W)
    ro_all, rd_all = ro_all.to(device), rd_all.to(device)
    if torch_adam or fp16:
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    else:
        from palettenerf_amd import optim
        opt = optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", enabled=fp16)
    target = torch.rand(rays, 3, device=device)
    inds_all = torch.randint(0, H * W, [64, rays], generator=g).to(device)     # the index draws of 64 steps, resident (the reference draws them on the device)

    from palettenerf_amd.train_loss import train_loss
    lam = dict(lambda_sparsity=2e-4, lambda_offsets=0.03, lambda_view_dep=0.1, lambda_palette=1e-3)     # main_palette.py:83-89 (no smooth loss, no weight guide)
    target = target[None]
    origin = (m.basis_color.detach() + 0.02).clone() if model_kind == "palette" else None    # basis_color_origin: the extracted palette the anchor term pulls towards

    def step(i):
        inds = inds_all[i % 64]
        ro, rd = ro_all[i % 17, inds][None], rd_all[i % 17, inds][None]
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            r = m.run_cuda(ro, rd, dt_gamma=1 / 128, perturb=True, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
            if not torch_loss:
                if model_kind == "palette":
                    loss, _ = train_loss(r, target, basis_color=m.basis_color, basis_color_origin=origin, **lam)
                else:
                    loss, _ = train_loss(r, target)
            else:
                loss = ((r["image"] - target) ** 2).mean(-1)
                if model_kind == "palette":
                    loss = loss + lam["lambda_sparsity"] * r["omega_sparsity"].mean() + lam["lambda_offsets"] * r["offsets_norm"].mean()
                    loss = loss + lam["lambda_view_dep"] * r["view_dep_norm"].mean()
                    loss = loss + lam["lambda_palette"] * ((m.basis_color - origin) ** 2).sum(dim=-1).mean()
                    loss = loss + ((r["direct_rgb"] - target) ** 2).mean()
                loss = loss.mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()

    return m, step


def training_leg(model_kind, device, steps=50, warmup=8, rays=4096, torch_loss=False):
    """Wall ms per training step over `steps` steps, and -- from a torch.profiler trace of 10 further steps -- the device time of a step's
    kernels and the number of launches per step (every kernel of the process is traced, the C-ABI ones included).  torch_loss: the trainer's
    loss written with torch on the result dict, as the reference's trainer has it (what pnr_train_loss_* replaces)."""
    import torch
    m, step = make_training_step(model_kind, rays, device, torch_loss=torch_loss)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    rec = {"wall_ms_per_step": wall, "steps": steps, "rays_per_step": rays, "samples_per_step": int(m.step_counter[(m.local_step - 1) % 16, 0]),
           "what": f"configs[3] shape: -m {model_kind} training step, {rays} rays, slab scene, dt_gamma 1/128, Adam; synthetic targets; "
                   + ("PaletteTrainer.train_step's loss (main_palette.py's default weights) " if model_kind == "palette" else "Trainer.train_step's colour MSE (nerf/utils.py:535) ")
                   + ("written with torch on the result dict" if torch_loss else "through pnr_train_loss_* (one launch each way)")}
    try:
        from torch.profiler import ProfilerActivity, profile
        n = 10
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(n):
                step(warmup + steps + i)
            torch.cuda.synchronize()
        kernels = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
        if kernels:
            rec["kernel_ms_per_step"] = sum(e.device_time for e in kernels) / n / 1e3
            rec["launches_per_step"] = len(kernels) / n
            hip = [e for e in kernels if e.name.startswith("pnr::") or "pnr::" in e.name[:12]]
            rec["launches_hip_per_step"] = len(hip) / n          # this repository's kernels: march, lookups, MLP stacks, composites, binned gradient, Adam
            rec["kernel_ms_hip_per_step"] = sum(e.device_time for e in hip) / n / 1e3
            rec["launches_torch_per_step"] = (len(kernels) - len(hip)) / n   # torch's: ray selection, march_rays_train's bookkeeping (noise, counter, one host read), gradient zero fills, x -> [0, 1] in front of the encoders, gradient accumulation
            rec["wall_over_kernel"] = wall / rec["kernel_ms_per_step"]
    except Exception as e:   # noqa: BLE001 -- a profiler that does not work on this box is reported, the wall figure stands
        rec["profiler_error"] = repr(e)
    return rec


def occupancy_leg(device):
    """one update_extra_state of a 2 x 128^3 grid through csrc/occupancy.hip -- full sweep (iter_density < 16) and partial --
    device time between two events around the call (its random numbers drawn outside) and host time of the call (nothing waits)."""
    import torch
    from palettenerf_amd import network, scene
    m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(device).train()
    G, n = 128, 128 ** 3 // 4
    out = {}
    for name, start in (("full", 0), ("partial", 16)):
        draws = dict(noise=torch.rand(2, G ** 3, 3, device=device)) if start == 0 else dict(
            noise=torch.rand(2, 2 * n, 3, device=device), coords=torch.randint(0, G, (2, n, 3), device=device, dtype=torch.int32),
            occ_rand=torch.randint(0, 2 ** 31 - 1, (2, n), device=device, dtype=torch.int32))
        for _ in range(3):
            m.iter_density = start
            m.update_extra_state(**draws)
        torch.cuda.synchronize()
        dev_ms, host_ms = [], []
        for _ in range(10):
            m.iter_density = start
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            h0 = time.perf_counter()
            m.update_extra_state(**draws)
            host_ms.append((time.perf_counter() - h0) * 1e3)
            e1.record()
            torch.cuda.synchronize()
            dev_ms.append(e0.elapsed_time(e1))
        out[name] = {"device_ms": sorted(dev_ms)[5], "host_ms": sorted(host_ms)[5], "samples": 2 * G ** 3 if start == 0 else 4 * n}
    out["what"] = "update_extra_state, 2 x 128^3 cells, fused HIP sweep (no host wait); median of 10"
    return out


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
NERF_FLOP_PER_SAMPLE = 18688.0      # 9 344 MAC
PALETTE_FLOP_PER_SAMPLE = 36094.0   # 18 047 MAC (nb = 4, no clip head)
CLOCK_GHZ = 2.4             # MI355X_MICROARCH.md: peak engine clock


def mfma_leg(m, model_kind, rows, device, precision, reps=30):
    """north_star: "MFMA utilisation ... against gfx950 peak".  The fused field kernel -- the path's only MFMA stage -- launched alone on `rows`
    rows (the headline's average live rows per launch) of seeded encoder features, HIP events on the launch stream; useful FLOP = the
    reference's MAC count of the layers x 2, issued FLOP = useful x 3 for the split-fp16 form (three matrix products per layer)."""
    import ctypes
    import torch
    from palettenerf_amd import _lib
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = m._fused
    prec = {"fp32": 0, "f16x3": 1, "f16x2": 2}[precision]
    g = torch.Generator().manual_seed(0)
    B = int(rows)
    enc = [((torch.rand(16, B, 2, generator=g) - 0.5) * 0.4).to(device) for _ in range(2)]
    d = torch.randn(B, 3, generator=g)
    d = (d / d.norm(dim=1, keepdim=True)).to(device)
    sig, rgb = torch.empty(B, device=device), torch.empty(B, 3, device=device)
    blob = f._pack(prec)
    if model_kind == "nerf":
        name, flop = "k_nerf_field_fwd (pnr_nerf_field_forward)", NERF_FLOP_PER_SAMPLE
        fn = lambda: lib.pnr_nerf_field_forward(enc[0].data_ptr(), d.data_ptr(), blob.data_ptr(), B, sig.data_ptr(), rgb.data_ptr(), prec, ctypes.c_float(1.0), stream)
    else:
        name, flop = "k_palette_field_fwd (pnr_palette_field_forward)", PALETTE_FLOP_PER_SAMPLE
        aux = torch.empty(B, f.aux_channels, device=device)
        a = _lib.PaletteFieldArgs()
        a.ctl, a.B, a.level_stride = None, B, B
        a.enc, a.enc_palette, a.enc_clip = enc[0].data_ptr(), enc[1].data_ptr(), None
        a.dirs, a.deltas, a.packed = d.data_ptr(), None, blob.data_ptr()
        a.num_basis, a.clip_dim, a.pred_clip = f.nb, f.clip_dim, int(f.pred_clip)
        a.density_scale, a.offsets_weight, a.view_dep_weight, a.aux_stride = 1.0, 1.0, 1.0, f.aux_channels
        a.sigmas, a.rgbs, a.aux, a.precision = sig.data_ptr(), rgb.data_ptr(), aux.data_ptr(), prec
        fn = lambda: lib.pnr_palette_field_forward(ctypes.byref(a), stream)
    for _ in range(3):
        if fn() != 0:
            raise RuntimeError("field kernel launch failed")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    ms = t[len(t) // 2]
    useful = B * flop / (ms * 1e-3) / 1e12
    products = {0: 1, 1: 3, 2: 3}[prec]     # f16x2 rounds the colour layers' activations once (2 products there): counted as 3, an upper bound of what is issued
    rec = {"bound": "mfma", "kernel": name, "rows": B, "avg_launch_ms": ms, "flop_per_sample": flop, "achieved": useful, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": useful / MFMA_PEAK_TFLOPS, "matrix_products_per_layer": products, "issued_frac": useful * products / MFMA_PEAK_TFLOPS if prec else None,
           "precision": precision,
           "note": ("stand-alone launches of the fused field kernel on the headline's average live rows per launch; useful FLOP = 2 x the reference layers' MACs; "
                    + ("issued = useful x 3 split-fp16 products on v_mfma_f32_32x32x16_f16 against the dense fp16 peak" if prec else
                       "exact-fp32 path on v_mfma_f32_32x32x2_f32: priced against the fp16 peak for comparison only (the fp32 matrix peak is 1/16 of it)"))}
    return rec


def gather_ceilings(timeout=90):
    """The random-row gather ceilings of THIS box: profiles/micro/gather_rate (built from gather_rate.hip by __graft_entry__.build) run as a child
    process -- 8-byte rows, every lane a random row, 8 loads in flight per lane, over a 4 MB (L2-resident) and a 50 MB (the whole hash table) range.
    Returns {"l2_resident", "table_50mb"} in lane-loads per clock per CU at the binary's nominal 2.4 GHz, or None when the binary is not there."""
    exe = os.path.join(ROOT, "profiles", "micro", "gather_rate")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=timeout).stdout
    except (OSError, subprocess.TimeoutExpired):
        return None
    got = {}
    for line in out.splitlines():
        if line.startswith("8-byte rows (fp32 row)") and "per clock per CU" in line:
            mb = float(line.split("table")[1].split("MB")[0])
            val = float(line.split("=")[1].split("per clock")[0])
            got[{4.0: "l2_resident", 50.0: "table_50mb"}.get(mb, str(mb))] = val
    return got if "l2_resident" in got and "table_50mb" in got else None


def l2_bound_of(samples_per_launch, n_tables_rows, launch_ms, ceilings=None):
    """The lookup kernel's other bound: divergent lane-requests per clock per CU.  Every (sample, level) issues 8 row gathers; the
    ceilings are profiles/micro/gather_rate.hip's (every lane a random row, 8 loads in flight per lane) over an L2-resident 4 MB table and over
    the whole 50 MB table -- measured in this run when `ceilings` (gather_ceilings()) is given, else round 2's constants 0.43 / 0.106."""
    reqs = samples_per_launch * 16 * 8 * n_tables_rows
    per_clk_cu = reqs / (launch_ms * 1e-3) / (CLOCK_GHZ * 1e9) / 256.0
    c_l2, c_50 = (ceilings["l2_resident"], ceilings["table_50mb"]) if ceilings else (0.43, 0.106)
    return {"lane_requests_per_launch": reqs, "lane_requests_per_clk_per_cu": per_clk_cu, "clock_ghz_assumed": CLOCK_GHZ,
            "random_row_ceiling_l2_resident": c_l2, "random_row_ceiling_50mb_table": c_50,
            "ratio_to_l2_resident_random_ceiling": per_clk_cu / c_l2,
            "ceilings_source": "profiles/micro/gather_rate run in this bench run" if ceilings else "round-2 constants (profiles/micro/gather_rate not built here)",
            "note": "rows gathered per clock per CU; above the random-row ceilings because neighbouring lanes share 128-byte lines (8x8-pixel wave tiles, level-major launch)"}


LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE",
                "PNR_BENCH_FORCE_DIST")


def child_env(**more):
    """Environment of a child pass of this script: never the launcher's rank variables (parse() switches the workload and scaling defaults on WORLD_SIZE, and a
    child must not try to join the parent's communicator)."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    env.update(more)
    return env


L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md, "L2 (per XCD)": 4 MiB x 8, ~34.5 TB/s aggregate


def roofline_block(kernel_name, achieved, traffic, traffic_info, n_launches, k_ms, k_units, per_sample, n_tables):
    """The lookup kernel's roofline object.  `achieved` = algorithmic GB/s (SURVEY 8d bytes per sample x live samples / launch time).  While that stays below the
    HBM peak the bound is HBM (configs[1]: one 50 MB table, 0.7-0.8 of 8 TB/s).  When it does NOT -- the PaletteNeRF lookups read two or three tables through ONE
    interleaved row per corner and each row serves the eight-fold corner re-use of neighbouring samples out of L2 / Infinity Cache: 17 TB/s of algorithmic bytes
    -- HBM is not what bounds the kernel and a fraction of its peak above 1 says nothing.  The bound is then the on-die path: `peak` = the guide's aggregate L2
    bandwidth, `hbm_frac_of_measured_traffic` = the HBM-side bytes the PMC passes measured / launch time / 8 TB/s is what HBM itself sees."""
    avg_ms = k_ms / max(1, n_launches)
    on_die = achieved > HBM_PEAK_GBS
    peak = L2_PEAK_GBS if on_die else HBM_PEAK_GBS
    r = {"bound": "l2" if on_die else "hbm", "kernel": kernel_name, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
         "traffic": traffic, "traffic_source": traffic_info, "launches": n_launches, "avg_launch_ms": avg_ms,
         "avg_live_samples_per_launch": k_units / max(1, n_launches) / n_tables,
         "algorithmic_bytes_per_sample": per_sample * n_tables, "algorithmic_bytes_per_launch": per_sample * k_units / max(1, n_launches),
         "algorithmic_over_hbm_peak": achieved / HBM_PEAK_GBS}
    if traffic and avg_ms > 0:
        r["hbm_frac_of_measured_traffic"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        r["traffic_over_algorithmic"] = traffic / max(1.0, r["algorithmic_bytes_per_launch"])
    if on_die and traffic and avg_ms > 0:
        # which of the two ceilings the launch sits closer to: the L2's (algorithmic bytes) or HBM's (the bytes that actually crossed it)
        hf = r["hbm_frac_of_measured_traffic"]
        r["closer_ceiling"] = {"name": "hbm, by measured traffic" if hf > r["frac"] else "l2, by algorithmic bytes", "frac": max(hf, r["frac"])}
    if on_die:
        r["note"] = ("algorithmic bytes per second exceed the HBM peak: the interleaved table rows are re-used out of L2 / Infinity Cache, so the kernel is priced against the "
                     "aggregate L2 bandwidth (MI355X_MICROARCH.md: ~34.5 TB/s); `traffic` = HBM-side bytes per launch from the PMC passes, `hbm_frac_of_measured_traffic` what "
                     "HBM itself sees, `l2_bound` the row-gather rate against this box's random-row ceilings")
    return r


def measure_traffic(argv_core, kernel_substr="k_frame_grid", timeout=240):
    """roofline.traffic measured in THIS run: two child passes of this script's headline workload under `rocprofv3 --pmc <counter> --kernel-trace`
    (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, PMC slots), 2 + 3 frames each; HBM-side bytes per lookup launch =
    2 x FETCH_SIZE + WRITE_SIZE KiB (the guide's gfx950 correction: FETCH_SIZE reports half the bytes of a coalesced read), mean over the
    launches that did work.  Children are started as separate processes (never exec'd from this one).  Returns (bytes or None, info)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, {"error": "rocprofv3 not found"}
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, {"error": "already running under a profiler"}
    vals, info = {}, {}
    tmp = tempfile.mkdtemp(prefix="pnr_pmc_", dir="/tmp")
    env = child_env(TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   *argv_core, "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--no-traffic"]
            t0 = time.perf_counter()
            try:
                pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, {"error": f"{counter} pass timed out after {timeout} s"}
            info[counter.lower() + "_pass_seconds"] = time.perf_counter() - t0
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if pr.returncode != 0 or not files:
                return None, {"error": f"{counter} pass failed (rc {pr.returncode})", "tail": pr.stdout.decode(errors="replace")[-400:]}
            rows = [r for r in csv.DictReader(open(files[0])) if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not rows:
                return None, {"error": f"no {kernel_substr} dispatches in the {counter} pass"}
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            keep = [i for i, d in enumerate(dur) if d > max(dur) / 4]
            vals[counter] = sum(float(rows[i]["Counter_Value"]) for i in keep) / len(keep)
            info[counter.lower() + "_kib_per_launch"] = vals[counter]
            info["launches_counted"] = len(keep)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    info["formula"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read's bytes)"
    return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, info


def dropin_leg(args, device, bank, model_kind, steps, fuse_field=False):
    """The operator-API path under the reference's own loop (north_star: "nerf/renderer.py and palette/renderer.py drop in unchanged"): this
    repository's mirror of run_cuda in its `compat` mode issues exactly what an unchanged run_cuda issues -- march_rays / GridEncoder /
    SHEncoder / composite_rays[_flex] through the per-op HIP kernels behind the reference's Python signatures, torch nn.Linear MLPs, the
    boolean-mask compaction with its host sync (nerf/renderer.py:354-380, palette/renderer.py:430-550).  No fused field, no device loop."""
    import torch
    from palettenerf_amd import _torch_glue
    cargs = argparse.Namespace(**vars(args))
    cargs.mode, cargs.fp16, cargs.half_tables = "compat", False, False
    mm = build_model(cargs, device, model_kind)
    if fuse_field:      # palettenerf_amd.dropin.fuse_field: the model's forward() = lookup op + ONE fused MFMA field launch; renderer loop and operator calls unchanged
        from palettenerf_amd import dropin
        dropin.fuse_field(mm, args.field_precision)
    kk = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    if model_kind == "palette":
        kk["gui_mode"] = False
    timed_frames(mm, bank, kk, 2, False)
    ms, rend = timed_frames(mm, bank, kk, steps, False, first_step=args.warmup)
    rec = {"ms_per_step": ms, "value": rend / (ms * 1e-3), "unit": "samples/s", "rendered_per_step": rend, "steps": steps, "march_mode": "compat",
           "what": f"-m {model_kind} inference, {args.wl['H']}x{args.wl['W']}: per-op HIP kernels behind the reference's operator API + "
                   + ("dropin.fuse_field(model): forward() = lookup op + one fused MFMA field launch" if fuse_field else "torch nn.Linear MLPs")
                   + ", the reference's loop (host-side boolean-mask compaction, one sync per iteration)"}
    # the stand-alone lookup op inside this loop: HIP events around every pnr_grid_encode_forward* call of one frame (rows include dead / padded slots)
    names = ["pnr_grid_encode_forward", "pnr_grid_encode_forward_layout"]
    prof = _torch_glue.profile_kernels(names)
    ro, rd = bank.get(args.warmup)
    with torch.no_grad():
        mm.render(ro, rd, **kk)
    torch.cuda.synchronize()
    _torch_glue.profile_kernels(None)
    ev = [e for n in names for e in prof[n]]
    if ev:
        k_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ev)
        rows = sum(u for _, _, u in ev)
        rec["lookup_op"] = {"kernel": "k_grid_fwd_d3c2 (pnr_grid_encode_forward)", "launches": len(ev), "avg_launch_ms": k_ms / len(ev), "rows_per_launch": rows / len(ev),
                            "achieved_gbs_evaluated_rows": rows * GRID_BYTES_PER_SAMPLE_FP32 / (k_ms * 1e-3) / 1e9,
                            "frac_evaluated_rows": rows * GRID_BYTES_PER_SAMPLE_FP32 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "rows = every row the loop hands the encoder (dead slots and the 128-row alignment padding included: their gathers all hit one cell); events around the call"}
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as tp:
            with torch.no_grad():
                mm.render(ro, rd, **kk)
            torch.cuda.synchronize()
        kernels = [e for e in tp.events() if e.device_type == torch.autograd.DeviceType.CUDA and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
        hip = [e for e in kernels if "pnr::" in e.name[:16]]
        rec.update(launches_per_frame=len(kernels), launches_hip_per_frame=len(hip), kernel_ms_per_frame=sum(e.device_time for e in kernels) / 1e3,
                   kernel_ms_hip_per_frame=sum(e.device_time for e in hip) / 1e3)
    except Exception as e:   # noqa: BLE001 -- reported, the wall figure stands
        rec["profiler_error"] = repr(e)
    del mm
    return rec


def strong_leg(args, m, kw, device, world, rank, steps, n_views=1):
    """The OTHER scaling question next to the headline's.  n_views = 1 (next to a weak-scaling headline): ONE frame's rays split over the ranks
    (32 x 32 tiles round-robin), every rank renders its share, one all-gather assembles the frame on every rank -- north_star's N > 1 question.
    n_views = world (next to the strong-scaling headline, `extra.weak`): a step renders `world` views stacked into one image, per-GPU work fixed.
    Returns ms per step (max over ranks, gathers pipelined as in the headline loop), the same with a blocking gather per step, and the all-gather
    alone."""
    import torch
    import torch.distributed as dist
    from palettenerf_amd import dist as pdist
    from palettenerf_amd.fused import tile_ray_order
    H, W = n_views * args.wl["H"], args.wl["W"]      # the views stacked vertically (as in main)
    nb = int(getattr(m, "num_basis", 0))
    K = 5 if args.model == "nerf" else 8 + 4 * nb
    saved = getattr(m._fused, "ray_order", None)

    def agree(ok):
        """All ranks leave the leg together or none does: a rank that failed locally must not walk on to later collectives while the others
        still sit in this leg's (the job would hang until the RCCL timeout).  One all_reduce of an error flag on the control path."""
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item()) == 0

    err = None
    try:   # set-up: allocations that can fail on one rank only
        idx, _ = pdist.shard_indices(H, W, rank, world)
        bank = RayBank(args, n_views, idx, device)
        m._fused.ray_order = tile_ray_order(idx, W, 8).to(device)
        g = pdist.FrameGatherer(H, W, K, device)
        for i in range(min(steps + 3, bank.n_steps)):
            bank.get(i)
    except RuntimeError as e:
        err = e
    if not agree(err is None):
        m._fused.ray_order = saved
        raise RuntimeError(f"strong leg skipped on every rank: set-up failed on {'this' if err is not None else 'another'} rank" + (f" ({err})" if err is not None else ""))

    def run(n, first, blocking):
        pending, rendered = None, 0
        for i in range(n):
            ro, rd = bank.get(first + i)
            with torch.no_grad():
                r = m.render(ro, rd, **kw)
            rendered += int(r["rendered"].sum())
            h = g.start(gather_parts(args, r, nb))
            if blocking:
                g.finish(h)
            else:
                if pending is not None:
                    g.finish(pending)
                pending = h
        if pending is not None:
            g.finish(pending)
        return rendered

    out = {}
    try:
        # past this point every rank is inside the same sequence of collectives: a failure here is re-raised by main() (the launcher tears the
        # job down) instead of being swallowed on one rank
        run(3, 0, False)
        for name, blocking in (("pipelined", False), ("blocking_gather", True)):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            rendered = run(steps, 3, blocking)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0, float(rendered)], dtype=torch.float64, device=device)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out[name] = {"ms_per_frame": float(tmax[0]) / steps * 1e3, "value": float(t[1]) / float(tmax[0]), "unit": "samples/s"}
        parts = [torch.zeros(idx.numel(), K, device=device)]
        for _ in range(3):
            g(parts)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            g(parts)
        torch.cuda.synchronize()
        out["all_gather_ms"] = (time.perf_counter() - t0) / 20 * 1e3
        what = (f"ONE {H}x{W} frame split over {world} ranks in 32x32 tiles + one all_gather_into_tensor per frame (strong scaling)" if n_views == 1 else
                f"{n_views} views of {args.wl['H']}x{W} per step, their 32x32 tiles dealt over {world} ranks + one all_gather_into_tensor per step (weak scaling: per-GPU work fixed)")
        out.update(steps=steps, rays_per_rank=int(idx.numel()), gathered_floats_per_ray=K, bytes_per_rank=int(idx.numel()) * K * 4, what=what, workload=args.wl["label"])
    finally:
        m._fused.ray_order = saved
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    world, rank, local_rank = world_from_env(args, argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("PNR_BENCH_FORCE_DIST") == "1"  # the latter exercises the RCCL path on a 1-GPU box
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        # (30 min: rank 0 measures roofline.traffic in two child passes while the others wait at a barrier)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=30))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from palettenerf_amd import _torch_glue, dist as pdist, scene
    from palettenerf_amd.fused import tile_ray_order
    wl = args.wl
    H, W = wl["H"], wl["W"]
    m = build_model(args, device)
    nb = int(getattr(m, "num_basis", 0))
    n_views = world if args.scaling == "weak" else 1
    VH = n_views * H   # the views stacked vertically: one (n_views * H) x W image
    if args.shard is not None and not use_dist:      # one GPU standing in for rank R of an N-way split (no communicator): the PMC child passes of an N > 1 run
        idx, n_max = pdist.shard_indices(VH, W, args.shard[0], args.shard[1])
    else:
        idx, n_max = pdist.shard_indices(VH, W, rank, world)
    bank = RayBank(args, n_views, idx, device)
    kw = dict(perturb=False, dt_gamma=wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    if args.model == "palette":
        kw["gui_mode"] = False
    native = m.march_mode == "native"
    if native and args.ray_order != "rowmajor":
        m._fused.ray_order = tile_ray_order(idx, W, {"tile8": 8, "tile4": 4, "tile16": 16, "morton": 0}[args.ray_order]).to(device)   # idx: row-major pixel ids (of the stacked views) this rank renders

    K = 5 if args.model == "nerf" else 8 + 4 * nb
    gatherer = pdist.FrameGatherer(VH, W, K, device) if use_dist else None
    pending = []   # all-gather of the previous frame, still in flight

    render_ev = []   # N > 1: HIP events around this rank's render of every timed step (the shard alone: no packing, no gather)

    def frame(i, timed=False):
        ro, rd = bank.get(i)
        if timed and use_dist:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=args.fp16):
            r = m.render(ro, rd, **kw)
        if timed and use_dist:
            ev[1].record()
            render_ev.append(ev)
        if use_dist:  # one all-gather of the packed rows; every rank ends up with the full frame.  It is started here and completed
            # after the NEXT frame has been rendered (or at the end of the timed region): communication overlaps compute
            handle = gatherer.start(gather_parts(args, r, nb))
            full = gatherer.finish(pending.pop()) if pending else None
            pending.append(handle)
            return r, full
        return r, None

    n_frames = args.warmup + args.steps
    for i in range(min(n_frames, bank.n_steps)):   # rays of every pose of the run: resident before the timed region
        bank.get(i)
    for i in range(args.warmup):
        frame(i)
    if native:   # untimed, on every rank (a frame holds a collective): one frame with the in-library HIP-event timing on, so that the events exist
        m._fused.time_grid_kernel = True
        frame(args.warmup)
        m._fused.time_grid_kernel = False
    if use_dist:
        if pending:
            gatherer.finish(pending.pop())
        dist.barrier()
    torch.cuda.synchronize()
    GRID_OPS = ["pnr_grid_encode_forward", "pnr_grid_encode_forward_layout"]
    prof = _torch_glue.profile_kernels(GRID_OPS) if rank == 0 else None
    rendered = torch.zeros(1, dtype=torch.int64, device=device)
    rendered_host, rows, looks, iterations = 0, 0, 0, 0
    native_ms, native_launches, native_live = 0.0, 0, 0
    timed_native = native and rank == 0
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # diagnostics only (no sync inside the loop)
    import gc
    gc.collect()
    gc.disable()   # as timeit does: a generation-2 collection of the interpreter's heap (tens of ms with torch imported) is not part of a frame
    F_main = max(1, int(args.main_frames_in_flight))
    if F_main > 1:
        # The timed steps with F frames in flight (pipeline.FramesInFlight): frame i on handle i % F, each on its own host thread and stream; with
        # N > 1 ranks the all-gather of frame i is issued as the i-th collective on every rank (dist.OrderedGather), each thread finishing its
        # previous gather after it has started the next one.
        import threading
        from palettenerf_amd.pipeline import FramesInFlight
        fif = FramesInFlight(m, F_main, device, shared_stream=args.shared_stream)
        if use_dist:
            gatherer = pdist.FrameGatherer(VH, W, K, device, slots=F_main + 1)
        acc = {"rendered": 0, "rows": 0, "looks": 0, "iterations": 0}
        lock = threading.Lock()
        first = {}

        def make_run(n, offset, timed):
            og = pdist.OrderedGather(gatherer) if use_dist else None
            last = {}

            def before(i, model):
                model._fused.time_grid_kernel = bool(timed and timed_native and i == 0)

            def consume(i, r):
                if og is not None:
                    h = og.submit(i, gather_parts(args, r, nb))
                    prev = last.get(i % F_main)
                    if prev is not None:
                        og.finish(prev)
                    last[i % F_main] = h
                if timed:
                    with lock:
                        acc["rendered"] += int(r["rendered"].sum())
                        acc["rows"] += r["n_samples"]
                        acc["looks"] += int(r.get("host_looks", 0))
                        acc["iterations"] += int(r.get("iterations", 0))
                        if i == 0:
                            first.update(grid_ms=r.get("grid_ms", 0.0), grid_launches=r.get("grid_launches", 0), rendered=int(r["rendered"].sum()))
                return None

            fif.render(lambda i: bank.get(offset + i), n, consume=consume, before=before if native else None, abort=og.abort if og is not None else None, **kw)
            for h in last.values():      # the last gather of every thread completes inside the (timed) region
                og.finish(h)

        make_run(2 * F_main, 0, False)   # every handle: workspace, packed weights, iteration prediction
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        gc.collect()
        t0 = time.perf_counter()
        step_ev[0].record()
        make_run(args.steps, args.warmup, True)
        for i in range(args.steps):
            step_ev[i + 1].record()     # per-step diagnostics do not exist in this mode
        rendered_host, rows, looks, iterations = acc["rendered"], acc["rows"], acc["looks"], acc["iterations"]
        native_ms, native_launches, native_live = first.get("grid_ms", 0.0), first.get("grid_launches", 0), first.get("rendered", 0)
    else:
        t0 = time.perf_counter()
        step_ev[0].record()
    # Round 6: the timed frames go through render_prepare / render_launch / render_wait / render_result -- one frame on the device at a time, the host's work for
    # frame i + 1 (outputs, argument struct, the next pose's rays) done under frame i's kernels and frame i + 1 enqueued before frame i's result dict is built (and,
    # with N > 1 ranks, before its rows are packed for the all-gather, which therefore overlaps frame i + 1's render as before).  The same loop at every N, so that a
    # scaling curve compares like with like; 0.1-0.15 ms of host turnaround per frame otherwise (3 % of the 800 x 800 frame, 7 % of an eighth of the garden frame).
    # --one-call-per-frame: m.render() per frame, as until round 5 (`extra.one_call_per_frame` holds that figure next to the headline).
    queued = native and F_main == 1 and not args.one_call_per_frame
    if queued:
        q_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        def prepare(i):
            with torch.autocast("cuda", dtype=torch.float16, enabled=args.fp16):
                return m.render_prepare(*bank.get(args.warmup + i), **kw)

        m._fused.time_grid_kernel = bool(timed_native)
        cur = prepare(0)
        q_ev[0][0].record()
        cur = m.render_launch(cur)
        for i in range(args.steps):
            nxt = None
            if i + 1 < args.steps:
                m._fused.time_grid_kernel = False
                nxt = prepare(i + 1)
            done = cur
            ok = m.render_wait(done)
            q_ev[i][1].record()
            step_ev[i + 1].record()          # (behind frame i, in front of frame i + 1: a step's diagnostic time is its frame + the turnaround in front of it)
            if not ok:                       # (never on this script's fixed weights: the frame is rendered again before anything else is enqueued)
                r = m.render_result(done)
            if nxt is not None:
                q_ev[i + 1][0].record()
                cur = m.render_launch(nxt)   # frame i + 1 is on the device before frame i's result dict is built and its rows are packed
            if ok:
                r = m.render_result(done)
            if use_dist:
                handle = gatherer.start(gather_parts(args, r, nb))
                if pending:
                    _full = gatherer.finish(pending.pop())
                pending.append(handle)
            rendered_host += int(r["rendered"])
            rows += r["n_samples"]
            looks += int(r.get("host_looks", 0))
            iterations += int(r.get("iterations", 0))
            if timed_native and i == 0:
                native_ms, native_launches, native_live = r.get("grid_ms", 0.0), r.get("grid_launches", 0), int(r["rendered"])
        if use_dist:
            render_ev.extend(q_ev)
    for i in range(args.steps if (F_main == 1 and not queued) else 0):
        # HIP events around every grid-encode launch cost ~6 us each (two per iteration): instrument the launches of the
        # FIRST timed step only, so the measurement lives inside the timed region without distorting it
        if timed_native:
            m._fused.time_grid_kernel = i == 0
        r, _full = frame(args.warmup + i, timed=True)
        if r["rendered"].is_cuda:
            rendered += r["rendered"]
        else:
            rendered_host += int(r["rendered"])   # native loop: the count is already on the host (it came back with the control block)
        rows += r["n_samples"]
        looks += int(r.get("host_looks", 0))
        iterations += int(r.get("iterations", 0))
        if timed_native and i == 0:
            native_ms, native_launches, native_live = r.get("grid_ms", 0.0), r.get("grid_launches", 0), int(r["rendered"])
        step_ev[i + 1].record()
    if pending:
        _full = gatherer.finish(pending.pop())   # the last frame's all-gather completes inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    rendered += rendered_host
    _torch_glue.profile_kernels(None)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rendered, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rendered = int(rendered.item())

    dist_info = None
    if use_dist and F_main == 1:
        # what the judge of a strong-scaling line needs next to `value`: every rank's own render time per step (HIP events around m.render: the
        # shard alone), the all-gather on its own by HIP events on the stream that waits for it (pack + collective + wait, 20 repeats on the timed
        # loop's buffers), and -- strong scaling -- the SAME frame rendered whole by one GPU inside this job (rank 0 alone, the others wait),
        # so that the speed-up does not depend on a separate N = 1 run of another workload.  Every rank takes part in the collectives below.
        mine = torch.tensor([sum(a.elapsed_time(b) for a, b in render_ev) / max(1, len(render_ev))], dtype=torch.float64, device=device)
        per_rank = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(per_rank, mine)
        parts0 = [torch.zeros(idx.numel(), K, device=device)]
        for _ in range(3):
            gatherer(parts0)
        torch.cuda.synchronize()
        dist.barrier()
        gev = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gatherer(parts0)
            e1.record()
            gev.append((e0, e1))
        torch.cuda.synchronize()
        ag = sorted(a.elapsed_time(b) for a, b in gev)
        ag_t = torch.tensor([ag[len(ag) // 2]], dtype=torch.float64, device=device)
        dist.all_reduce(ag_t, op=dist.ReduceOp.MAX)
        dist_info = {"shard_render_ms_per_rank": [float(v.item()) for v in per_rank], "all_gather_ms": float(ag_t.item()),
                     "all_gather_bytes_per_rank": int(gatherer.n_max) * K * 4,
                     "all_gather_how": "median of 20, HIP events on the render stream around pack + all_gather_into_tensor + wait + un-tile, max over ranks; in the timed "
                                       "loop the collective of frame i overlaps the render of frame i + 1"}
        if args.scaling == "strong" and native and not args.fp16:
            one = torch.zeros(2, dtype=torch.float64, device=device)
            if rank == 0:
                saved_order = m._fused.ray_order
                try:
                    idx_all, _ = pdist.shard_indices(VH, W, 0, 1)
                    bank1 = RayBank(args, 1, idx_all, device)
                    m._fused.ray_order = tile_ray_order(idx_all, W, 8).to(device)
                    n1 = max(3, min(args.steps, 10))
                    for i in range(n1 + 2):
                        bank1.get(args.warmup + i)
                    timed_frames(m, bank1, kw, 2, args.fp16, first_step=args.warmup)
                    ms1, rend1 = timed_frames(m, bank1, kw, n1, args.fp16, first_step=args.warmup)
                    one[0], one[1] = ms1, rend1
                    del bank1
                except Exception as e:      # noqa: BLE001 -- reported; the headline stands, and rank 0 MUST reach the broadcast the other ranks wait in
                    dist_info["single_gpu_error"] = repr(e)
                finally:
                    m._fused.ray_order = saved_order
            dist.broadcast(one, src=0)         # (also the barrier that keeps the other ranks behind rank 0's solo frames)
            if float(one[0]) > 0:
                dist_info["single_gpu_ms_per_step"] = float(one[0])
                dist_info["single_gpu_samples_per_step"] = int(one[1])
                dist_info["strong_speedup_vs_single_gpu_in_this_job"] = float(one[0]) / (elapsed / args.steps * 1e3)

    out = None
    if rank == 0:
        half_rows = args.half_tables or (args.fp16 and native)
        per_sample = GRID_BYTES_PER_SAMPLE_FP16 if args.fp16 else GRID_BYTES_PER_SAMPLE_FP32
        if half_rows:
            per_sample = 12 + 16 * 8 * 2 * 2 + 32 * 4   # half rows gathered, fp32 encoder output written
        n_tables = 1 if args.model == "nerf" else (3 if args.pred_clip else 2)  # palette: encoder + encoder_palette (+ encoder_clip with --pred-clip)
        launches = [e for n in GRID_OPS for e in prof[n]]
        k_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in launches)
        k_units = sum(u for _, _, u in launches)
        n_launches = len(launches)
        kernel_name = "k_grid_fwd_d3c2 (pnr_grid_encode_forward)"
        if native:  # events recorded inside pnr_*_render_frame around every grid launch of the first timed step; LIVE samples (delta > 0), dead slots are skipped by the kernel
            k_ms, k_units, n_launches = native_ms, native_live * n_tables, native_launches
            kernel_name = "k_frame_grid (device-driven frame loop)" if args.model == "nerf" else "k_frame_grid_pair/_triple (device-driven frame loop, tables interleaved)"
            if half_rows:
                kernel_name = ("k_frame_grid_h1" if args.model == "nerf" else "k_frame_grid_h2") + " (device-driven frame loop, fp16 table rows with the reference's half accumulator; fp32 features out)"
        achieved = (k_units * per_sample) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM-side bytes per lookup launch, MEASURED IN THIS RUN: two child passes of the same workload under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE)
        traffic, traffic_info = None, None
        if native and not args.no_traffic and F_main == 1:
            # N = 1: two child passes of this very command.  N > 1 (or the one-rank communicator of the tests): the children are SINGLE-GPU passes on this rank's
            # device that render rank 0's shard of the same split (--emulate-shard 0/N: the same rays, the same shard-sized launches; no communicator, the
            # launcher's environment stripped) -- nothing about a PMC pass needs the collective.  The other ranks wait at the barrier below.
            targv = core_argv(args)
            if use_dist:
                targv = [a for a in targv] + (["--emulate-shard", f"0/{world}"] if world > 1 else [])
            traffic, traffic_info = measure_traffic(targv, timeout=240 if not use_dist else 180)
        field_note = {"f16x3": "field: f16x3 split products, fp32 accumulate", "f16x2": "field: f16x3 for sigma_net, colour layers with activations rounded once to f16", "fp32": "field: exact fp32 MFMA"}[args.field_precision]
        dtype_label = ("f16 tables + autocast" if args.fp16 else "f32") + (f" ({field_note})" if getattr(m, "fused_field", False) else "")
        raw_steps = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
        slowest = max(range(args.steps), key=lambda i: raw_steps[i])
        per_step = sorted(raw_steps)
        out = {
            "metric": "rendered_samples_per_sec", "value": total_rendered / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1], "slowest_step": slowest}, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": f"{wl['label']}, {H}x{W}, {n_views} view(s)/step, camera moving ({'static pose' if args.static_pose else 'one pose of the path per step'})",
                       "rays_per_step": n_views * H * W, "rendered_samples_per_step": total_rendered // args.steps,
                       "evaluated_rows_per_step_rank0": rows // args.steps, "density_scale": args.density_scale, "dt_gamma": wl["dt_gamma"], "march_mode": m.march_mode,
                       "fused_field": bool(getattr(m, "fused_field", False)), "field_precision": args.field_precision, "half_tables": bool(half_rows), "ray_order": args.ray_order,
                       "iterations_per_frame_rank0": iterations / max(1, args.steps), "host_looks_per_frame": looks / max(1, args.steps),
                       "frames_in_flight": F_main, "host_prepares_next_frame_under_this_one": bool(queued),
                       "gathered_floats_per_ray": K if use_dist else 0, "rccl_ranks": dist.get_world_size() if use_dist else 1,
                       "parallelism": f"32x32 ray tiles of {n_views} view(s) round-robin over {world} GPUs + one all_gather/step" if world > 1 else "single GPU",
                       **({"defaulted_for_ranks": True} if args.defaulted_for_ranks else {}), **(dist_info or {})},
            "roofline": roofline_block(kernel_name, achieved, traffic, traffic_info, n_launches, k_ms, k_units, per_sample, n_tables if native else 1),
        }
        if dist_info:      # what a strong-scaling curve should be read from, at the top level (the N = 1 run of the driver is configs[1], another workload)
            for k in ("strong_speedup_vs_single_gpu_in_this_job", "single_gpu_ms_per_step", "shard_render_ms_per_rank", "all_gather_ms"):
                if k in dist_info:
                    out[k] = dist_info[k]
        if native and k_ms > 0 and n_launches > 0:
            out["roofline"]["l2_bound"] = l2_bound_of(k_units / n_launches / n_tables, 1, k_ms / n_launches,
                                                      ceilings=None if args.no_extras else gather_ceilings())
            try:    # north_star: MFMA utilisation of the fused field kernel against the gfx950 peak
                out["roofline"]["mfma"] = mfma_leg(m, args.model, max(1024, int(k_units / n_launches / n_tables)), device, args.field_precision)
            except RuntimeError as e:
                out["roofline"]["mfma_error"] = str(e)
        if native and not os.environ.get("PNR_NO_HOSTED_TAIL"):
            # The timed lookup launches also finish the march's stragglers (frame.hip: hosted_march_tail): their duration is the lookup's plus what
            # the hosted waves cost it.  The same frame once more with the tail switched off gives the lookup kernel on its own.
            from palettenerf_amd import _lib as _plib
            _l = _plib.load()
            if not use_dist and F_main == 1 and _l.pnr_set_option(b"hosted_tail", 0) == 0:
                try:
                    m._fused.time_grid_kernel = True
                    with torch.no_grad():
                        ra, _ = frame(args.warmup)
                    torch.cuda.synchronize()
                    a_ms, a_n, a_live = ra.get("grid_ms", 0.0), ra.get("grid_launches", 0), int(ra["rendered"])
                    if a_ms > 0:
                        a_ach = a_live * n_tables * per_sample / (a_ms * 1e-3) / 1e9
                        out["roofline"]["lookup_alone"] = {"achieved": a_ach, "frac": a_ach / HBM_PEAK_GBS, "avg_launch_ms": a_ms / max(1, a_n), "launches": a_n,
                                                           "note": "same frame, hosted march tail off (pnr_set_option hosted_tail 0): the lookup launches without the stragglers' march inside them"}
                finally:
                    m._fused.time_grid_kernel = False
                    _l.pnr_set_option(b"hosted_tail", 1)
            out["roofline"]["note_events"] = ("avg_launch_ms: HIP events carried by the lookup launches of the FIRST timed step (hipExtLaunchKernelGGL start / stop); an instrumented launch "
                                              "includes its completion signal and release fence and reads ~6 us longer than rocprofv3's dispatch time stamps for the same launch "
                                              "(profiles/frame_launch_avgs.py on a kernel trace of this command): `achieved` / `frac` are the conservative figures")
            out["roofline"]["note_hosted_tail"] = ("the timed launches host the march tail: their first workgroups march the rays the march launch handed over and look those rows up "
                                                   "(algorithmic bytes count the lookup only; `lookup_alone` is the kernel without that work)")
        if F_main > 1:
            out["roofline"]["note_frames_in_flight"] = (f"{F_main} frames in flight: the timed launches share the chip with another frame's kernels, and "
                                                        "ms_per_step is elapsed / steps, not the latency of a frame")
    if use_dist:
        dist.barrier()
    extra = {}
    if use_dist and native and args.scaling == "strong" and not args.no_extras and F_main == 1:
        # the weak-scaling question (rounds 1-4's N > 1 headline) next to the strong-scaling one: `world` views of the configs[1] lego frame per step,
        # per-GPU work fixed; every rank takes part, rank 0 reports
        try:
            wargs = parse(["--workload", "lego", "--scaling", "weak", "--res", str(args.res), "--density-scale", repr(float(args.density_scale)),
                           "--field-precision", args.field_precision, "--no-cpu-baseline", "--no-extras"])
            wm = build_model(wargs, device)
            wkw = dict(perturb=False, dt_gamma=wargs.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
            weak = strong_leg(wargs, wm, wkw, device, world, rank, max(5, args.extra_steps), n_views=world)
            del wm
            if rank == 0:
                extra["weak"] = weak
        except RuntimeError as e:
            if "skipped on every rank" not in str(e):
                raise
            if rank == 0:
                extra["weak_error"] = str(e)
    if use_dist and native and args.scaling == "weak" and not args.no_extras and F_main == 1:   # (PNR_BENCH_FORCE_DIST=1 runs it over a one-rank communicator)
        # every rank takes part; rank 0 reports.  (--scaling strong makes this split the headline itself.)
        try:
            strong = strong_leg(args, m, kw, device, world, rank, max(5, args.extra_steps))
            if rank == 0:
                extra["strong"] = strong
        except RuntimeError as e:
            if "skipped on every rank" not in str(e):
                raise      # a failure in the middle of the leg's collectives: the other ranks are still inside them -- tear the job down
            if rank == 0:
                extra["strong_error"] = str(e)
    crop_ref = None   # (rays_o, rays_d, oracle results) of the parity crop, for the extra legs
    if rank == 0 and world == 1:
        # --- PSNR / max-abs against the oracle + the CPU baseline, on a centre crop of pose 0 (same device-generated rays for both sides)
        if not args.no_cpu_baseline:
            from palettenerf_amd import rays as prays
            cidx = crop_indices(H, W, args.cpu_crop).to(device)
            pose0 = torch.from_numpy(pose_of(args, 0)[None]).to(device)
            cro, crd = prays.rays_from_indices(pose0, intrinsics_of(args), H, W, cidx[None])
            saved_order = getattr(m._fused, "ray_order", None) if native else None
            if native:
                m._fused.ray_order = None
            with torch.no_grad():
                g = m.render(cro, crd, **kw)
            if native:
                m._fused.ray_order = saved_order
            rec, ref = cpu_baseline(args, (cro.cpu(), crd.cpu()))
            crop_ref = (cro, crd, ref)
            out["cpu_baseline"] = rec
            gi, ri = g["image"].cpu(), ref["image"]
            out["parity"] = {"psnr_vs_oracle_db": scene.psnr(gi, ri), "max_abs_rgb": float((gi - ri).abs().max()),
                             "max_abs_alpha": float((g["weights_sum"].cpu() - ref["weights_sum"]).abs().max()),
                             "rendered_samples_gpu": int(g["rendered"].sum()), "rendered_samples_oracle": int(ref["rendered"].sum()),
                             "sample": rec["sample"].split(" (")[0], "tolerance": "1e-4 abs (north_star)"}
        # --- extra driver-observed legs on the same box: the exact-fp32 field and the PaletteNeRF model (configs[2]) on the same camera path
        if not args.no_extras and native and F_main == 1:
            # the headline's frames as m.render() calls, one call per frame (rounds 1-5's timed loop; what the extra legs below are timed with as well)
            try:
                n1 = max(5, args.extra_steps)
                timed_frames(m, bank, kw, 2, args.fp16, first_step=args.warmup)
                ms1, rend1 = timed_frames(m, bank, kw, n1, args.fp16, first_step=args.warmup)
                extra["one_call_per_frame"] = {"ms_per_step": ms1, "value": rend1 / (ms1 * 1e-3), "unit": "samples/s", "steps": n1,
                                               "what": "the same frames as m.render() calls: the host prepares a frame only after the previous one has returned"}
            except RuntimeError as e:
                extra["one_call_per_frame_error"] = str(e)
        if not args.no_extras and native and args.workload == "lego" and not args.fp16:
            n = max(1, args.extra_steps)
            for name, kind, prec in (("fp32_field", "nerf", "fp32"), ("f16x2_field", "nerf", "f16x2"), ("palette", "palette", "f16x3"), ("palette_f16x2_field", "palette", "f16x2"),
                                     ("palette_fp32_field", "palette", "fp32")):
                try:
                    mm = build_model(args, device, kind, prec)
                    mm._fused.ray_order = m._fused.ray_order
                    kk = dict(kw)
                    if kind == "palette":
                        kk["gui_mode"] = False
                    timed_frames(mm, bank, kk, 3, False)
                    ms, rend = timed_frames(mm, bank, kk, n, False, first_step=args.warmup)
                    extra[f"{name}_ms_per_step"] = ms
                    extra[f"{name}_rendered_per_step"] = rend
                    if name == "f16x2_field" and crop_ref is not None:   # the rounded-activation form against the same oracle crop as the headline's parity block
                        saved = mm._fused.ray_order
                        mm._fused.ray_order = None
                        with torch.no_grad():
                            g2 = mm.render(crop_ref[0], crop_ref[1], **kk)
                        mm._fused.ray_order = saved
                        extra["f16x2_field_parity"] = {"psnr_vs_oracle_db": scene.psnr(g2["image"].cpu(), crop_ref[2]["image"]),
                                                       "max_abs_rgb": float((g2["image"].cpu() - crop_ref[2]["image"]).abs().max()),
                                                       "rendered_samples_gpu": int(g2["rendered"].sum()), "tolerance": "1e-4 abs (north_star)"}
                    if name == "palette":   # the same frames with a RegionEdit active (palette/renderer.py:121-147): it runs inside the field kernel's epilogue
                        from palettenerf_amd import renderer as prenderer
                        mm.edit = prenderer.RegionEdit(mm.opt)
                        mm.edit.update_cent(mean_xyz=torch.tensor([0.2, 0.1, -0.1], device=device))
                        mm.edit.update_std(std_xyz=0.3)
                        mm.edit.update_delta_hsv(mm.basis_color.data.clamp(0, 1), (mm.basis_color.data * 0.5 + 0.3).flip(0).clamp(0, 1))
                        timed_frames(mm, bank, kk, 3, False)
                        extra["palette_region_edit_ms_per_step"], _ = timed_frames(mm, bank, kk, n, False, first_step=args.warmup)
                    del mm
                except RuntimeError as e:   # reported, never hidden
                    extra[f"{name}_error"] = str(e)
            extra["extra_steps"] = n
        # --- the same camera path with several frames in flight (palettenerf_amd/pipeline.py: one host thread, fused-field object and stream per
        #     frame in flight, the same weights).  Throughput of a video render; the headline above stays one frame at a time (its ms/frame is a latency).
        if not args.no_extras and native and not args.fp16 and args.frames_in_flight:
            import gc
            from palettenerf_amd.pipeline import FramesInFlight
            legs = {}
            for F in args.frames_in_flight:
                n = -(-max(6, int(1.5 * args.extra_steps)) // F) * F   # every handle renders the same number of frames
                try:
                    fif = FramesInFlight(m, F, device)
                    t_sub, t_done = {}, {}

                    def rays_of(i, t_sub=t_sub):
                        t_sub[i] = time.perf_counter()
                        return bank.get(args.warmup + i)

                    def consume(i, r, t_done=t_done):
                        t_done[i] = time.perf_counter()
                        return int(r["rendered"].sum())
                    fif.render(lambda i: bank.get(i), 2 * F, consume=lambda i, r: 0, **kw)   # every handle: workspace, packed weights, iteration prediction
                    torch.cuda.synchronize()
                    gc.collect()
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    try:
                        t0 = time.perf_counter()
                        counts = fif.render(rays_of, n, consume=consume, **kw)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                    finally:
                        if gc_was_on:
                            gc.enable()
                    lat = sorted(t_done[i] - t_sub[i] for i in range(n))
                    legs[str(F)] = {"value": sum(counts) / dt, "unit": "samples/s", "ms_per_step": dt / n * 1e3, "steps": n,
                                    "frame_latency_ms_median": lat[n // 2] * 1e3}
                    del fif
                except RuntimeError as e:   # reported, never hidden
                    legs[str(F)] = {"error": str(e)}
            extra["frames_in_flight"] = legs
        if args.shard_emulation > 1 and native:
            S = args.shard_emulation
            times, samples = [], []
            for s in range(S):
                sidx, _ = pdist.shard_indices(H, W, s, S)
                sargs = argparse.Namespace(**vars(args))
                sargs.static_pose = True
                sbank = RayBank(sargs, 1, sidx, device)
                m._fused.ray_order = tile_ray_order(sidx, W, 8).to(device)
                timed_frames(m, sbank, kw, 2, args.fp16)
                ms, rend = timed_frames(m, sbank, kw, 5, args.fp16)
                times.append(ms)
                samples.append(rend)
            extra["shard_emulation"] = {"shards": S, "ms": times, "max_ms": max(times), "mean_ms": sum(times) / S, "samples": samples,
                                        "imbalance_max_over_mean": max(times) / (sum(times) / S)}
        # --- round 3 legs: the reference's -O mode, a long run, configs[4] split 8 ways (emulated), configs[3] training steps, the occupancy sweep
        if not args.no_extras and native and args.workload == "lego" and not args.fp16 and F_main == 1:
            n = max(1, args.extra_steps)
            # --- round 4: the operator-API (drop-in) path on the driver's record: configs[1] and configs[2] under the reference's own loop
            dropin = {}
            for kind in ("nerf", "palette"):
                try:
                    dropin[kind] = dropin_leg(args, device, bank, kind, max(5, n // 2))
                except RuntimeError as e:
                    dropin[kind] = {"error": str(e)}
            for kind in ("nerf", "palette"):     # one step beyond the operator boundary: the same loop, the network's forward() served by the fused field (option A+)
                try:
                    dropin[kind + "_fuse_field"] = dropin_leg(args, device, bank, kind, max(5, n // 2), fuse_field=True)
                except RuntimeError as e:
                    dropin[kind + "_fuse_field"] = {"error": str(e)}
            extra["dropin"] = dropin
            try:     # SURVEY Appendix B's other regime: the translucent field (density_scale 0.02): every ray marches to `far`, ~63 M samples per frame
                targs = argparse.Namespace(**vars(args))
                targs.density_scale = 0.02
                tm = build_model(targs, device, "nerf")
                tm._fused.ray_order = m._fused.ray_order
                timed_frames(tm, bank, kw, 2, False)
                t_ms, t_rend = timed_frames(tm, bank, kw, max(3, n // 4), False, first_step=args.warmup)
                extra["translucent"] = {"ms_per_step": t_ms, "rendered_per_step": t_rend, "value": t_rend / (t_ms * 1e-3), "unit": "samples/s", "density_scale": 0.02,
                                        "steps": max(3, n // 4), "what": "configs[1] frame with the translucent field of SURVEY Appendix B (no early termination: every ray to `far`)"}
                del tm
            except RuntimeError as e:
                extra["translucent_error"] = str(e)
            try:     # `-O` = fp16 autocast + half tables (main_nerf.py:72-75): what every script of the reference runs
                fargs = argparse.Namespace(**vars(args))
                fargs.fp16 = True
                mm = build_model(fargs, device, "nerf")
                mm._fused.ray_order = m._fused.ray_order
                timed_frames(mm, bank, kw, 3, True)
                extra["fp16_mode_ms_per_step"], extra["fp16_mode_rendered_per_step"] = timed_frames(mm, bank, kw, n, True, first_step=args.warmup)
                # the lookup kernel of THAT mode against SURVEY 8(d)'s fp16 figure (588 B per sample): the frame of the headline's first timed step once more with the
                # in-library HIP events around every lookup launch (k_frame_grid_h1: half rows, the reference's at::Half accumulator)
                mm._fused.time_grid_kernel = True
                try:
                    ro16, rd16 = bank.get(args.warmup)
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                        r16 = mm.render(ro16, rd16, **kw)
                    torch.cuda.synchronize()
                finally:
                    mm._fused.time_grid_kernel = False
                h_ms, h_n, h_live = float(r16.get("grid_ms", 0.0)), int(r16.get("grid_launches", 0)), int(r16["rendered"].sum())
                if h_ms > 0 and h_n > 0:
                    h_ach = h_live * GRID_BYTES_PER_SAMPLE_FP16 / (h_ms * 1e-3) / 1e9
                    written = 12 + 16 * 8 * 2 * 2 + 32 * 4
                    out["roofline_fp16"] = {"bound": "hbm", "kernel": "k_frame_grid_h1 (device-driven frame loop, fp16 table: the reference's -O mode)", "achieved": h_ach, "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": h_ach / HBM_PEAK_GBS, "traffic": None, "launches": h_n, "avg_launch_ms": h_ms / h_n,
                                            "avg_live_samples_per_launch": h_live / h_n, "algorithmic_bytes_per_sample": GRID_BYTES_PER_SAMPLE_FP16,
                                            "algorithmic_bytes_per_launch": GRID_BYTES_PER_SAMPLE_FP16 * h_live / h_n,
                                            "frac_counting_the_fp32_outputs_it_writes": h_live * written / (h_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "ms_per_step": extra["fp16_mode_ms_per_step"],
                                            "note": "SURVEY 8(d)'s fp16 figure (half rows AND half outputs: 588 B); this kernel hands the field fp32 features (652 B written and read). The launch "
                                                    "lasts as long as the fp32 table's: the gather is bound by row requests per clock (one L2 line per hashed row), not by bytes -- "
                                                    "half the bytes in the same time is half the fraction"}
                del mm
            except RuntimeError as e:
                extra["fp16_mode_error"] = str(e)
            try:     # stability over seconds: 500 frames of the camera path, per-frame time from events (no sync inside the loop)
                L = 500
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
                import gc
                gc.collect()
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    evs[0].record()
                    for i in range(L):
                        ro, rd = bank.get(args.warmup + i)
                        with torch.no_grad():
                            m.render(ro, rd, **kw)
                        evs[i + 1].record()
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t0
                finally:
                    if gc_was_on:      # (main() re-enabled it after the timed region; the legs that follow run under the state they found)
                        gc.enable()
                ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(L))
                extra["long_run"] = {"steps": L, "seconds": wall, "ms_per_step": wall / L * 1e3, "step_ms": {"min": ms[0], "p05": ms[L // 20], "median": ms[L // 2], "p95": ms[L - L // 20], "max": ms[-1]}}
            except RuntimeError as e:
                extra["long_run_error"] = str(e)
            try:     # configs[4] (garden video frame) split into 8 tile shards rendered one after another on this GPU: the load balance an 8-GPU split would see
                gargs = parse(["--workload", "garden", "--no-cpu-baseline"])
                gm = build_model(gargs, device)
                gH, gW = gargs.wl["H"], gargs.wl["W"]
                gkw = dict(perturb=False, dt_gamma=gargs.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4, gui_mode=False)
                gargs.static_pose = True
                full_idx, _ = pdist.shard_indices(gH, gW, 0, 1)
                gbank = RayBank(gargs, 1, full_idx, device)
                gm._fused.ray_order = tile_ray_order(full_idx, gW, 8).to(device)
                timed_frames(gm, gbank, gkw, 2, False)
                full_ms1, full_mean1, full_rend = timed_frames_median(gm, gbank, gkw, 15)
                full_ms, full_mean, _ = timed_frames_queue(gm, gbank, gkw, 15)
                times, means, samples, times1 = [], [], [], []
                for sh in range(8):
                    sidx, _ = pdist.shard_indices(gH, gW, sh, 8)
                    sbank = RayBank(gargs, 1, sidx, device)
                    gm._fused.ray_order = tile_ray_order(sidx, gW, 8).to(device)
                    timed_frames(gm, sbank, gkw, 2, False)
                    ms1_, _, rend = timed_frames_median(gm, sbank, gkw, 21)
                    ms_, mean_, _ = timed_frames_queue(gm, sbank, gkw, 21)
                    times1.append(ms1_)
                    times.append(ms_)
                    means.append(mean_)
                    samples.append(rend)
                extra["garden_shard_emulation_8"] = {"full_frame_ms": full_ms, "full_frame_ms_mean": full_mean, "full_frame_samples": full_rend, "shard_ms": times, "shard_ms_mean": means,
                                                     "how": "round 6: a rank's frames through pipeline.render_queue (render_prepare / render_launch / render_finish over pnr_*_render_frame_submit / _finish): one frame on the device at a "
                                                            "time, the host's work for frame i + 1 under frame i's kernels -- what the N > 1 loop of this script does; the full frame is timed the same way",
                                                     "one_call_per_frame": {"full_frame_ms": full_ms1, "full_frame_ms_mean": full_mean1, "shard_ms": times1, "max_shard_ms": max(times1),
                                                                            "speedup_before_all_gather": full_ms1 / max(times1),
                                                                            "what": "rounds 1-5's way: m.render() per frame (the host turns around between two frames), device time between events"},
                                                     "timing": "median of 15 (full frame) / 21 (each shard) frames (host time between consecutive frames' completions); means beside them",
                                                     "shard_samples": samples,
                                                     "max_shard_ms": max(times), "imbalance_max_over_mean": max(times) / (sum(times) / 8),
                                                     "speedup_before_all_gather": full_ms / max(times),
                                                     # what the all-gather would add on 8 GPUs (an ESTIMATE from SURVEY 8e's link figure, not a measurement: this box has one GPU):
                                                     # every rank's packed rows go to its 7 peers over 7 xGMI links concurrently, ~153 GB/s per link
                                                     "all_gather_estimate": {k: {"floats_per_ray": K_, "bytes_per_rank": int(gH * gW / 8 * K_ * 4), "ms_at_153GBs_per_link": gH * gW / 8 * K_ * 4 / 153e9 * 1e3,
                                                                                 "speedup_with_it": full_ms / (max(times) + gH * gW / 8 * K_ * 4 / 153e9 * 1e3)}
                                                                             for k, K_ in (("video_rows_K24", 24), ("full_palette_output_set_K55", 55))},
                                                     "what": f"configs[4]: one {gH}x{gW} PaletteNeRF garden frame, its 8 interleaved-tile shards rendered one after another on this GPU"}
                del gm
            except RuntimeError as e:
                extra["garden_shard_emulation_8_error"] = str(e)
            for kind in ("palette", "nerf"):
                try:
                    extra[f"train_{kind}"] = training_leg(kind, device)
                    if kind == "palette":   # the same step with the loss as the reference's trainer writes it: what the fused tail replaces
                        t = training_leg(kind, device, steps=20, torch_loss=True)
                        extra[f"train_{kind}"]["with_torch_loss"] = {k: t[k] for k in ("wall_ms_per_step", "kernel_ms_per_step", "launches_per_step") if k in t}
                except RuntimeError as e:
                    extra[f"train_{kind}_error"] = str(e)
            try:
                extra["occupancy_sweep"] = occupancy_leg(device)
            except RuntimeError as e:
                extra["occupancy_sweep_error"] = str(e)
    if rank == 0 and extra:
        out["extra"] = extra
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
