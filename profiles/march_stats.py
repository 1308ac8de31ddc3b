"""Probe statistics of the native frame loop's march kernel (needs a -DPNR_MARCH_STATS build:
PNR_EXTRA_HIPCC_FLAGS=-DPNR_MARCH_STATS python -m palettenerf_amd.build --force)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from palettenerf_amd import _lib, scene

sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
m.march_mode = "native"
H = W = 800
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro.to(dev), rd.to(dev)
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    r = m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    lib.pnr_debug_march_stats(out, 1)
    lib.pnr_debug_march_max((ctypes.c_uint * 64)(), 1)
    r = m.render(ro, rd, perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
    lib.pnr_debug_march_stats(out, 1)
mx = (ctypes.c_uint * 64)()
lib.pnr_debug_march_max(mx, 1)
print("per-iteration max probes of a ray:", list(mx)[:36])
kd = (ctypes.c_uint * 512)()
lib.pnr_debug_march_kinds(kd)
print("kinds of a slowest ray per iteration [emit, cell, -, s8?, ...]:")
for it in range(30):
    print(it, list(kd)[it * 8:it * 8 + 8])
names = ["emit", "cell step (mixed brick)", "block jump", "-", "-", "jump refused: margins / cascade / step", "jump refused: exit plane's window taken", "jump refused: inner planes too (or memo)"]
for row, what in ((62, "first launch of a frame"), (63, "later launches")):
    tot = list(kd)[row * 8:row * 8 + 8]
    if sum(tot):
        print(f"probe kinds, {what} (both frames):", ", ".join(f"{n}: {v} ({100.0 * v / sum(tot):.1f}%)" for n, v in zip(names, tot) if v))
print("first launch:", list(out[0:4]), "later launches:", list(out[4:8]))
print("probes", out[0], "empty", out[1], "sum over waves of max-lane probes x64", out[2] * 64, "ray-launches", out[3], "rendered", int(r["rendered"].item()))

hist = (ctypes.c_uint * 64)()
if hasattr(lib, "pnr_debug_march_hist") and lib.pnr_debug_march_hist(hist) == 0:
    for name, h in (("first launch", list(hist)[:32]), ("later launches", list(hist)[32:])):
        tot = sum(h)
        if tot:
            tail = [sum(h[k + 1:]) / tot for k in range(12)]
            print(f"probes per ray, {name} ({tot} ray-launches over both frames): share with MORE than k probes, k = 0..11:", " ".join(f"{v * 100:.2f}%" for v in tail))
