import sys, numpy as np, torch
sys.path.insert(0, '.')
from palettenerf_amd import raymarching, _torch_glue
cuda = torch.device('cuda')
rng = np.random.default_rng(91)
N, n_alive, n_step = 3000, 1700, 4
alive = np.sort(rng.choice(N, n_alive, replace=False)).astype(np.int32)
M = n_alive * n_step
dev = lambda a: torch.from_numpy(a).to(cuda)
sig, rgb = dev((rng.random(M) * 60).astype(np.float32)), dev(rng.random((M, 3)).astype(np.float32))
dl = np.stack([rng.random(M) * 0.02 + 0.003, rng.random(M) * 0.05 + 0.003], 1).astype(np.float32)
dl[rng.random(M) < 0.1] = 0
dl = dev(dl)
chans = [3, 3, 4, 12, 12, 16]
ins = [dev(rng.standard_normal((M, c)).astype(np.float32)) for c in chans]
def state():
    g = torch.Generator().manual_seed(5)
    return dict(alive=dev(alive), t=torch.rand(N, generator=g).to(cuda), ws=(torch.rand(N, generator=g) * 0.7).to(cuda), dep=torch.rand(N, generator=g).to(cuda),
                img=torch.rand(N, 3, generator=g).to(cuda), outs=[torch.rand(N, c, generator=g).to(cuda) for c in chans])
def iteration(st, arm):
    if arm:
        raymarching.arm_flex_deferral()
    for c, i, o in zip(chans, ins, st["outs"]):
        raymarching.composite_rays_flex(n_alive, n_step, c, st["alive"], st["t"], sig, i, dl, st["ws"], o, 1e-4)
    raymarching.composite_rays(n_alive, n_step, st["alive"], st["t"], sig, rgb, dl, st["ws"], st["dep"], st["img"], 1e-4)
for use_prof in (False, True):
    a, b = state(), state()
    init = [o.clone() for o in a["outs"]]
    print("init equal", all(torch.equal(x, y) for x, y in zip(a["outs"], b["outs"])))
    iteration(a, False)
    prof = _torch_glue.profile_kernels(["pnr_composite_rays_flex", "pnr_composite_rays_flex_multi"]) if use_prof else None
    iteration(b, True)
    _torch_glue.profile_kernels(None)
    for ch, x, y, z in zip(chans, a["outs"], b["outs"], init):
        d = (x - y).abs().amax(dim=1)
        bad = torch.nonzero(d > 0).flatten().tolist()
        print("prof", use_prof, ch, torch.equal(x, y), bad[:10], "alive?" , [int(r in set(alive.tolist())) for r in bad[:10]])
        for r in bad[:2]:
            print("   a", x[r].tolist(), "\n   b", y[r].tolist(), "\n   init", z[r].tolist())
