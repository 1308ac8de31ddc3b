import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from palettenerf_amd import _lib, network, renderer, scene
from palettenerf_amd.fused import PaletteFieldFused
cuda = torch.device("cuda:0")
m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
scene.seed_field_(m, 5)
m = m.to(cuda).eval()
f = PaletteFieldFused(m)
lib = _lib.load()
g = torch.Generator().manual_seed(0)
B = 5000
x = (torch.rand(B, 3, generator=g) * 1.2 - 0.6).to(cuda)
d = torch.randn(B, 3, generator=g); d = (d / d.norm(dim=1, keepdim=True)).to(cuda)
perm = torch.randperm(B, generator=g).to(cuda)
out = {}
with torch.no_grad():
    for w12 in (0, 1):
        lib.pnr_set_option(b"palette_waves12", w12)
        out[(w12, "id")] = f(x, d)
        s, c, a = f(x[perm].contiguous(), d[perm].contiguous())
        inv = torch.empty_like(perm); inv[perm] = torch.arange(B, device=cuda)
        out[(w12, "perm")] = (s[inv], c[inv], a[inv])
for k, v in out.items():
    base = out[(0, "id")]
    print(k, [float((p - q).abs().max()) for p, q in zip(v, base)], [int((p != q).sum()) for p, q in zip(v, base)])
# which rows differ between 12-wave id and 8-wave id?
s12, c12, a12 = out[(1, "id")]; s8, c8, a8 = out[(0, "id")]
bad = (c12 != c8).any(dim=1).nonzero()[:, 0]
print("rows differing:", bad[:40].tolist(), "count", bad.numel(), "rows mod 32:", sorted(set((bad % 32).tolist()))[:40], "rows//32 mod 12:", sorted(set(((bad // 32) % 12).tolist())))
