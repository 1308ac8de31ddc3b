"""Which roundings do torch's Adam kernels apply on this build?  Stage by stage, against candidates evaluated in float64 (fma = one rounding)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = 1 << 16
p0 = (torch.rand(n, generator=g) - 0.5).to(dev)
gr = torch.randn(n, generator=g).to(dev)
m0 = (torch.randn(n, generator=g) * 0.1).to(dev)
v0 = (torch.rand(n, generator=g) * 0.01).to(dev)
beta1, beta2, lr, eps, step = 0.9, 0.99, 1e-2, 1e-15, 3
f64 = lambda t: t.double()
r32 = lambda t: t.float()
def report(name, got, cands):
    print(name, {k: int((got != v).sum()) for k, v in cands.items()})
# lerp
m1 = m0.clone().lerp_(gr, 1 - beta1)
w = torch.tensor(1 - beta1, dtype=torch.float32, device=dev)
report("lerp", m1, {"fma(w,g-m,m)": r32(f64(w) * f64(gr - m0) + f64(m0)), "m+round(w*(g-m))": m0 + w * (gr - m0),
                    "w_double": r32((1 - beta1) * f64(gr - m0) + f64(m0))})
# mul + addcmul
v1a = v0.clone().mul_(beta2)
b2 = torch.tensor(beta2, dtype=torch.float32, device=dev)
report("mul", v1a, {"v*f32(beta2)": v0 * b2, "double": r32(f64(v0) * beta2)})
v1 = v1a.clone().addcmul_(gr, gr, value=1 - beta2)
c2 = torch.tensor(1 - beta2, dtype=torch.float32, device=dev)
report("addcmul", v1, {"fma(c2*g,g,v)": r32(f64(c2 * gr) * f64(gr) + f64(v1a)), "v+round((c2*g)*g)": v1a + (c2 * gr) * gr,
                       "fma(c2,g*g,v)": r32(f64(c2) * f64(gr * gr) + f64(v1a)), "v+c2*(g*g)": v1a + c2 * (gr * gr),
                       "all_double": r32(f64(c2) * f64(gr) * f64(gr) + f64(v1a)), "fma(g, c2*g)": r32(f64(gr) * f64(c2 * gr) + f64(v1a)),
                       "c2 double": r32((1 - beta2) * f64(gr) * f64(gr) + f64(v1a))})
bc2s = (1 - beta2 ** step) ** 0.5
sq = v1.sqrt()
report("sqrt", sq, {"f64 sqrt": r32(f64(v1).sqrt())})
d1 = sq / bc2s
inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(bc2s, dtype=torch.float32)
report("div scalar", d1, {"mul by f32 reciprocal": sq * inv.to(dev), "true div f32": sq / torch.tensor(bc2s, dtype=torch.float32, device=dev),
                          "double div": r32(f64(sq) / bc2s), "mul by double recip": r32(f64(sq) * (1.0 / bc2s))})
d2 = d1.clone().add_(eps)
report("add eps", d2, {"f32": d1 + torch.tensor(eps, dtype=torch.float32, device=dev)})
ss = lr / (1 - beta1 ** step)
p1 = p0.clone().addcdiv_(m1, d2, value=-ss)
a = torch.tensor(-ss, dtype=torch.float32, device=dev)
q = m1 / d2
report("addcdiv", p1, {"fma(a,m/d,p)": r32(f64(a) * f64(q) + f64(p0)), "p+round(a*(m/d))": p0 + a * q, "fma(a*m, 1/d..)": r32(f64(a * m1) / f64(d2) + f64(p0)),
                       "p+(a*m)/d": p0 + (a * m1) / d2, "double all": r32(f64(a) * f64(m1) / f64(d2) + f64(p0))})
# whole optimizer: foreach or not?
pp = torch.nn.Parameter(p0.clone()); pp.grad = gr.clone()
opt = torch.optim.Adam([pp], lr=lr, betas=(beta1, beta2), eps=eps)
print("defaults", opt.defaults.get("foreach"), opt.defaults.get("fused"), opt.defaults.get("capturable"))
# the foreach forms torch.optim.Adam uses by default on the GPU (_multi_tensor_adam)
fe = [sq.clone(), sq[:1000].clone()]
torch._foreach_div_(fe, [bc2s, bc2s])
report("_foreach_div_ scalarlist", fe[0], {"mul by f32 reciprocal": sq * inv.to(dev), "true div f32": sq / torch.tensor(bc2s, dtype=torch.float32, device=dev)})
fe = [m0.clone(), m0[:1000].clone()]
torch._foreach_lerp_(fe, [gr, gr[:1000]], 1 - beta1)
report("_foreach_lerp_", fe[0], {"fma(w,g-m,m)": r32(f64(w) * f64(gr - m0) + f64(m0)), "m+round(w*(g-m))": m0 + w * (gr - m0)})
fe = [v1a.clone(), v1a[:1000].clone()]
torch._foreach_addcmul_(fe, [gr, gr[:1000]], [gr, gr[:1000]], 1 - beta2)
report("_foreach_addcmul_", fe[0], {"fma(c2,g*g,v)": r32(f64(c2) * f64(gr * gr) + f64(v1a)), "v+c2*(g*g)": v1a + c2 * (gr * gr), "fma(c2*g,g,v)": r32(f64(c2 * gr) * f64(gr) + f64(v1a))})
fe = [p0.clone(), p0[:1000].clone()]
torch._foreach_addcdiv_(fe, [m1, m1[:1000]], [d2, d2[:1000]], [-ss, -ss])
report("_foreach_addcdiv_ scalarlist", fe[0], {"fma(a,m/d,p)": r32(f64(a) * f64(q) + f64(p0)), "p+round(a*(m/d))": p0 + a * q})
fe = [v0.clone(), v0[:1000].clone()]
torch._foreach_mul_(fe, beta2)
report("_foreach_mul_", fe[0], {"v*f32(beta2)": v0 * b2})
fe = [d1.clone(), d1[:1000].clone()]
torch._foreach_add_(fe, eps)
report("_foreach_add_", fe[0], {"f32": d1 + torch.tensor(eps, dtype=torch.float32, device=dev)})
# and the optimiser itself, one step from (m0, v0) state
for foreach in (None, False, True):
    pp = torch.nn.Parameter(p0.clone()); pp.grad = gr.clone()
    opt = torch.optim.Adam([pp], lr=lr, betas=(beta1, beta2), eps=eps, foreach=foreach)
    opt.state[pp] = {"step": torch.tensor(float(step - 1)), "exp_avg": m0.clone(), "exp_avg_sq": v0.clone()}
    opt.step()
    report(f"Adam(foreach={foreach}) param", pp.detach(), {"slow-path chain": p1})
    dd = (v1.sqrt() / torch.tensor(bc2s, dtype=torch.float32, device=dev)).add_(eps)
    report(f"Adam(foreach={foreach}) param", pp.detach(), {"true-division chain": r32(f64(a) * f64(m1 / dd) + f64(p0))})
