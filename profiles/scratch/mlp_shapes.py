#!/usr/bin/env python3
"""fused MLP forward/backward time by shape (626 k rows), split-fp16 and fp32 launches"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
from palettenerf_amd import mlp, _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B = 626000
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for dims, act, out, need_dx in (((31, 64, 64, 3), F.relu, torch.sigmoid, True), ((32, 64, 64, 3), F.relu, torch.sigmoid, True), ((32, 64, 64, 3), F.relu, None, True),
                                ((32, 64, 64, 16), F.relu, None, True), ((31, 64, 64, 3), F.relu, None, True), ((31, 64, 64, 3), F.relu, torch.sigmoid, False), ((32, 64, 16), F.relu, None, False), ((35, 64, 15), F.elu, None, True)):
    net = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)]).to(dev)
    x = torch.randn(B, dims[0], device=dev, requires_grad=need_dx)
    wy = torch.randn(B, dims[-1], device=dev)
    for opt in (1, 0):
        lib.pnr_set_option(b"mlp_f16x3", opt)
        y = mlp.run_mlp(net, x, act, out)
        def fwd(): mlp.run_mlp(net, x, act, out)
        def bwd():
            y.backward(wy, retain_graph=True)
        print(dims, act.__name__, "sigmoid" if out else "-", "dx" if need_dx else "no-dx", "f16x3" if opt else "fp32", f"fwd {timed(fwd):.0f} us  bwd {timed(bwd):.0f} us", flush=True)
lib.pnr_set_option(b"mlp_f16x3", 1)
