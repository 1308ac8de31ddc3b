#!/usr/bin/env python3
"""Forward + backward of one fused training MLP (csrc/mlp.hip) at the training batch of configs[3] (627 k samples): the colour net
31 -> 64 -> 64 -> 3 (ReLU) by default.  Used under rocprofv3 (kernel trace / PMC passes, profiles/pmc_pass.sh with PNR_PMC_SCRIPT)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from palettenerf_amd import mlp  # noqa: E402


def main():
    cuda = torch.device("cuda:0")
    dims = tuple(int(v) for v in os.environ.get("PNR_MLP_DIMS", "31,64,64,3").split(","))
    act = F.elu if os.environ.get("PNR_MLP_ACT") == "elu" else F.relu
    B = 626688
    net = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)]).to(cuda)
    x = torch.randn(B, dims[0], device=cuda, requires_grad=os.environ.get("PNR_MLP_DX") == "1")
    wy = torch.randn(B, dims[-1], device=cuda)

    def run():
        mlp.run_mlp(net, x, act).backward(wy)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"mlp {dims} {act.__name__}: forward + backward {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per call (pack, two launches, dW reduce)")


if __name__ == "__main__":
    main()
