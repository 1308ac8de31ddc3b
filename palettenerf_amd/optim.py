"""The training path's optimiser step as one HIP launch: `Adam` is a drop-in for `torch.optim.Adam` as the reference constructs it
(main_nerf.py:113, main_palette.py:223: lr 1e-2, betas (0.9, 0.99), eps 1e-15) -- same constructor, param groups, state_dict keys
(`step`, `exp_avg`, `exp_avg_sq`) and, bit for bit, the same updates (pnr_adam_step) -- so schedulers, GradScaler and checkpoints of the
reference's trainer work unchanged.  torch runs seven elementwise kernels per parameter tensor (~75 launches and seven passes over the
50 MB hash tables per step); this makes one pass in one launch.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("palettenerf_amd.optim.Adam covers the reference's configuration: no weight decay, no amsgrad")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.grad_scale = None   # set to GradScaler's scale (a float) to fold unscale_ into the step; None / 1.0 = gradients as they are

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        cap = int(lib.pnr_adam_max_tensors())
        f32 = np.float32
        # ONE launch for every tensor that shares its hyperparameters and step count (the reference's get_params makes ten param groups with the
        # same lr / betas / eps: torch runs them group by group); tensors that joined later have their own bias corrections and go separately
        batches = {}
        todo = []
        for group in self.param_groups:      # first pass: every check, nothing touched -- a rejected tensor must not leave earlier ones a step ahead
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("palettenerf_amd.optim.Adam: dense fp32 CUDA(HIP) parameters only (no CPU fallback)")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("palettenerf_amd.optim.Adam: parameters and gradients must be contiguous")
                todo.append((group, p))
        for group, p in todo:
            beta1, beta2 = group["betas"]
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)   # host tensor, as torch keeps it for non-capturable Adam
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            key = (float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), int(st["step"].item()), p.device.index)
            batches.setdefault(key, []).append((p, st))
        for (lr, beta1, beta2, eps, step, _dev), items in batches.items():
            # the host scalars exactly as torch/optim/adam.py forms them (Python floats), narrowed where its CUDA kernels narrow them
            bias_correction1 = 1 - beta1 ** step
            bias_correction2 = 1 - beta2 ** step
            step_size = lr / bias_correction1
            bias_correction2_sqrt = bias_correction2 ** 0.5
            sc = _lib.AdamScalars()
            sc.one_minus_beta1, sc.beta2, sc.one_minus_beta2 = 1 - beta1, beta2, 1 - beta2
            sc.bias_correction2_sqrt = bias_correction2_sqrt
            sc.eps, sc.neg_step_size = eps, -step_size
            sc.inv_grad_scale = 1.0 if not self.grad_scale else float(f32(1.0) / f32(self.grad_scale))
            for i in range(0, len(items), cap):
                chunk = items[i:i + cap]
                arr = (_lib.AdamTensor * len(chunk))()
                for k, (p, st) in enumerate(chunk):
                    arr[k].param, arr[k].grad = p.data_ptr(), p.grad.data_ptr()
                    arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                with torch.cuda.device(_dev):   # the launch goes to the tensors' device, on torch's current stream there
                    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    _lib.check(lib.pnr_adam_step(arr, ctypes.c_uint32(len(chunk)), ctypes.byref(sc), stream), "pnr_adam_step")
            for p, _ in items:
                torch.autograd.graph.increment_version(p)   # written by a raw kernel: caches keyed on the version (packed blobs, pair tables) must see it
        return loss
