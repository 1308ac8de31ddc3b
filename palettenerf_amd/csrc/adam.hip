// adam.hip -- the optimiser step of the training path as ONE launch over every parameter tensor.
//
// The reference trains with torch.optim.Adam(lr, betas=(0.9, 0.99), eps=1e-15) (main_nerf.py:113, main_palette.py:223); torch runs it as
// seven elementwise kernels per parameter tensor (lerp, mul, addcmul, sqrt, mul, add, addcdiv): ~75 launches and seven passes over the
// 50 MB hash tables per step.  Here every element of every tensor makes one trip: read p, g, m, v -- write p, m, v.
// The arithmetic is torch's, operation for operation and rounding for rounding (torch/optim/adam.py:_multi_tensor_adam, what
// torch.optim.Adam runs by default on the GPU; the forms its kernels use on this build were measured with profiles/micro/adam_probe.py:
// lerp = fma(w, g - m, m); addcmul = fma(alpha, g * g, v); division by the fp32 scalar; addcdiv = fma(alpha, m / denom, p)), so that a
// training run is bit-identical to one driven by torch.optim.Adam (tests/test_gpu_ops.py).
#include "pnr_common.hpp"

namespace pnr {

constexpr int kAdamMaxTensors = 32;
constexpr uint32_t kAdamThreads = 256, kAdamPerThread = 8, kAdamChunk = kAdamThreads * kAdamPerThread;

struct AdamTensor { float* p; const float* g; float* m; float* v; uint64_t n; };
struct AdamTable {
    AdamTensor t[kAdamMaxTensors];
    uint32_t first_chunk[kAdamMaxTensors + 1];   // chunk index at which every tensor starts (prefix sum of ceil(n / kAdamChunk))
    int count;
};
struct AdamScalars { float w1, beta2, c2, bc2_sqrt, inv_bc2_sqrt, eps, neg_step_size; float inv_grad_scale; int variant; };

__global__ void __launch_bounds__(kAdamThreads) k_adam(AdamTable tab, AdamScalars s) {
    int ti = 0;
    while (ti + 1 < tab.count && blockIdx.x >= tab.first_chunk[ti + 1]) ti++;   // <= 32 scalar compares
    const AdamTensor t = tab.t[ti];
    const uint64_t base = (uint64_t)(blockIdx.x - tab.first_chunk[ti]) * kAdamChunk;
#pragma unroll
    for (uint32_t k = 0; k < kAdamPerThread; k++) {
        const uint64_t i = base + (uint64_t)k * kAdamThreads + threadIdx.x;
        if (i >= t.n) break;
        float g = t.g[i];
        if (s.inv_grad_scale != 1.0f) g *= s.inv_grad_scale;        // GradScaler.unscale_ folded in (a separate torch kernel otherwise)
        float m = t.m[i], v = t.v[i];
        // exp_avg.lerp_(grad, 1 - beta1): |weight| < 0.5 -> self + weight * (end - self)           (ATen/native/Lerp.h)
        m = fmaf(s.w1, g - m, m);
        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2): a + alpha * (b * c), the product b * c rounded first
        // (PointwiseOpsKernel.cu; which of the candidate forms torch's build uses was measured: profiles/micro/adam_probe.py)
        v = v * s.beta2;
        v = (s.variant & 2) ? fmaf(s.c2 * g, g, v) : fmaf(s.c2, g * g, v);
        // denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps).  torch's default (foreach) implementation divides by the fp32 scalar;
        // its single-tensor kernels turn a division by a host scalar into a multiplication by the fp32 reciprocal -- variant bit 0
        const float denom = ((s.variant & 1) ? sqrtf(v) * s.inv_bc2_sqrt : sqrtf(v) / s.bc2_sqrt) + s.eps;
        // param.addcdiv_(exp_avg, denom, value = -step_size): a + alpha * (b / c)
        const float q = m / denom;
        t.p[i] = (s.variant & 4) ? t.p[i] + s.neg_step_size * q : fmaf(s.neg_step_size, q, t.p[i]);
        t.m[i] = m; t.v[i] = v;
    }
}

}  // namespace pnr

using namespace pnr;

extern int g_opt_adam_variant;

extern "C" {

uint32_t pnr_adam_max_tensors(void) { return kAdamMaxTensors; }

int pnr_adam_step(const pnr_adam_tensor* tensors, uint32_t count, const pnr_adam_scalars* sc, pnr_stream_t stream) {
    if (count == 0) return PNR_OK;
    if (!tensors || !sc) return PNR_ERR_INVALID;
    if (count > (uint32_t)kAdamMaxTensors) return PNR_ERR_UNSUPPORTED;
    AdamTable tab;
    uint64_t chunks = 0;
    int used = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (tensors[i].n == 0) continue;
        if (!tensors[i].param || !tensors[i].grad || !tensors[i].exp_avg || !tensors[i].exp_avg_sq) return PNR_ERR_INVALID;
        tab.t[used] = AdamTensor{tensors[i].param, tensors[i].grad, tensors[i].exp_avg, tensors[i].exp_avg_sq, tensors[i].n};
        tab.first_chunk[used] = (uint32_t)chunks;
        chunks += (tensors[i].n + kAdamChunk - 1) / kAdamChunk;
        used++;
    }
    if (used == 0) return PNR_OK;
    if (chunks > 0x7fffffffull) return PNR_ERR_UNSUPPORTED;
    tab.first_chunk[used] = (uint32_t)chunks;
    tab.count = used;
    AdamScalars s;
    s.w1 = sc->one_minus_beta1; s.beta2 = sc->beta2; s.c2 = sc->one_minus_beta2; s.bc2_sqrt = sc->bias_correction2_sqrt; s.inv_bc2_sqrt = 1.0f / sc->bias_correction2_sqrt; s.eps = sc->eps;
    s.neg_step_size = sc->neg_step_size;
    s.inv_grad_scale = sc->inv_grad_scale > 0.0f ? sc->inv_grad_scale : 1.0f;
    s.variant = g_opt_adam_variant;
    hipLaunchKernelGGL(k_adam, dim3((uint32_t)chunks), dim3(kAdamThreads), 0, as_stream(stream), tab, s);
    return check_launch();
}

}  // extern "C"
