// palette_heads.hip -- the two heads of PaletteNetwork.color for TRAINING, one launch each way (palette/network.py:262-268):
//   offsets_radiance = offsets_radiance_net(h)                       nn.Linear(geo_feat_dim, 3 nb + 1), the only layer with a bias
//   omega            = softplus(omega_net.0(h)) + 0.05;  omega /= omega.sum(-1)
// In the reference that is two library GEMMs, a bias add, softplus, add, row sum and divide forward, and the same again plus two more GEMMs
// backward (~20 launches over M ~ 6e5 samples for two 15-wide dot-product blocks).  Here: one thread per sample, the (3 nb + 1 + nb) x 16
// weight block in LDS (broadcast float4 reads), outputs staged through an LDS tile so that a wave writes consecutive addresses.
// The backward recomputes softplus/normalise from h, returns dL/dh and the concatenated pre-activation gradient
// dz = [d offsets_radiance | d omega_pre] which pnr_linear_wgrad / pnr_linear_bgrad reduce to the weight and bias gradients.
// HBM-bound: forward (in + 4 nb + 1) * 4 B, backward (2 in + 2 (4 nb + 1)) * 4 B per sample (+ the dz re-read by the weight gradient).
#include "pnr_common.hpp"

namespace pnr {

constexpr uint32_t kHeadsIn = 16;                         // h columns, padded with zeros (geo_feat_dim is 15)
constexpr uint32_t kHeadsMaxBasis = 10;                   // PNR_MAX_BASIS
constexpr uint32_t kHeadsMaxRows = 4 * kHeadsMaxBasis + 1;
constexpr uint32_t kHeadsTileStride = (kHeadsMaxRows + kHeadsIn) | 1u;

__device__ __forceinline__ float heads_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }        // F.softplus, beta 1, threshold 20
__device__ __forceinline__ float heads_softplus_grad(float x) { if (x > 20.0f) return 1.0f; const float z = expf(x); return z / (z + 1.0f); }

__device__ __forceinline__ void heads_load_weights(float* __restrict__ w, float* __restrict__ bias, const float* __restrict__ w_or,
                                                   const float* __restrict__ b_or, const float* __restrict__ w_om, uint32_t nb, uint32_t in_dim) {
    const uint32_t orw = 3 * nb + 1, rows = orw + nb;
    for (uint32_t idx = threadIdx.x; idx < rows * kHeadsIn; idx += blockDim.x) {
        const uint32_t r = idx / kHeadsIn, k = idx % kHeadsIn;
        float v = 0.0f;
        if (k < in_dim) v = r < orw ? w_or[r * in_dim + k] : w_om[(r - orw) * in_dim + k];
        w[idx] = v;
    }
    if (b_or && threadIdx.x < orw) bias[threadIdx.x] = b_or[threadIdx.x];
}

__device__ __forceinline__ void heads_load_row(float (&x)[kHeadsIn], const float* __restrict__ h, uint32_t i, uint32_t M, uint32_t in_dim) {
#pragma unroll
    for (uint32_t k = 0; k < kHeadsIn; k++) x[k] = (i < M && k < in_dim) ? h[(size_t)i * in_dim + k] : 0.0f;
}

__device__ __forceinline__ float heads_dot(const float (&x)[kHeadsIn], const float* __restrict__ wrow, float acc) {
    const float4* w4 = reinterpret_cast<const float4*>(wrow);
#pragma unroll
    for (uint32_t q = 0; q < kHeadsIn / 4; q++) {
        const float4 w = w4[q];
        acc = fmaf(x[4 * q + 0], w.x, acc);
        acc = fmaf(x[4 * q + 1], w.y, acc);
        acc = fmaf(x[4 * q + 2], w.z, acc);
        acc = fmaf(x[4 * q + 3], w.w, acc);
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_palette_heads_fwd(uint32_t M, uint32_t nb, uint32_t in_dim, const float* __restrict__ h,
                                                           const float* __restrict__ w_or, const float* __restrict__ b_or,
                                                           const float* __restrict__ w_om, float* __restrict__ offrad, float* __restrict__ omega) {
    __shared__ __attribute__((aligned(16))) float w[kHeadsMaxRows * kHeadsIn];
    __shared__ float bias[3 * kHeadsMaxBasis + 1];
    __shared__ float tile[256 * (kHeadsMaxRows | 1u)];
    heads_load_weights(w, bias, w_or, b_or, w_om, nb, in_dim);
    __syncthreads();
    const uint32_t orw = 3 * nb + 1, rows = orw + nb, ws = rows | 1u;
    const uint32_t ntiles = (M + 255) / 256;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t row0 = t * 256, i = row0 + threadIdx.x, nrows = M - row0 < 256u ? M - row0 : 256u;
        float x[kHeadsIn];
        heads_load_row(x, h, i, M, in_dim);
        float* out = tile + threadIdx.x * ws;
        for (uint32_t j = 0; j < orw; j++) out[j] = heads_dot(x, w + j * kHeadsIn, bias[j]);
        float sum = 0.0f;
        for (uint32_t b = 0; b < nb; b++) {
            const float s = heads_softplus(heads_dot(x, w + (orw + b) * kHeadsIn, 0.0f)) + 0.05f;
            out[orw + b] = s;
            sum += s;
        }
        for (uint32_t b = 0; b < nb; b++) out[orw + b] = out[orw + b] / sum;
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < nrows * orw; idx += 256) offrad[(size_t)row0 * orw + idx] = tile[(idx / orw) * ws + idx % orw];
        for (uint32_t idx = threadIdx.x; idx < nrows * nb; idx += 256) omega[(size_t)row0 * nb + idx] = tile[(idx / nb) * ws + orw + idx % nb];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_palette_heads_bwd(uint32_t M, uint32_t nb, uint32_t in_dim, const float* __restrict__ h,
                                                           const float* __restrict__ w_or, const float* __restrict__ w_om,
                                                           const float* __restrict__ d_offrad, const float* __restrict__ d_omega,
                                                           float* __restrict__ dh, float* __restrict__ dz) {
    __shared__ __attribute__((aligned(16))) float w[kHeadsMaxRows * kHeadsIn];
    __shared__ float tile[256 * kHeadsTileStride];
    heads_load_weights(w, nullptr, w_or, nullptr, w_om, nb, in_dim);      // the bias plays no part in the backward
    __syncthreads();
    const uint32_t orw = 3 * nb + 1, rows = orw + nb, ws = (rows + kHeadsIn) | 1u;
    const uint32_t ntiles = (M + 255) / 256;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t row0 = t * 256, i = row0 + threadIdx.x, nrows = M - row0 < 256u ? M - row0 : 256u;
        float x[kHeadsIn];
        heads_load_row(x, h, i, M, in_dim);
        float* out = tile + threadIdx.x * ws;
        float g[kHeadsIn];
#pragma unroll
        for (uint32_t k = 0; k < kHeadsIn; k++) g[k] = 0.0f;
        // offsets_radiance head: dz_j = d_offrad_j
        for (uint32_t j = 0; j < orw; j++) {
            const float d = i < M ? d_offrad[(size_t)i * orw + j] : 0.0f;
            out[j] = d;
            const float4* w4 = reinterpret_cast<const float4*>(w + j * kHeadsIn);
#pragma unroll
            for (uint32_t q = 0; q < kHeadsIn / 4; q++) {
                const float4 wv = w4[q];
                g[4 * q + 0] = fmaf(d, wv.x, g[4 * q + 0]);
                g[4 * q + 1] = fmaf(d, wv.y, g[4 * q + 1]);
                g[4 * q + 2] = fmaf(d, wv.z, g[4 * q + 2]);
                g[4 * q + 3] = fmaf(d, wv.w, g[4 * q + 3]);
            }
        }
        // omega head: omega_b = s_b / S, s_b = softplus(z_b) + 0.05  ->  dL/ds_b = (g_b - sum_c g_c omega_c) / S
        float sum = 0.0f;
        for (uint32_t b = 0; b < nb; b++) {
            const float z = heads_dot(x, w + (orw + b) * kHeadsIn, 0.0f);
            out[orw + b] = z;
            sum += heads_softplus(z) + 0.05f;
        }
        float gdot = 0.0f;
        for (uint32_t b = 0; b < nb; b++) {
            const float gb = i < M ? d_omega[(size_t)i * nb + b] : 0.0f;
            gdot = fmaf(gb, (heads_softplus(out[orw + b]) + 0.05f) / sum, gdot);
        }
        for (uint32_t b = 0; b < nb; b++) {
            const float gb = i < M ? d_omega[(size_t)i * nb + b] : 0.0f;
            const float d = (gb - gdot) / sum * heads_softplus_grad(out[orw + b]);
            out[orw + b] = d;
            const float4* w4 = reinterpret_cast<const float4*>(w + (orw + b) * kHeadsIn);
#pragma unroll
            for (uint32_t q = 0; q < kHeadsIn / 4; q++) {
                const float4 wv = w4[q];
                g[4 * q + 0] = fmaf(d, wv.x, g[4 * q + 0]);
                g[4 * q + 1] = fmaf(d, wv.y, g[4 * q + 1]);
                g[4 * q + 2] = fmaf(d, wv.z, g[4 * q + 2]);
                g[4 * q + 3] = fmaf(d, wv.w, g[4 * q + 3]);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kHeadsIn; k++) out[rows + k] = g[k];
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < nrows * rows; idx += 256) dz[(size_t)row0 * rows + idx] = tile[(idx / rows) * ws + idx % rows];
        if (dh)
            for (uint32_t idx = threadIdx.x; idx < nrows * in_dim; idx += 256)
                dh[(size_t)row0 * in_dim + idx] = tile[(idx / in_dim) * ws + rows + idx % in_dim];
        __syncthreads();
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_palette_heads_forward(const float* h, const float* w_offsets_radiance, const float* b_offsets_radiance, const float* w_omega, uint32_t M,
                              uint32_t num_basis, uint32_t in_dim, float* offsets_radiance, float* omega, pnr_stream_t stream) {
    if (num_basis == 0 || num_basis > kHeadsMaxBasis || in_dim == 0 || in_dim > kHeadsIn) return PNR_ERR_UNSUPPORTED;
    if (M == 0) return PNR_OK;
    if (!h || !w_offsets_radiance || !b_offsets_radiance || !w_omega || !offsets_radiance || !omega) return PNR_ERR_INVALID;
    const uint32_t blocks = cdiv(M, 256);
    hipLaunchKernelGGL(k_palette_heads_fwd, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, as_stream(stream), M, num_basis, in_dim, h,
                       w_offsets_radiance, b_offsets_radiance, w_omega, offsets_radiance, omega);
    return check_launch();
}

int pnr_palette_heads_backward(const float* h, const float* w_offsets_radiance, const float* w_omega, const float* grad_offsets_radiance,
                               const float* grad_omega, uint32_t M, uint32_t num_basis, uint32_t in_dim, float* grad_h, float* grad_pre,
                               pnr_stream_t stream) {
    if (num_basis == 0 || num_basis > kHeadsMaxBasis || in_dim == 0 || in_dim > kHeadsIn) return PNR_ERR_UNSUPPORTED;
    if (M == 0) return PNR_OK;
    if (!h || !w_offsets_radiance || !w_omega || !grad_offsets_radiance || !grad_omega || !grad_pre) return PNR_ERR_INVALID;
    const uint32_t blocks = cdiv(M, 256);
    hipLaunchKernelGGL(k_palette_heads_bwd, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, as_stream(stream), M, num_basis, in_dim, h,
                       w_offsets_radiance, w_omega, grad_offsets_radiance, grad_omega, grad_h, grad_pre);
    return check_launch();
}

}  // extern "C"
