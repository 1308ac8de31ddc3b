// linear.hip -- weight gradient of the fields' bias-free dense layers for training:  dW[o][i] = sum_b dY[b][o] * X[b][i]
// with B = hundreds of thousands of samples and in, out <= 64 (nerf/network.py:60-93, palette/network.py:60-153 build every
// layer as nn.Linear(in, out, bias=False); autograd sends this product to the BLAS library, whose tall-skinny path runs at
// 1.0-1.5 ms per layer on MI355X -- 40 % of a PaletteNeRF training step).
//
// MI355X formulation: the reduction index (the sample) is the MFMA k dimension.  v_mfma_f32_32x32x2_f32 takes, per lane,
// A[i = lane % 32][k = lane / 32] and B[k = lane / 32][j = lane % 32]: with k = two consecutive samples both operands are the
// two samples' rows exactly as they lie in memory (lanes 0-31 read 32 consecutive channels of sample 2p, lanes 32-63 of
// sample 2p+1: two 128-byte segments per load).  A wave streams sample pairs and keeps the whole dW (up to 2 x 2 tiles of
// 32 x 32) in accumulator registers; products are exact fp32 fma chains.  Waves are reduced through LDS per workgroup, the
// workgroup partials by a second tiny launch in a fixed order (deterministic, no atomics).  The kernel is HBM-bound: it reads
// X and dY once (B * (in + out) * 4 bytes).
#include "pnr_common.hpp"
#include <hip/hip_fp16.h>

namespace pnr {

typedef float lin_f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t kLinThreads = 256;
constexpr uint32_t kLinWaves = kLinThreads / PNR_WAVE;

template <typename T>
__device__ __forceinline__ float lin_load(const T* __restrict__ p, size_t i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float lin_load<__half>(const __half* __restrict__ p, size_t i) { return __half2float(p[i]); }

template <typename TX, typename TY, int TI, int TO>   // TI / TO: 32-wide tiles of the input / output dimension (1 or 2)
__global__ void __launch_bounds__(kLinThreads) k_linear_wgrad(const TX* __restrict__ x, const TY* __restrict__ dy, uint32_t B, uint32_t in_dim,
                                                              uint32_t out_dim, float* __restrict__ partial /* [gridDim.x][out_dim * in_dim] */) {
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    const uint32_t c = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    lin_f32x16 acc[TO][TI];
#pragma unroll
    for (int o = 0; o < TO; o++)
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[o][i][r] = 0.0f;
    const uint32_t npairs = (B + 1) / 2;
    const uint32_t wave_global = blockIdx.x * kLinWaves + wave, nwaves = gridDim.x * kLinWaves;
    constexpr int U = 4;  // sample pairs in flight per wave
    for (uint32_t p0 = wave_global * U; p0 < npairs; p0 += nwaves * U) {
        float xv[U][TI], yv[U][TO];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t b = (p0 + u) * 2 + half;
            const bool ok = b < B;
#pragma unroll
            for (int i = 0; i < TI; i++) xv[u][i] = (ok && i * 32 + c < in_dim) ? (x ? lin_load<TX>(x, (size_t)b * in_dim + i * 32 + c) : 1.0f) : 0.0f;
#pragma unroll
            for (int o = 0; o < TO; o++) yv[u][o] = (ok && o * 32 + c < out_dim) ? lin_load<TY>(dy, (size_t)b * out_dim + o * 32 + c) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int o = 0; o < TO; o++)
#pragma unroll
                for (int i = 0; i < TI; i++) acc[o][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[u][o], xv[u][i], acc[o][i], 0, 0, 0);
    }
    // workgroup reduction through LDS, one tile at a time: element (row, col) of a 32x32 D fragment sits in register r of lane
    // with col = lane % 32, row = 8 (r / 4) + 4 (lane / 32) + r % 4
    __shared__ float red[kLinWaves][32 * 32];
    float* out = partial + (size_t)blockIdx.x * out_dim * in_dim;
#pragma unroll
    for (int o = 0; o < TO; o++)
#pragma unroll
        for (int i = 0; i < TI; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) red[wave][(8 * (r / 4) + 4 * half + (r % 4)) * 32 + c] = acc[o][i][r];
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < 32 * 32; e += kLinThreads) {
                float s = red[0][e];
#pragma unroll
                for (int wv = 1; wv < (int)kLinWaves; wv++) s += red[wv][e];
                const uint32_t row = o * 32 + e / 32, col = i * 32 + e % 32;
                if (row < out_dim && col < in_dim) out[(size_t)row * in_dim + col] = s;
            }
            __syncthreads();
        }
}

// dw[e] (+)= sum over the workgroup partials, fixed order: 32 elements x 8 partial groups per workgroup (128-byte coalesced rows),
// group g sums partials g, g + 8, ...; the 8 group sums are added in order.
__global__ void __launch_bounds__(256) k_linear_wgrad_reduce(const float* __restrict__ partial, uint32_t nparts, uint32_t n, float* __restrict__ dw,
                                                             int accumulate) {
    __shared__ float red[8][32];
    const uint32_t c = threadIdx.x & 31u, g = threadIdx.x >> 5;
    const uint32_t e = blockIdx.x * 32 + c;
    float s = 0.0f;
    if (e < n)
        for (uint32_t p = g; p < nparts; p += 8) s += partial[(size_t)p * n + e];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && e < n) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; k++) t += red[k][c];
        dw[e] = accumulate ? dw[e] + t : t;
    }
}

static uint32_t wgrad_blocks(uint32_t B) {
    const uint32_t want = cdiv(cdiv(B, 2), kLinWaves * 4 * 4);  // >= 4 rounds of 4 pairs per wave before adding workgroups
    return want < 1 ? 1 : (want > 512 ? 512 : want);
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_linear_wgrad_workspace_bytes(uint32_t B, uint32_t in_dim, uint32_t out_dim) {
    return (uint64_t)wgrad_blocks(B) * in_dim * out_dim * 4;
}

int pnr_linear_wgrad(const void* x, int x_dtype, const void* dy, int dy_dtype, uint32_t B, uint32_t in_dim, uint32_t out_dim, float* dw,
                     int accumulate, void* workspace, uint64_t workspace_bytes, pnr_stream_t stream) {
    if (in_dim == 0 || out_dim == 0 || in_dim > 64 || out_dim > 64) return PNR_ERR_UNSUPPORTED;
    if ((x_dtype != PNR_DTYPE_F32 && x_dtype != PNR_DTYPE_F16) || (dy_dtype != PNR_DTYPE_F32 && dy_dtype != PNR_DTYPE_F16)) return PNR_ERR_UNSUPPORTED;
    if (!dw) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    const uint32_t n = in_dim * out_dim;
    if (B == 0) {
        if (!accumulate && hipMemsetAsync(dw, 0, (size_t)n * 4, s) != hipSuccess) return PNR_ERR_LAUNCH;
        return PNR_OK;
    }
    if ((!x && in_dim != 1) || !dy || !workspace) return PNR_ERR_INVALID;   // x == NULL with in_dim 1: a column of ones (pnr_linear_bgrad)
    if (workspace_bytes < pnr_linear_wgrad_workspace_bytes(B, in_dim, out_dim)) return PNR_ERR_INVALID;
    const uint32_t blocks = wgrad_blocks(B);
    float* partial = static_cast<float*>(workspace);
    const int ti = in_dim > 32 ? 2 : 1, to = out_dim > 32 ? 2 : 1;
#define PNR_WG(TX, TY, TIV, TOV)                                                                                                  \
    hipLaunchKernelGGL((k_linear_wgrad<TX, TY, TIV, TOV>), dim3(blocks), dim3(kLinThreads), 0, s, static_cast<const TX*>(x),       \
                       static_cast<const TY*>(dy), B, in_dim, out_dim, partial)
#define PNR_WG_T(TX, TY)                                                        \
    do {                                                                        \
        if (ti == 1 && to == 1) PNR_WG(TX, TY, 1, 1);                           \
        else if (ti == 2 && to == 1) PNR_WG(TX, TY, 2, 1);                      \
        else if (ti == 1 && to == 2) PNR_WG(TX, TY, 1, 2);                      \
        else PNR_WG(TX, TY, 2, 2);                                              \
    } while (0)
    if (x_dtype == PNR_DTYPE_F32 && dy_dtype == PNR_DTYPE_F32) PNR_WG_T(float, float);
    else if (x_dtype == PNR_DTYPE_F16 && dy_dtype == PNR_DTYPE_F16) PNR_WG_T(__half, __half);
    else if (x_dtype == PNR_DTYPE_F32) PNR_WG_T(float, __half);
    else PNR_WG_T(__half, float);
#undef PNR_WG_T
#undef PNR_WG
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3(cdiv(n, 32)), dim3(256), 0, s, partial, blocks, n, dw, accumulate);
    return check_launch();
}

/* bias gradient db[o] = sum_b dY[b][o]: the same kernel with X = a column of ones (autograd's column reduction of a [6e5, 13] gradient
 * takes 0.8 ms on MI355X; this takes the time of reading dY once) */
int pnr_linear_bgrad(const void* dy, int dy_dtype, uint32_t B, uint32_t out_dim, float* db, int accumulate, void* workspace, uint64_t workspace_bytes,
                     pnr_stream_t stream) {
    return pnr_linear_wgrad(nullptr, PNR_DTYPE_F32, dy, dy_dtype, B, 1, out_dim, db, accumulate, workspace, workspace_bytes, stream);
}

}  // extern "C"
