// palette_train.hip -- training-mode palette colour-basis composite for gfx950 (palette/renderer.py:344-386).
//
// Between the field's heads and the two ray composites the reference runs ~40 elementwise / reduction / concatenation launches per
// step over M ~ 6e5 samples (softplus, clamp, broadcasts over [M, nb, 3], five row sums, a 33-column torch.cat) and as many again in
// the backward.  Here the whole block is one kernel each way, one thread per sample:
//   sp            = softplus(radiance)                                   radiance = offsets_radiance[:, 3 nb]
//   rgbs[c]       = sum_b omega_b * sp * (clamp(P_b[c], 0, 1) + offsets[b][c])  +  view_dep[c]      (view_dep detached on this path)
//   all_buffer    = [omega_sparsity, view_dep_norm, offsets_norm, smooth_norm, view_dep, diffuse + view_dep, diffuse, clip_feat, omega]
//   omega_sparsity = sum omega / (sum omega^2 + 1e-6) - 1,  offsets_norm = sum offsets^2,  view_dep_norm = sum view_dep^2
// The backward recomputes the few forward values it needs and reduces the gradient of the nb x 3 basis colours deterministically:
// wave shuffle -> LDS -> one partial row per workgroup -> a second tiny launch sums the partials in a fixed order.
// HBM-bound: (4 nb + 8 + clip) * 4 B read and (16 + clip + nb) * 4 B written per sample forward.
#include "pnr_common.hpp"

namespace pnr {

constexpr uint32_t kShadeBlocks = 512;    // workgroups of the backward (= rows of the basis-colour partials)
constexpr uint32_t kShadeMaxBasis = 16;

__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }       // F.softplus, beta 1, threshold 20
__device__ __forceinline__ float softplus_grad(float x) { if (x > 20.0f) return 1.0f; const float z = expf(x); return z / (z + 1.0f); }

// The 13 + clip + nb columns of a row are written / read through an LDS tile (256 rows x (13 + nb) columns, odd stride) so that a
// wave touches consecutive addresses instead of 64 rows 132 bytes apart; the clip columns are a plain strided copy done cooperatively.
constexpr uint32_t kShadeTileCols = 13 + kShadeMaxBasis;          // the non-clip columns
constexpr uint32_t kShadeTileStride = kShadeTileCols | 1u;

__global__ void __launch_bounds__(256) k_palette_train_shade_fwd(uint32_t M, uint32_t nb, uint32_t clip, const float* __restrict__ omega,
                                                                 const float* __restrict__ offrad, const float* __restrict__ view_dep,
                                                                 const float* __restrict__ diffuse, const float* __restrict__ clip_feat,
                                                                 const float* __restrict__ smooth, const float* __restrict__ basis_color,
                                                                 float* __restrict__ rgbs, float* __restrict__ all_buffer) {
    __shared__ float bc[kShadeMaxBasis * 3];
    __shared__ float tile[256 * kShadeTileStride];
    if (threadIdx.x < nb * 3) bc[threadIdx.x] = fminf(fmaxf(basis_color[threadIdx.x], 0.0f), 1.0f);
    __syncthreads();
    const uint32_t ch = 13 + clip + nb, orw = 3 * nb + 1, w = 13 + nb, ws = w | 1u;
    const uint32_t ntiles = (M + 255) / 256;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t row0 = t * 256, i = row0 + threadIdx.x, nrows = M - row0 < 256u ? M - row0 : 256u;
        if (i < M) {
            const float* orow = offrad + (size_t)i * orw;
            const float sp = softplus_t(orow[3 * nb]);
            float s1 = 0.0f, s2 = 0.0f, on = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f};
            float* out = tile + threadIdx.x * ws;
            for (uint32_t b = 0; b < nb; b++) {
                const float wgt = omega[(size_t)i * nb + b];
                s1 += wgt;
                s2 += wgt * wgt;
                float onb = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float o = orow[b * 3 + c];
                    onb += o * o;
                    rgb[c] += wgt * (sp * (bc[b * 3 + c] + o));
                }
                on += onb;
                out[13 + b] = wgt;
            }
            float vd[3], df[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { vd[c] = view_dep[(size_t)i * 3 + c]; df[c] = diffuse[(size_t)i * 3 + c]; }
#pragma unroll
            for (int c = 0; c < 3; c++) rgbs[(size_t)i * 3 + c] = rgb[c] + vd[c];
            out[0] = s1 / (s2 + 1e-6f) - 1.0f;
            out[1] = vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2];
            out[2] = on;
            out[3] = smooth ? smooth[i] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) { out[4 + c] = vd[c]; out[7 + c] = df[c] + vd[c]; out[10 + c] = df[c]; }
        }
        __syncthreads();
        float* dst = all_buffer + (size_t)row0 * ch;
        for (uint32_t f = threadIdx.x; f < nrows * w; f += 256) {
            const uint32_t r = f / w, c = f - r * w;
            dst[(size_t)r * ch + (c < 13 ? c : c + clip)] = tile[r * ws + c];
        }
        for (uint32_t f = threadIdx.x; f < nrows * clip; f += 256) {
            const uint32_t r = f / clip, c = f - r * clip;
            dst[(size_t)r * ch + 13 + c] = clip_feat ? clip_feat[(size_t)row0 * clip + f] : 0.0f;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_palette_train_shade_bwd(uint32_t M, uint32_t nb, uint32_t clip, const float* __restrict__ omega,
                                                                 const float* __restrict__ offrad, const float* __restrict__ view_dep,
                                                                 const float* __restrict__ basis_color, const float* __restrict__ g_rgbs,
                                                                 const float* __restrict__ g_all, float* __restrict__ g_omega,
                                                                 float* __restrict__ g_offrad, float* __restrict__ g_view_dep,
                                                                 float* __restrict__ g_diffuse, float* __restrict__ g_clip,
                                                                 float* __restrict__ g_smooth, float* __restrict__ bc_partial /* [gridDim.x][nb*3] or null */) {
    __shared__ float bc[kShadeMaxBasis * 3];
    __shared__ float bc_pass[kShadeMaxBasis * 3];   // 1 where the clamp lets the gradient through (0 <= P <= 1)
    __shared__ float red[4][kShadeMaxBasis * 3];
    __shared__ float tile[256 * kShadeTileStride];
    if (threadIdx.x < nb * 3) {
        const float p = basis_color[threadIdx.x];
        bc[threadIdx.x] = fminf(fmaxf(p, 0.0f), 1.0f);
        bc_pass[threadIdx.x] = (p >= 0.0f && p <= 1.0f) ? 1.0f : 0.0f;
    }
    __syncthreads();
    const uint32_t ch = 13 + clip + nb, orw = 3 * nb + 1, w = 13 + nb, ws = w | 1u;
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    float gbc[kShadeMaxBasis * 3];   // this thread's share of d loss / d clamp(P); fully unrolled below so that it lives in registers
#pragma unroll
    for (int k = 0; k < (int)kShadeMaxBasis * 3; k++) gbc[k] = 0.0f;
    const uint32_t ntiles = (M + 255) / 256;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t row0 = t * 256, i = row0 + threadIdx.x, nrows = M - row0 < 256u ? M - row0 : 256u;
        const float* src = g_all + (size_t)row0 * ch;
        for (uint32_t f = threadIdx.x; f < nrows * w; f += 256) {
            const uint32_t r = f / w, c = f - r * w;
            tile[r * ws + c] = src[(size_t)r * ch + (c < 13 ? c : c + clip)];
        }
        if (g_clip)
            for (uint32_t f = threadIdx.x; f < nrows * clip; f += 256) {
                const uint32_t r = f / clip, c = f - r * clip;
                g_clip[(size_t)row0 * clip + f] = src[(size_t)r * ch + 13 + c];
            }
        __syncthreads();
        if (i < M) {
            const float* orow = offrad + (size_t)i * orw;
            const float* ga = tile + threadIdx.x * ws;
            const float rad = orow[3 * nb];
            const float sp = softplus_t(rad);
            float gr[3];
#pragma unroll
            for (int c = 0; c < 3; c++) gr[c] = g_rgbs[(size_t)i * 3 + c];
            float s1 = 0.0f, s2 = 0.0f;
            for (uint32_t b = 0; b < nb; b++) { const float wgt = omega[(size_t)i * nb + b]; s1 += wgt; s2 += wgt * wgt; }
            const float den = s2 + 1e-6f, g_os = ga[0], g_vdn = ga[1], g_on = ga[2];
            float g_rad = 0.0f;
#pragma unroll
            for (uint32_t b = 0; b < kShadeMaxBasis; b++) {
                if (b < nb) {
                    const float wgt = omega[(size_t)i * nb + b];
                    float gw = ga[13 + b] + g_os * (1.0f / den - s1 * 2.0f * wgt / (den * den));
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float o = orow[b * 3 + c];
                        const float base = bc[b * 3 + c] + o;
                        gw += gr[c] * (sp * base);
                        const float gfc = gr[c] * wgt;          // d loss / d final_color[b][c]
                        g_rad += gfc * base;
                        g_offrad[(size_t)i * orw + b * 3 + c] = gfc * sp + g_on * 2.0f * o;
                        gbc[b * 3 + c] += gfc * sp;
                    }
                    g_omega[(size_t)i * nb + b] = gw;
                }
            }
            g_offrad[(size_t)i * orw + 3 * nb] = g_rad * softplus_grad(rad);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float vd = view_dep[(size_t)i * 3 + c];
                g_view_dep[(size_t)i * 3 + c] = ga[4 + c] + ga[7 + c] + g_vdn * 2.0f * vd;   // rgbs uses view_dep.detach()
                g_diffuse[(size_t)i * 3 + c] = ga[7 + c] + ga[10 + c];
            }
            if (g_smooth) g_smooth[i] = ga[3];
        }
        __syncthreads();
    }
    if (!bc_partial) return;
#pragma unroll
    for (uint32_t k = 0; k < kShadeMaxBasis * 3; k++) {
        if (k < nb * 3) {
            float v = gbc[k];
#pragma unroll
            for (int off = PNR_WAVE / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) red[wave][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < nb * 3)
        bc_partial[(size_t)blockIdx.x * nb * 3 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * bc_pass[threadIdx.x];
}

// out[e] = sum of the workgroup partials in a fixed order: 16 groups of 64 threads, group g sums partials g, g + 16, ...; then the 16 group sums
__global__ void __launch_bounds__(1024) k_palette_train_shade_reduce(const float* __restrict__ partial, uint32_t nparts, uint32_t n, float* __restrict__ out) {
    __shared__ float red[16][64];
    const uint32_t e = threadIdx.x & 63u, g = threadIdx.x >> 6;
    float s = 0.0f;
    if (e < n)
        for (uint32_t p = g; p < nparts; p += 16) s += partial[(size_t)p * n + e];
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && e < n) {
        float tsum = red[0][e];
#pragma unroll
        for (int k = 1; k < 16; k++) tsum += red[k][e];
        out[e] = tsum;
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_palette_train_shade_workspace_bytes(uint32_t num_basis) { return (uint64_t)kShadeBlocks * num_basis * 3 * 4; }

int pnr_palette_train_shade_forward(uint32_t M, uint32_t num_basis, uint32_t clip_dim, const float* omega, const float* offsets_radiance,
                                    const float* view_dep, const float* diffuse, const float* clip_feat, const float* smooth_norm,
                                    const float* basis_color, float* rgbs, float* all_buffer, pnr_stream_t stream) {
    if (num_basis == 0 || num_basis > kShadeMaxBasis) return PNR_ERR_UNSUPPORTED;
    if (M == 0) return PNR_OK;
    if (!omega || !offsets_radiance || !view_dep || !diffuse || !basis_color || !rgbs || !all_buffer) return PNR_ERR_INVALID;
    const uint32_t blocks = cdiv(M, 256);
    hipLaunchKernelGGL(k_palette_train_shade_fwd, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, as_stream(stream), M, num_basis, clip_dim,
                       omega, offsets_radiance, view_dep, diffuse, clip_feat, smooth_norm, basis_color, rgbs, all_buffer);
    return check_launch();
}

int pnr_palette_train_shade_backward(uint32_t M, uint32_t num_basis, uint32_t clip_dim, const float* omega, const float* offsets_radiance,
                                     const float* view_dep, const float* basis_color, const float* grad_rgbs, const float* grad_all,
                                     float* grad_omega, float* grad_offsets_radiance, float* grad_view_dep, float* grad_diffuse,
                                     float* grad_clip_feat, float* grad_smooth_norm, float* grad_basis_color, void* workspace,
                                     uint64_t workspace_bytes, pnr_stream_t stream) {
    if (num_basis == 0 || num_basis > kShadeMaxBasis) return PNR_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    if (M == 0) {
        if (grad_basis_color && hipMemsetAsync(grad_basis_color, 0, (size_t)num_basis * 3 * 4, s) != hipSuccess) return PNR_ERR_LAUNCH;
        return PNR_OK;
    }
    if (!omega || !offsets_radiance || !view_dep || !basis_color || !grad_rgbs || !grad_all || !grad_omega || !grad_offsets_radiance || !grad_view_dep ||
        !grad_diffuse)
        return PNR_ERR_INVALID;
    if (grad_basis_color && (!workspace || workspace_bytes < pnr_palette_train_shade_workspace_bytes(num_basis))) return PNR_ERR_INVALID;
    const uint32_t want = cdiv(M, 256), blocks = want < kShadeBlocks ? want : kShadeBlocks;
    float* partial = grad_basis_color ? static_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(k_palette_train_shade_bwd, dim3(blocks), dim3(256), 0, s, M, num_basis, clip_dim, omega, offsets_radiance, view_dep, basis_color,
                       grad_rgbs, grad_all, grad_omega, grad_offsets_radiance, grad_view_dep, grad_diffuse, grad_clip_feat, grad_smooth_norm, partial);
    if (grad_basis_color)
        hipLaunchKernelGGL(k_palette_train_shade_reduce, dim3(1), dim3(1024), 0, s, partial, blocks, num_basis * 3, grad_basis_color);
    return check_launch();
}

}  // extern "C"
