// train_loss.hip -- the ray-level tail of a training step for gfx950: the renderer's epilogue (palette/renderer.py:387-403; background blend
// of image and direct_rgb, depth normalisation) and the trainer's loss on it (palette/utils.py:483-600, MSE criterion; nerf/utils.py:534-556
// when there is no all_map) as ONE launch forward and ONE backward.
//
// In torch this block is ~25 elementwise / reduction launches over 4096-element tensors forward and ~20 backward (slice backward = fill +
// copy + accumulate per term), each a few microseconds of launch latency for a few kilobytes of work: a fifth of the launches of a step.
// Forward: one thread per ray forms the blended colours, the depth and the ray's contribution to the eight sums; the sums go wave shuffle ->
// LDS -> one partial row per workgroup, and the LAST workgroup to finish (an agent-scope acq_rel ticket) adds the partial rows in a fixed
// order and writes the ten loss terms -- reproducible, no second launch.  Backward: one thread per gradient element; every element of
// grad_weights_sum / grad_image_raw / grad_all_map is written (no zero fill by the caller), scaled by the incoming device scalar.
// Latency-bound by construction (N = 4096 rays: 0.6 MB in, 16 workgroups); the point is the launches it replaces.
#include "pnr_common.hpp"

namespace pnr {

constexpr int kLossSums = 8;      // mse, sparsity, offsets, view_dep, smooth, weight, direct, clip
constexpr uint32_t kLossHdr = 64;  // bytes in front of the partial rows: the ticket

__device__ __forceinline__ void loss_bg(const pnr_train_loss_args& a, uint32_t n, float bg[3]) {
    if (a.bg_mode == 0) { bg[0] = bg[1] = bg[2] = a.bg_const; }
    else if (a.bg_mode == 1) { bg[0] = a.bg_color[0]; bg[1] = a.bg_color[1]; bg[2] = a.bg_color[2]; }
    else { bg[0] = a.bg_color[(size_t)n * 3]; bg[1] = a.bg_color[(size_t)n * 3 + 1]; bg[2] = a.bg_color[(size_t)n * 3 + 2]; }
}

__global__ void __launch_bounds__(256) k_train_loss_fwd(pnr_train_loss_args a) {
    __shared__ float red[4][kLossSums];
    __shared__ float tot[kLossSums];
    __shared__ uint32_t is_last;
    uint32_t* ticket = reinterpret_cast<uint32_t*>(a.workspace);
    float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(a.workspace) + kLossHdr);
    const uint32_t n = blockIdx.x * 256 + threadIdx.x, C = a.n_channel, nb = a.num_basis, clip = a.clip_dim;
    float s[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; k++) s[k] = 0.0f;
    if (n < a.N) {
        const float om = 1.0f - a.weights_sum[n];
        float bg[3];
        loss_bg(a, n, bg);
        const float* row = a.all_map ? a.all_map + (size_t)n * C : nullptr;
        float e = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float gt = a.gt_rgb[(size_t)n * 3 + c];
            const float img = a.image_raw[(size_t)n * 3 + c] + om * bg[c];
            if (a.image) a.image[(size_t)n * 3 + c] = img;
            const float d = img - gt;
            e += d * d;
            if (row) {
                const float dir = row[7 + c] + om * bg[c];
                if (a.direct_rgb) a.direct_rgb[(size_t)n * 3 + c] = dir;
                const float dd = dir - gt;
                s[6] += dd * dd;
            }
        }
        e *= 1.0f / 3.0f;
        s[0] = e;
        if (a.loss_ray) a.loss_ray[n] = e;
        if (row) {
            s[1] = row[0]; s[2] = row[2]; s[3] = row[1]; s[4] = row[3];
            if (a.gt_weights)
                for (uint32_t b = 0; b < nb; b++) { const float d = a.gt_weights[(size_t)n * nb + b] - row[13 + clip + b]; s[5] += d * d; }
            if (a.gt_clip)
                for (uint32_t j = 0; j < clip; j++) { const float d = row[13 + j] - a.gt_clip[(size_t)n * clip + j]; s[7] += d * d; }
        }
        if (a.depth) a.depth[n] = fmaxf(a.depth_raw[n] - a.nears[n], 0.0f) / (a.fars[n] - a.nears[n]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kLossSums; k++) {
        float v = s[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kLossSums) {
        partials[(size_t)blockIdx.x * kLossSums + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        __threadfence();   // the row is out at agent scope before this workgroup takes its ticket
    }
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!is_last) return;
    __threadfence();       // every lane of the last workgroup reads the other workgroups' rows behind the ticket
    {
        const uint32_t k = threadIdx.x >> 5, l = threadIdx.x & 31;   // 32 lanes per sum, rows l, l + 32, ... then a butterfly: a fixed order
        float v = 0.0f;
        for (uint32_t b = l; b < gridDim.x; b += 32) v += partials[(size_t)b * kLossSums + k];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (l == 0) tot[k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_n = 1.0f / (float)a.N;
        float t[PNR_TRAIN_LOSS_TERMS];
        t[1] = tot[0] * inv_n;
        t[2] = a.lambda_sparsity * (tot[1] * inv_n);
        t[3] = a.lambda_offsets * (tot[2] * inv_n);
        t[4] = a.lambda_view_dep * (tot[3] * inv_n);
        t[5] = a.lambda_smooth * (tot[4] * inv_n);
        float pal = 0.0f;
        if (a.basis_color && a.basis_color_origin && nb > 0) {
            for (uint32_t i = 0; i < nb * 3; i++) { const float d = a.basis_color[i] - a.basis_color_origin[i]; pal += d * d; }
            pal = a.lambda_palette * (pal / (float)nb);
        }
        t[6] = pal;
        t[7] = a.gt_weights && nb > 0 ? a.lambda_weight * (tot[5] / ((float)a.N * (float)nb)) : 0.0f;
        t[8] = a.all_map ? tot[6] / ((float)a.N * 3.0f) : 0.0f;
        t[9] = a.gt_clip && clip > 0 ? tot[7] / ((float)a.N * (float)clip) : 0.0f;
        float loss = t[1];
        if (a.all_map)
            for (int i = 2; i < PNR_TRAIN_LOSS_TERMS; i++) loss += t[i];   // the order palette/utils.py:551-577 adds them in
        else
            for (int i = 2; i < PNR_TRAIN_LOSS_TERMS; i++) t[i] = 0.0f;
        t[0] = loss;
        for (int i = 0; i < PNR_TRAIN_LOSS_TERMS; i++) a.terms[i] = t[i];
        *ticket = 0u;      // ready for the next launch on this workspace (kernel boundaries order it)
    }
}

__global__ void __launch_bounds__(256) k_train_loss_bwd(pnr_train_loss_args a) {
    const uint32_t C = a.n_channel, W = C + 4, nb = a.num_basis, clip = a.clip_dim;
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    const float g = a.grad_loss[0];
    if (blockIdx.x == 0 && a.grad_basis_color && threadIdx.x < nb * 3) {
        const float d = a.basis_color && a.basis_color_origin ? a.basis_color[threadIdx.x] - a.basis_color_origin[threadIdx.x] : 0.0f;
        a.grad_basis_color[threadIdx.x] = a.lambda_palette * (2.0f * d / (float)nb) * g;
    }
    if (idx >= a.N * W) return;
    const uint32_t n = idx / W, c = idx - n * W;
    const float inv_n = 1.0f / (float)a.N, inv_3n = 1.0f / ((float)a.N * 3.0f);
    if (c < C) {
        const float x = a.all_map[(size_t)n * C + c];
        float v = 0.0f;
        if (c == 0) v = a.lambda_sparsity * inv_n;
        else if (c == 1) v = a.lambda_view_dep * inv_n;
        else if (c == 2) v = a.lambda_offsets * inv_n;
        else if (c == 3) v = a.lambda_smooth * inv_n;
        else if (c >= 7 && c < 10) {
            float bg[3];
            loss_bg(a, n, bg);
            const float dir = x + (1.0f - a.weights_sum[n]) * bg[c - 7];
            v = 2.0f * (dir - a.gt_rgb[(size_t)n * 3 + (c - 7)]) * inv_3n;
        } else if (c >= 13 && c < 13 + clip) {
            if (a.gt_clip) v = 2.0f * (x - a.gt_clip[(size_t)n * clip + (c - 13)]) / ((float)a.N * (float)clip);
        } else if (c >= 13 + clip) {
            if (a.gt_weights) v = a.lambda_weight * (2.0f * (x - a.gt_weights[(size_t)n * nb + (c - 13 - clip)]) / ((float)a.N * (float)nb));
        }
        a.grad_all_map[(size_t)n * C + c] = v * g;
        return;
    }
    float bg[3];
    loss_bg(a, n, bg);
    const float om = 1.0f - a.weights_sum[n];
    if (c < C + 3) {
        const uint32_t k = c - C;
        const float img = a.image_raw[(size_t)n * 3 + k] + om * bg[k];
        a.grad_image_raw[(size_t)n * 3 + k] = 2.0f * (img - a.gt_rgb[(size_t)n * 3 + k]) * inv_3n * g;
        return;
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float gt = a.gt_rgb[(size_t)n * 3 + k];
        float gk = 2.0f * ((a.image_raw[(size_t)n * 3 + k] + om * bg[k]) - gt) * inv_3n;
        if (a.all_map) gk += 2.0f * ((a.all_map[(size_t)n * C + 7 + k] + om * bg[k]) - gt) * inv_3n;
        acc += bg[k] * gk;
    }
    a.grad_weights_sum[n] = -acc * g;
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_train_loss_workspace_bytes(uint32_t N) { return kLossHdr + (uint64_t)cdiv(N ? N : 1u, 256) * kLossSums * sizeof(float); }

static int train_loss_check(const pnr_train_loss_args* a) {
    if (!a) return PNR_ERR_INVALID;
    if (!a->weights_sum || !a->image_raw || !a->gt_rgb) return PNR_ERR_INVALID;
    if (a->bg_mode < 0 || a->bg_mode > 2 || (a->bg_mode != 0 && !a->bg_color)) return PNR_ERR_INVALID;
    if (a->all_map) {
        if (a->n_channel != 13 + a->clip_dim + a->num_basis || a->n_channel > PNR_CHANNEL_MAXIMUM || a->num_basis > PNR_MAX_BASIS) return PNR_ERR_INVALID;
    } else if (a->n_channel != 0 || a->gt_clip || a->gt_weights) {
        return PNR_ERR_INVALID;
    }
    if ((uint64_t)a->N * (a->n_channel + 4) >= (1ull << 32)) return PNR_ERR_UNSUPPORTED;
    return PNR_OK;
}

int pnr_train_loss_forward(const pnr_train_loss_args* a, pnr_stream_t stream) {
    if (const int rc = train_loss_check(a)) return rc;
    if (a->N == 0 || !a->terms || !a->workspace || a->workspace_bytes < pnr_train_loss_workspace_bytes(a->N)) return PNR_ERR_INVALID;
    if (a->depth && (!a->depth_raw || !a->nears || !a->fars)) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_train_loss_fwd, dim3(cdiv(a->N, 256)), dim3(256), 0, as_stream(stream), *a);
    return check_launch();
}

int pnr_train_loss_backward(const pnr_train_loss_args* a, pnr_stream_t stream) {
    if (const int rc = train_loss_check(a)) return rc;
    if (a->N == 0) return PNR_OK;
    if (!a->grad_loss || !a->grad_weights_sum || !a->grad_image_raw || (a->all_map && !a->grad_all_map)) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_train_loss_bwd, dim3(cdiv(a->N * (a->n_channel + 4), 256)), dim3(256), 0, as_stream(stream), *a);
    return check_launch();
}

}  // extern "C"
