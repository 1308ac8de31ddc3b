// composite.hip -- volume compositing along rays for gfx950 (MI355X).
//
// One thread integrates one ray (the recurrence is inherently serial per ray); what differs from
// the reference's kernels is the handling of the generic-channel ("flex") variants: instead of a
// 128-entry per-thread local array (which lives in scratch memory), channels are processed in
// register-resident chunks of kChunk, re-running the cheap transmittance recurrence per chunk --
// identical weights, identical per-channel results, no scratch traffic.
#include "pnr_common.hpp"

namespace pnr {

constexpr uint32_t kBlock = 256;
constexpr int kChunk = 16;

// alpha of one sample; __expf == v_exp_f32(x * log2e), the gfx950 counterpart of CUDA's __expf
__device__ __forceinline__ float alpha_of(float sigma, float delta) { return 1.0f - __expf(-sigma * delta); }

// reference raymarching.cu:504-580
__global__ void __launch_bounds__(kBlock) k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                uint32_t M, uint32_t N, float T_thresh, float* __restrict__ weights_sum,
                                                                float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        weights_sum[index] = 0; depth[index] = 0;
        image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
        return;
    }
    const float* s = sigmas + offset;
    const float* c = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float alpha = alpha_of(s[0], dl[0]);
        const float w = alpha * T;
        r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b);
        t += dl[1];
        d = fmaf(w, t, d);
        ws += w;
        T *= 1.0f - alpha;
        if (T < T_thresh) break;
        s++; c += 3; dl += 2;
    }
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// reference raymarching.cu:583-645
__global__ void __launch_bounds__(kBlock) k_composite_flex_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ input,
                                                                     const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                     uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh,
                                                                     float* __restrict__ output) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float* out = output + (size_t)index * n_channel;
    if (num_steps == 0 || offset + num_steps >= M) {  // '>=' here, '>' in the rgb variant (reference quirk)
        for (uint32_t i = 0; i < n_channel; i++) out[i] = 0;
        return;
    }
    for (uint32_t c0 = 0; c0 < n_channel; c0 += kChunk) {
        const int nc = (int)min((uint32_t)kChunk, n_channel - c0);
        const float* s = sigmas + offset;
        const float* in = input + (size_t)offset * n_channel + c0;
        const float* dl = deltas + (size_t)offset * 2;
        float acc[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; i++) acc[i] = 0;
        float T = 1.0f;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = alpha_of(s[0], dl[0]);
            const float w = alpha * T;
#pragma unroll
            for (int i = 0; i < kChunk; i++) if (i < nc) acc[i] = fmaf(w, in[i], acc[i]);
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; in += n_channel; dl += 2;
        }
#pragma unroll
        for (int i = 0; i < kChunk; i++) if (i < nc) out[c0 + i] = acc[i];
    }
}

// Same arithmetic with 16 lanes per ray: a training batch has only a few thousand rays, one thread per ray leaves the chip idle
// and walks a 4 n_channel-byte-strided column.  Every lane re-runs the cheap transmittance recurrence (identical weights,
// identical order) and owns channels q, q + 16, q + 32, q + 48 of a 64-channel pass: the 16 lanes of a ray read 64 contiguous
// bytes per load; four samples are fetched ahead of the serial recurrence.
constexpr uint32_t kLanesPerRay = 16;
constexpr int kCoopCh = 4;   // channels per lane and pass
__global__ void __launch_bounds__(kBlock) k_composite_flex_train_fwd_coop(const float* __restrict__ sigmas, const float* __restrict__ input,
                                                                          const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                          uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh,
                                                                          float* __restrict__ output) {
    const uint32_t n = (blockIdx.x * kBlock + threadIdx.x) / kLanesPerRay, q = threadIdx.x % kLanesPerRay;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    const bool empty = num_steps == 0 || offset + num_steps >= M;   // '>=' here, '>' in the rgb variant (reference quirk)
    for (uint32_t c0 = q; c0 < n_channel; c0 += kLanesPerRay * kCoopCh) {
        float acc[kCoopCh] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!empty) {
            const float* s = sigmas + offset;
            const float* dl = deltas + (size_t)offset * 2;
            const float* in = input + (size_t)offset * n_channel + c0;
            float T = 1.0f;
            bool stop = false;
            for (uint32_t base = 0; base < num_steps && !stop; base += 4) {
                float sg[4], dt[4], v[4][kCoopCh];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t k = base + u < num_steps ? base + u : num_steps - 1;   // clamped: stays inside the ray's rows
                    sg[u] = s[k]; dt[u] = dl[(size_t)k * 2];
#pragma unroll
                    for (int j = 0; j < kCoopCh; j++) v[u][j] = c0 + j * kLanesPerRay < n_channel ? in[(size_t)k * n_channel + j * kLanesPerRay] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (stop || base + u >= num_steps) break;
                    const float alpha = alpha_of(sg[u], dt[u]);
                    const float w = alpha * T;
#pragma unroll
                    for (int j = 0; j < kCoopCh; j++) acc[j] = fmaf(w, v[u][j], acc[j]);
                    T *= 1.0f - alpha;
                    if (T < T_thresh) stop = true;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kCoopCh; j++) if (c0 + j * kLanesPerRay < n_channel) output[(size_t)index * n_channel + c0 + j * kLanesPerRay] = acc[j];
    }
}

// reference raymarching.cu:764-819, 16 lanes per ray as above (rows past the terminating sample keep the caller's zeros)
__global__ void __launch_bounds__(kBlock) k_composite_flex_train_bwd_coop(const float* __restrict__ grad_output, const float* __restrict__ sigmas,
                                                                          const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                          uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh,
                                                                          float* __restrict__ grad_input) {
    const uint32_t n = (blockIdx.x * kBlock + threadIdx.x) / kLanesPerRay, q = threadIdx.x % kLanesPerRay;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    for (uint32_t c0 = q; c0 < n_channel; c0 += kLanesPerRay * kCoopCh) {
        float go[kCoopCh];
#pragma unroll
        for (int j = 0; j < kCoopCh; j++) go[j] = c0 + j * kLanesPerRay < n_channel ? grad_output[(size_t)index * n_channel + c0 + j * kLanesPerRay] : 0.0f;
        const float* s = sigmas + offset;
        const float* dl = deltas + (size_t)offset * 2;
        float* gin = grad_input + (size_t)offset * n_channel + c0;
        float T = 1.0f;
        bool stop = false;
        for (uint32_t base = 0; base < num_steps && !stop; base += 4) {
            float sg[4], dt[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = base + u < num_steps ? base + u : num_steps - 1;
                sg[u] = s[k]; dt[u] = dl[(size_t)k * 2];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (stop || base + u >= num_steps) break;
                const float alpha = alpha_of(sg[u], dt[u]);
                const float w = alpha * T;
                T *= 1.0f - alpha;
                if (T < T_thresh) { stop = true; break; }   // break BEFORE the write: the breaking sample gets no gradient (reference quirk)
#pragma unroll
                for (int j = 0; j < kCoopCh; j++)
                    if (c0 + j * kLanesPerRay < n_channel) gin[(size_t)(base + u) * n_channel + j * kLanesPerRay] = go[j] * w;
            }
        }
    }
}

// reference raymarching.cu:681-761
__global__ void __launch_bounds__(kBlock) k_composite_train_bwd(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                uint32_t M, uint32_t N, float T_thresh, float* __restrict__ grad_sigmas,
                                                                float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad_image[index * 3], g1 = grad_image[index * 3 + 1], g2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2], ws_final = weights_sum[index];
    const float* s = sigmas + offset;
    const float* c = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + (size_t)offset * 3;
    float T = 1.0f, r = 0, g = 0, b = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float alpha = alpha_of(s[0], dl[0]);
        const float w = alpha * T;
        r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b);
        T *= 1.0f - alpha;
        gc[0] = g0 * w; gc[1] = g1 * w; gc[2] = g2 * w;
        float acc = g0 * fmaf(T, c[0], -(r_final - r));
        acc = fmaf(g1, fmaf(T, c[1], -(g_final - g)), acc);
        acc = fmaf(g2, fmaf(T, c[2], -(b_final - b)), acc);
        acc = fmaf(gws, 1.0f - ws_final, acc);
        gs[0] = dl[0] * acc;
        if (T < T_thresh) break;
        s++; c += 3; dl += 2; gs++; gc += 3;
    }
}

// reference raymarching.cu:764-819 ; the breaking sample gets no gradient (reference quirk)
__global__ void __launch_bounds__(kBlock) k_composite_flex_train_bwd(const float* __restrict__ grad_output, const float* __restrict__ sigmas,
                                                                     const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                     uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh,
                                                                     float* __restrict__ grad_input) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float* go = grad_output + (size_t)index * n_channel;
    const float* s = sigmas + offset;
    const float* dl = deltas + (size_t)offset * 2;
    float* gin = grad_input + (size_t)offset * n_channel;
    float T = 1.0f;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float alpha = alpha_of(s[0], dl[0]);
        const float w = alpha * T;
        T *= 1.0f - alpha;
        if (T < T_thresh) break;
        for (uint32_t i = 0; i < n_channel; i++) gin[i] = go[i] * w;
        s++; dl += 2; gin += n_channel;
    }
}

// reference raymarching.cu:1025-1111
__global__ void __launch_bounds__(kBlock) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* __restrict__ rays_alive,
                                                           float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                           const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                           float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float* s = sigmas + (size_t)n * n_step;
    const float* c = rgbs + (size_t)n * n_step * 3;
    const float* dl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index], ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = alpha_of(s[0], dl[0]);
        const float T = 1.0f - ws;
        const float w = alpha * T;
        ws += w;
        t += dl[1];
        d = fmaf(w, t, d);
        r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b);
        if (T < T_thresh) break;
        s++; c += 3; dl += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// reference raymarching.cu:1114-1185
__global__ void __launch_bounds__(kBlock) k_composite_rays_flex(uint32_t n_alive, uint32_t n_step, uint32_t n_channel, float T_thresh,
                                                                const int32_t* __restrict__ rays_alive, const float* __restrict__ sigmas,
                                                                const float* __restrict__ input, const float* __restrict__ deltas,
                                                                const float* __restrict__ weights_sum, float* __restrict__ output) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float* out = output + (size_t)index * n_channel;
    const float ws0 = weights_sum[index];
    for (uint32_t c0 = 0; c0 < n_channel; c0 += kChunk) {
        const int nc = (int)min((uint32_t)kChunk, n_channel - c0);
        const float* s = sigmas + (size_t)n * n_step;
        const float* in = input + (size_t)n * n_step * n_channel + c0;
        const float* dl = deltas + (size_t)n * n_step * 2;
        float acc[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; i++) acc[i] = (i < nc) ? out[c0 + i] : 0.0f;
        float ws = ws0;
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = alpha_of(s[0], dl[0]);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
#pragma unroll
            for (int i = 0; i < kChunk; i++) if (i < nc) acc[i] = fmaf(w, in[i], acc[i]);
            if (T < T_thresh) break;
            s++; in += n_channel; dl += 2; step++;
        }
#pragma unroll
        for (int i = 0; i < kChunk; i++) if (i < nc) out[c0 + i] = acc[i];
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                     float T_thresh, float* weights_sum, float* depth, float* image, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!rays || !weights_sum || !depth || !image) return PNR_ERR_INVALID;
    if (M > 0 && (!sigmas || !rgbs || !deltas)) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), sigmas, rgbs, deltas, rays, M, N,
                       T_thresh, weights_sum, depth, image);
    return check_launch();
}

int pnr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                      const float* deltas, const int32_t* rays, const float* weights_sum, const float* image, uint32_t M,
                                      uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs, pnr_stream_t stream) {
    if (N == 0 || M == 0) return PNR_OK;
    if (!grad_weights_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs)
        return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), grad_weights_sum, grad_image, sigmas,
                       rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    return check_launch();
}

int pnr_composite_rays_flex_train_forward(const float* sigmas, const float* input, const float* deltas, const int32_t* rays, uint32_t M,
                                          uint32_t N, uint32_t n_channel, float T_thresh, float* output, pnr_stream_t stream) {
    if (n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
    if (N == 0 || n_channel == 0) return PNR_OK;
    if (!rays || !output) return PNR_ERR_INVALID;
    if (M > 0 && (!sigmas || !input || !deltas)) return PNR_ERR_INVALID;
    if (n_channel >= 4)
        hipLaunchKernelGGL(k_composite_flex_train_fwd_coop, dim3(cdiv(N * kLanesPerRay, kBlock)), dim3(kBlock), 0, as_stream(stream), sigmas, input,
                           deltas, rays, M, N, n_channel, T_thresh, output);
    else
        hipLaunchKernelGGL(k_composite_flex_train_fwd, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), sigmas, input, deltas, rays, M,
                           N, n_channel, T_thresh, output);
    return check_launch();
}

int pnr_composite_rays_flex_train_backward(const float* grad_output, const float* sigmas, const float* input, const float* deltas,
                                           const int32_t* rays, const float* output, uint32_t M, uint32_t N, uint32_t n_channel,
                                           float T_thresh, float* grad_input, pnr_stream_t stream) {
    (void)input; (void)output;  // unused by the reference kernel as well (raymarching.cu:767,770)
    if (n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
    if (N == 0 || M == 0 || n_channel == 0) return PNR_OK;
    if (!grad_output || !sigmas || !deltas || !rays || !grad_input) return PNR_ERR_INVALID;
    if (n_channel >= 4)
        hipLaunchKernelGGL(k_composite_flex_train_bwd_coop, dim3(cdiv(N * kLanesPerRay, kBlock)), dim3(kBlock), 0, as_stream(stream), grad_output,
                           sigmas, deltas, rays, M, N, n_channel, T_thresh, grad_input);
    else
        hipLaunchKernelGGL(k_composite_flex_train_bwd, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), grad_output, sigmas, deltas,
                           rays, M, N, n_channel, T_thresh, grad_input);
    return check_launch();
}

int pnr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t, const float* sigmas,
                       const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image, pnr_stream_t stream) {
    if (n_alive == 0) return PNR_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_composite_rays, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive, n_step, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return check_launch();
}

int pnr_composite_rays_flex(uint32_t n_alive, uint32_t n_step, uint32_t n_channel, float T_thresh, const int32_t* rays_alive,
                            const float* rays_t, const float* sigmas, const float* input, const float* deltas, const float* weights_sum,
                            float* output, pnr_stream_t stream) {
    (void)rays_t;  // read but never used by the reference kernel (raymarching.cu:1140)
    if (n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
    if (n_alive == 0 || n_channel == 0) return PNR_OK;
    if (!rays_alive || !sigmas || !input || !deltas || !weights_sum || !output) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_composite_rays_flex, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive, n_step, n_channel,
                       T_thresh, rays_alive, sigmas, input, deltas, weights_sum, output);
    return check_launch();
}

}  // extern "C"
