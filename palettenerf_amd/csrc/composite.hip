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
constexpr int kAhead = 8;   // samples fetched ahead of the serial recurrence in the one-thread-per-ray training kernels

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
    // A training batch has a few thousand rays of ~150 samples: with one thread per ray the walk is a chain of dependent memory round trips
    // (64 waves on the whole chip).  kAhead samples are fetched before the serial recurrence consumes them -- same operations, same order.
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    bool stop = false;
    for (uint32_t base = 0; base < num_steps && !stop; base += kAhead) {
        float sg[kAhead], d0[kAhead], d1[kAhead], cr[kAhead], cg[kAhead], cb[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const uint32_t k = base + u < num_steps ? base + u : num_steps - 1;   // clamped: stays inside the ray's rows
            sg[u] = s[k]; d0[u] = dl[(size_t)k * 2]; d1[u] = dl[(size_t)k * 2 + 1];
            cr[u] = c[(size_t)k * 3]; cg[u] = c[(size_t)k * 3 + 1]; cb[u] = c[(size_t)k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            if (stop || base + u >= num_steps) break;
            const float alpha = alpha_of(sg[u], d0[u]);
            const float w = alpha * T;
            r = fmaf(w, cr[u], r); g = fmaf(w, cg[u], g); b = fmaf(w, cb[u], b);
            t += d1[u];
            d = fmaf(w, t, d);
            ws += w;
            T *= 1.0f - alpha;
            if (T < T_thresh) stop = true;
        }
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
constexpr uint32_t kCoopMinSamples = 1u << 16;   // below this the exact sample-order kernels are used (small batches, the oracle-parity tests)
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

// rgb / depth / weights variant with 16 lanes per ray, one SAMPLE per lane: the transmittance of a group of 16 consecutive samples is an
// exclusive prefix product over the lanes (4 shuffle steps), T in front of the group is carried along.  A 600-sample ray takes 38 group
// steps instead of 600 serial ones (the longest ray of a wave sets the pace of the one-thread-per-ray kernel: 140-200 us per launch on
// a 4096-ray batch).  Same terms as raymarching.cu:504-580, summed in scan / tree order instead of sample order (~1e-7 relative).
__device__ __forceinline__ float group_excl_product(float v, int q, float& total) {   // over the 16 lanes of a ray; total = product of all 16
    float inc = v;
#pragma unroll
    for (int off = 1; off < (int)kLanesPerRay; off <<= 1) {
        const float up = __shfl_up(inc, off, kLanesPerRay);
        if (q >= off) inc *= up;
    }
    total = __shfl(inc, kLanesPerRay - 1, kLanesPerRay);
    const float ex = __shfl_up(inc, 1, kLanesPerRay);
    return q == 0 ? 1.0f : ex;
}
__device__ __forceinline__ float group_incl_sum(float v, int q, float& total) {
    float inc = v;
#pragma unroll
    for (int off = 1; off < (int)kLanesPerRay; off <<= 1) {
        const float up = __shfl_up(inc, off, kLanesPerRay);
        if (q >= off) inc += up;
    }
    total = __shfl(inc, kLanesPerRay - 1, kLanesPerRay);
    return inc;
}
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = kLanesPerRay / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kLanesPerRay);
    return v;
}

__global__ void __launch_bounds__(kBlock) k_composite_train_fwd_coop(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                     const float* __restrict__ deltas, const int32_t* __restrict__ rays, uint32_t M,
                                                                     uint32_t N, float T_thresh, float* __restrict__ weights_sum,
                                                                     float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = (blockIdx.x * kBlock + threadIdx.x) / kLanesPerRay;
    const int q = (int)(threadIdx.x % kLanesPerRay);
    if (n >= N) return;   // whole 16-lane groups leave together
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (num_steps != 0 && offset + num_steps <= M) {
        float T = 1.0f, t = 0.0f;   // transmittance / ray parameter in front of the current group
        for (uint32_t base = 0; base < num_steps && !(T < T_thresh); base += kLanesPerRay) {
            const uint32_t k = base + (uint32_t)q;
            const bool in = k < num_steps;
            const size_t row = (size_t)offset + (in ? k : num_steps - 1);
            const float alpha = in ? alpha_of(sigmas[row], deltas[row * 2]) : 0.0f;
            float keep, tsum;
            const float Tq = T * group_excl_product(1.0f - alpha, q, keep);          // transmittance in front of this lane's sample
            const float tq = t + group_incl_sum(in ? deltas[row * 2 + 1] : 0.0f, q, tsum);
            const float w = (in && !(Tq < T_thresh)) ? alpha * Tq : 0.0f;            // the reference stops after the sample that drops T below the threshold
            r = fmaf(w, rgbs[row * 3], r); g = fmaf(w, rgbs[row * 3 + 1], g); b = fmaf(w, rgbs[row * 3 + 2], b);
            d = fmaf(w, tq, d);
            ws += w;
            T *= keep;
            t += tsum;
        }
    }
    r = group_sum(r); g = group_sum(g); b = group_sum(b); ws = group_sum(ws); d = group_sum(d);
    if (q == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

// raymarching.cu:681-761 with the same lane layout: the running colour of the reference (r, g, b after sample k) is an inclusive prefix sum
__global__ void __launch_bounds__(kBlock) k_composite_train_bwd_coop(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                     const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                     const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                     const float* __restrict__ weights_sum, const float* __restrict__ image, uint32_t M,
                                                                     uint32_t N, float T_thresh, float* __restrict__ grad_sigmas,
                                                                     float* __restrict__ grad_rgbs) {
    const uint32_t n = (blockIdx.x * kBlock + threadIdx.x) / kLanesPerRay;
    const int q = (int)(threadIdx.x % kLanesPerRay);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad_image[index * 3], g1 = grad_image[index * 3 + 1], g2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2], ws_final = weights_sum[index];
    float T = 1.0f, r = 0, g = 0, b = 0;   // state in front of the current group
    for (uint32_t base = 0; base < num_steps && !(T < T_thresh); base += kLanesPerRay) {
        const uint32_t k = base + (uint32_t)q;
        const bool in = k < num_steps;
        const size_t row = (size_t)offset + (in ? k : num_steps - 1);
        const float dl0 = deltas[row * 2];
        const float alpha = in ? alpha_of(sigmas[row], dl0) : 0.0f;
        const float c0 = rgbs[row * 3], c1 = rgbs[row * 3 + 1], c2 = rgbs[row * 3 + 2];
        float keep, sr, sg, sb;
        const float Tq = T * group_excl_product(1.0f - alpha, q, keep);
        const bool live = in && !(Tq < T_thresh);
        const float w = live ? alpha * Tq : 0.0f;
        const float rq = r + group_incl_sum(w * c0, q, sr), gq = g + group_incl_sum(w * c1, q, sg), bq = b + group_incl_sum(w * c2, q, sb);
        if (live) {
            const float Ta = Tq * (1.0f - alpha);   // transmittance after this sample
            grad_rgbs[row * 3] = g0 * w; grad_rgbs[row * 3 + 1] = g1 * w; grad_rgbs[row * 3 + 2] = g2 * w;
            float acc = g0 * fmaf(Ta, c0, -(r_final - rq));
            acc = fmaf(g1, fmaf(Ta, c1, -(g_final - gq)), acc);
            acc = fmaf(g2, fmaf(Ta, c2, -(b_final - bq)), acc);
            acc = fmaf(gws, 1.0f - ws_final, acc);
            grad_sigmas[row] = dl0 * acc;
        }
        T *= keep;
        r += sr; g += sg; b += sb;
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
    bool stop = false;
    for (uint32_t base = 0; base < num_steps && !stop; base += kAhead) {   // kAhead samples fetched ahead, as in the forward
        float sg[kAhead], d0[kAhead], cr[kAhead], cg[kAhead], cb[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const uint32_t k = base + u < num_steps ? base + u : num_steps - 1;
            sg[u] = s[k]; d0[u] = dl[(size_t)k * 2];
            cr[u] = c[(size_t)k * 3]; cg[u] = c[(size_t)k * 3 + 1]; cb[u] = c[(size_t)k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            if (stop || base + u >= num_steps) break;
            const uint32_t k = base + u;
            const float alpha = alpha_of(sg[u], d0[u]);
            const float w = alpha * T;
            r = fmaf(w, cr[u], r); g = fmaf(w, cg[u], g); b = fmaf(w, cb[u], b);
            T *= 1.0f - alpha;
            gc[(size_t)k * 3] = g0 * w; gc[(size_t)k * 3 + 1] = g1 * w; gc[(size_t)k * 3 + 2] = g2 * w;
            float acc = g0 * fmaf(T, cr[u], -(r_final - r));
            acc = fmaf(g1, fmaf(T, cg[u], -(g_final - g)), acc);
            acc = fmaf(g2, fmaf(T, cb[u], -(b_final - b)), acc);
            acc = fmaf(gws, 1.0f - ws_final, acc);
            gs[k] = d0[u] * acc;
            if (T < T_thresh) stop = true;
        }
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

// SURVEY 8(b)'s "multi-map variant" of composite_rays_flex: the reference's PaletteNeRF loop issues six (seven without gui_mode) flex composites per march
// iteration over the SAME sigmas / deltas / rays_alive / weights_sum (palette/renderer.py:508-516), each a launch of one thread per ray that walks the ray's
// samples again to rebuild the same weights and reads its input rows with a stride of n_channel floats between lanes (a wave's twelve 4-byte loads of a
// 12-channel map each touch the same 24 lines).  Here a workgroup takes 64 consecutive alive rays:
//   phase 1 (one thread per ray)  the weights w[k] of the ray's samples, ONCE (raymarching.cu:1150-1176: alpha, T = 1 - weights_sum, stop at a dead sample,
//                                 stop after the sample that sees T < T_thresh) and how many count, into LDS;
//   phase 2 (one thread per (ray, channel) element, map by map)  the workgroup's input block of a map is one contiguous run of 64 x n_step x n_channel
//                                 floats: lanes walk it element by element -- coalesced for n_step == 1 (every heavy iteration), runs of n_channel floats
//                                 otherwise -- and each element folds its ray's weights in sample order: per channel the same fmaf chain as
//                                 k_composite_rays_flex, bit-identical outputs.
// n_step <= kMultiSteps (the inference loop's n_step is at most 8: nerf/renderer.py:357).  A negative ray id (a list that was not compacted) is skipped.
constexpr int kMultiSteps = 8;
constexpr uint32_t kMultiRays = 64;      // rays per workgroup (a late iteration has a few thousand rays: 256 per workgroup left most CUs idle)
constexpr uint32_t kMultiThreads = 256;
struct FlexMaps {
    uint32_t n_maps;
    uint32_t n_channel[PNR_FLEX_MAX_MAPS];
    uint32_t magic[PNR_FLEX_MAX_MAPS];      // floor(2^32 / n_channel) + 1: e / n_channel == umulhi(e, magic) for e < 2^16 x ...
    const float* input[PNR_FLEX_MAX_MAPS];
    float* output[PNR_FLEX_MAX_MAPS];
};

__global__ void __launch_bounds__(kMultiThreads) k_composite_rays_flex_multi(uint32_t n_alive, uint32_t n_step, float T_thresh, const int32_t* __restrict__ rays_alive,
                                                                          const float* __restrict__ sigmas, const float* __restrict__ deltas,
                                                                          const float* __restrict__ weights_sum, const FlexMaps maps) {
    __shared__ float w_s[kMultiRays * kMultiSteps];
    __shared__ int32_t idx_s[kMultiRays];
    __shared__ uint32_t cnt_s[kMultiRays];
    const uint32_t n0 = blockIdx.x * kMultiRays;
    if (threadIdx.x < kMultiRays) {
        const uint32_t r = threadIdx.x, n = n0 + r;
        uint32_t cnt = 0;
        int index = -1;
        if (n < n_alive) {
            index = rays_alive[n];
            if (index >= 0) {
                const float* s = sigmas + (size_t)n * n_step;
                const float* dl = deltas + (size_t)n * n_step * 2;
                float ws = weights_sum[index];
#pragma unroll
                for (int k = 0; k < kMultiSteps; k++) {
                    if ((uint32_t)k >= n_step || cnt != (uint32_t)k) continue;     // (cnt == k: every earlier sample counted and none stopped the ray)
                    if (dl[2 * k] == 0) continue;
                    const float alpha = alpha_of(s[k], dl[2 * k]);
                    const float T = 1.0f - ws;
                    const float w = alpha * T;
                    ws += w;
                    w_s[r * kMultiSteps + k] = w;
                    cnt = (T < T_thresh) ? 0x80000000u | (uint32_t)(k + 1) : (uint32_t)(k + 1);     // (the sample that sees T < T_thresh still counts; nothing behind it does)
                }
                cnt &= 0x7fffffffu;
            }
        }
        idx_s[r] = index;
        cnt_s[r] = cnt;
    }
    __syncthreads();
    const uint32_t rays_here = min(kMultiRays, n_alive - n0);
    for (uint32_t m = 0; m < maps.n_maps; m++) {
        const uint32_t nc = maps.n_channel[m], magic = maps.magic[m];
        const float* __restrict__ in_base = maps.input[m] + (size_t)n0 * n_step * nc;
        float* __restrict__ out = maps.output[m];
        const uint32_t total = rays_here * nc;
        for (uint32_t e = threadIdx.x; e < total; e += kMultiThreads) {
            const uint32_t r = nc == 1u ? e : __umulhi(e, magic), c = e - r * nc;     // (2^32 / 1 + 1 does not fit the magic's 32 bits)
            const uint32_t cnt = cnt_s[r];
            if (cnt == 0) continue;
            float* o = out + (size_t)idx_s[r] * nc + c;
            const float* in = in_base + (size_t)r * n_step * nc + c;
            float acc = *o;
#pragma unroll
            for (int k = 0; k < kMultiSteps; k++)
                if ((uint32_t)k < cnt) acc = fmaf(w_s[r * kMultiSteps + k], in[(size_t)k * nc], acc);
            *o = acc;
        }
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
    if (M >= kCoopMinSamples)   // training batches: 16 lanes per ray
        hipLaunchKernelGGL(k_composite_train_fwd_coop, dim3(cdiv(N * kLanesPerRay, kBlock)), dim3(kBlock), 0, as_stream(stream), sigmas, rgbs, deltas,
                           rays, M, N, T_thresh, weights_sum, depth, image);
    else
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
    if (M >= kCoopMinSamples)
        hipLaunchKernelGGL(k_composite_train_bwd_coop, dim3(cdiv(N * kLanesPerRay, kBlock)), dim3(kBlock), 0, as_stream(stream), grad_weights_sum,
                           grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    else
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
    if (n_step <= (uint32_t)kMultiSteps && g_opt_flex_coop) {     // the inference loop's schedule: the workgroup-cooperative form (coalesced rows; same bits)
        FlexMaps fm;
        fm.n_maps = 1; fm.n_channel[0] = n_channel; fm.magic[0] = (uint32_t)((1ull << 32) / n_channel) + 1u; fm.input[0] = input; fm.output[0] = output;
        hipLaunchKernelGGL(k_composite_rays_flex_multi, dim3(cdiv(n_alive, kMultiRays)), dim3(kMultiThreads), 0, as_stream(stream), n_alive, n_step, T_thresh, rays_alive,
                           sigmas, deltas, weights_sum, fm);
        return check_launch();
    }
    hipLaunchKernelGGL(k_composite_rays_flex, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive, n_step, n_channel,
                       T_thresh, rays_alive, sigmas, input, deltas, weights_sum, output);
    return check_launch();
}

int pnr_composite_rays_flex_multi(uint32_t n_alive, uint32_t n_step, float T_thresh, const int32_t* rays_alive, const float* rays_t, const float* sigmas,
                                  const float* deltas, const float* weights_sum, const pnr_flex_map* maps, uint32_t n_maps, pnr_stream_t stream) {
    (void)rays_t;
    if (n_maps > PNR_FLEX_MAX_MAPS) return PNR_ERR_UNSUPPORTED;
    if (n_maps > 0 && !maps) return PNR_ERR_INVALID;
    FlexMaps fm;
    fm.n_maps = 0;
    for (uint32_t m = 0; m < n_maps; m++) {
        if (maps[m].n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
        if (maps[m].n_channel == 0) continue;                                   // (as pnr_composite_rays_flex: nothing to do for an empty map)
        if (n_alive > 0 && (!maps[m].input || !maps[m].output)) return PNR_ERR_INVALID;
        fm.n_channel[fm.n_maps] = maps[m].n_channel; fm.magic[fm.n_maps] = (uint32_t)((1ull << 32) / maps[m].n_channel) + 1u;
        fm.input[fm.n_maps] = maps[m].input; fm.output[fm.n_maps] = maps[m].output;
        fm.n_maps++;
    }
    if (n_alive == 0 || fm.n_maps == 0) return PNR_OK;
    if (!rays_alive || !sigmas || !deltas || !weights_sum) return PNR_ERR_INVALID;
    if (n_step > (uint32_t)kMultiSteps) {     // beyond the inference loop's schedule: the maps one by one through the single-map kernel (same results, n_maps launches)
        for (uint32_t m = 0; m < fm.n_maps; m++)
            hipLaunchKernelGGL(k_composite_rays_flex, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive, n_step, fm.n_channel[m], T_thresh,
                               rays_alive, sigmas, fm.input[m], deltas, weights_sum, fm.output[m]);
        return check_launch();
    }
    hipLaunchKernelGGL(k_composite_rays_flex_multi, dim3(cdiv(n_alive, kMultiRays)), dim3(kMultiThreads), 0, as_stream(stream), n_alive, n_step, T_thresh, rays_alive, sigmas,
                       deltas, weights_sum, fm);
    return check_launch();
}

}  // extern "C"
