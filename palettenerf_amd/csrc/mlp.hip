// mlp.hip -- the fields' small bias-free MLPs (2 or 3 layers, every width <= 64, ReLU or ELU between layers) as ONE launch forward
// and ONE launch backward for training, on the gfx950 matrix cores (nerf/network.py:33-93, palette/network.py:60-153 build them as
// nn.Linear(bias=False) stacks; autograd runs each as 3 library GEMMs + activation kernels forward and 2 GEMMs + an activation kernel per
// layer backward, every one a round trip of a [6e5, 64] activation through HBM at ~2 TB/s).
//
// MI355X formulation: a wave owns a tile of 32 samples and keeps every activation of the tile feature-major in registers --
// the D fragment of v_mfma_f32_32x32x2_f32 (register r of lane (s, h) = feature frag_row(r, h) of sample s) is exactly the B operand
// the next layer needs, so a layer is a chain of MFMAs against weights staged once per workgroup in LDS (exact fp32 fma chains).
// The backward recomputes the hidden activations from X (nothing but X and dY is read), propagates dY through the transposed
// weights the same way, and forms the weight gradients dW_l = dZ_l^T A_{l-1} with the SAMPLE as the MFMA k dimension: the two operands
// are written sample-major into a per-wave LDS tile and read back as "lane = feature, k = sample parity".  dW lives in accumulator
// registers for the whole launch, is reduced over the 4 waves through LDS and over workgroups by a second tiny launch in a fixed
// order (deterministic, no atomics).  HBM traffic: X and dY read once, dX written once.
// Two arithmetic forms of each launch: the exact fp32 matrix instructions described above (k_mlp_fwd / k_mlp_bwd) and, by default since round 4, split-fp16
// products with per-tile power-of-two scaling on v_mfma_f32_32x32x16_f16 (k_mlp_fwd_h / k_mlp_bwd_h, further down); pnr_mlp_pack writes the weights in both.
#include "field_core.hpp"
#include <initializer_list>

namespace pnr {

constexpr int kMlpThreads = 256;
constexpr int kMlpWaves = kMlpThreads / PNR_WAVE;
constexpr int kStage = 65;                       // floats per staged sample row (odd: conflict-free column walks)
constexpr int kStageFloats = 33 * kStage;        // one 32-sample tile + a row that swallows the lanes beyond the tile
constexpr uint32_t kMlpMaxBlocks = 256;          // one persistent workgroup per CU
__host__ __device__ constexpr uint32_t tiles32(uint32_t n) { return (n + 31u) / 32u; }

struct MlpPlan {
    uint32_t n_layers, act, out_act;   // out_act 1: sigmoid on the last layer's output
    uint32_t dims[4];
    uint32_t w_off[3], wt_off[3];    // float offsets of the packed W_l / W_l^T slots
    uint32_t dw_off[3];              // float offsets of dW_l inside a partial row
    uint32_t packed_floats, dw_floats;
    uint32_t magic[4];               // floor(2^32 / dims[d]) + 1: f / dims[d] = umulhi(f, magic[d]) exactly for f < 2^16
    uint32_t lm, tail, tail_magic;   // lm = 1: the first 32 input columns come from a level-major encoder output [16][B][2], `tail` more from a row-major [B][tail]
};

// tiles of 32 features the kernels use at layer boundary d: what the width needs at the input and the output, always 2 at hidden
// boundaries (zero padded), so that 8 kernel instances cover every stack
__host__ __device__ inline uint32_t plan_tiles(const MlpPlan& p, uint32_t d) { return (d == 0 || d == p.n_layers) ? tiles32(p.dims[d]) : 2u; }

static bool make_plan(const pnr_mlp_desc* d, MlpPlan& p) {
    if (!d || d->n_layers < 2 || d->n_layers > 3) return false;
    const int hidden = d->activation & ~PNR_MLP_OUT_SIGMOID;
    if (hidden != 0 && hidden != 1) return false;
    p.n_layers = d->n_layers;
    p.act = (uint32_t)hidden;
    p.out_act = (d->activation & PNR_MLP_OUT_SIGMOID) ? 1u : 0u;
    uint32_t off = 0, dw = 0;
    for (uint32_t l = 0; l <= d->n_layers; l++) {
        if (d->dims[l] == 0 || d->dims[l] > 64) return false;
        p.dims[l] = d->dims[l];
        p.magic[l] = (uint32_t)((1ull << 32) / d->dims[l]) + 1u;
    }
    for (uint32_t l = 0; l < d->n_layers; l++) { p.w_off[l] = off; off += plan_tiles(p, l + 1) * plan_tiles(p, l) * 1024; }
    for (uint32_t l = 0; l < d->n_layers; l++) { p.wt_off[l] = off; off += plan_tiles(p, l + 1) * plan_tiles(p, l) * 1024; }
    for (uint32_t l = 0; l < d->n_layers; l++) { p.dw_off[l] = dw; dw += p.dims[l + 1] * p.dims[l]; }
    p.packed_floats = off;
    p.dw_floats = dw;
    p.lm = p.tail = 0;
    p.tail_magic = 1;
    return true;
}

struct MlpWeights { const float* w[3]; };
struct MlpGrads { float* dw[3]; };

// The blob pnr_mlp_pack writes: [fp32 part: packed_floats floats][split-fp16 part: the same byte size][inverse weight scales: 4 floats].
//   fp32 slot of a matrix M [rows][cols]: [row tile][k tile][r 0..15][lane] = M[rt*32 + lane%32][kt*32 + frag_row(r, lane/32)]
//   fp16 slot (same byte offset inside its part, same size): [row tile][k block kb = 0..2*NKT-1] blocks of 2 KiB = [hi: 64 lanes x 8 halfs][lo: same],
//     element j of lane = the split of M[rt*32 + lane%32][f16_col_from_frag(kb, lane/32, j)] * wscale_l, where wscale_l is the power of two that
//     puts max|W_l| into [2^14, 2^15) (fp16 holds 11 bits from 2^-14 up: weights at the usual 1/sqrt(fan_in) scale would leave their lo halves
//     subnormal); 1 / wscale_l is stored behind the two parts and folded into the kernels' unscaling of each layer's accumulators.
__host__ __device__ inline uint32_t mlp_f16_part_floats(const MlpPlan& p) { return p.packed_floats; }           // offset of the fp16 part, in floats
__host__ __device__ inline uint32_t mlp_scales_floats(const MlpPlan& p) { return 2u * p.packed_floats; }       // offset of the 4 inverse scales
__device__ __forceinline__ float pow2_scale_for(float mx, float* inv) {   // s = 2^k with s * mx in [2^14, 2^15); mx == 0 or not finite: 1
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    if (e == 0 || e == 255) { *inv = 1.0f; return 1.0f; }
    e = e < 15 ? 15 : (e > 250 ? 250 : e);
    *inv = __uint_as_float((uint32_t)(e - 14) << 23);          // 2^(e - 127 - 14)
    return __uint_as_float((uint32_t)(268 - e) << 23);         // 2^(14 - (e - 127))
}
constexpr uint32_t kPackChunks = 4;     // workgroups per slot (each repeats the layer's max: 16 loads per thread, one wave reduction)
__global__ void __launch_bounds__(256) k_mlp_pack(MlpPlan p, MlpWeights ws, float* __restrict__ packed) {
    // slot blockIdx.x (W_l: l, W_l^T: n_layers + l), quarter blockIdx.y of it: the layer's max|w| first, then both formats of the quarter
    __shared__ float red[4];
    const uint32_t slot = blockIdx.x;
    const bool tr = slot >= p.n_layers;
    const uint32_t l = tr ? slot - p.n_layers : slot;
    const uint32_t in = p.dims[l], out = p.dims[l + 1];
    float mx = 0.0f;
    for (uint32_t e = threadIdx.x; e < in * out; e += 256) mx = fmaxf(mx, fabsf(ws.w[l][e]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    float winv;
    const float wscale = pow2_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), &winv);
    if (!tr && blockIdx.y == 0 && threadIdx.x == 0) packed[mlp_scales_floats(p) + l] = winv;
    const uint32_t off = tr ? p.wt_off[l] : p.w_off[l];
    const uint32_t rows = tr ? in : out, cols = tr ? out : in;
    const uint32_t nkt = plan_tiles(p, tr ? l + 1 : l), size = plan_tiles(p, tr ? l : l + 1) * nkt * 1024;
    _Float16* hpart = reinterpret_cast<_Float16*>(packed + mlp_f16_part_floats(p) + off);
    for (uint32_t q = blockIdx.y * 256 + threadIdx.x; q < size; q += 256 * kPackChunks) {
        {   // fp32 element q of the slot
            const uint32_t lane = q & 63, r = (q >> 6) & 15, t = q >> 10, kt = t % nkt, rt = t / nkt;
            const uint32_t row = rt * 32 + (lane & 31), col = kt * 32 + (uint32_t)frag_row((int)r, (int)(lane >> 5));
            float v = 0.0f;
            if (row < rows && col < cols) v = tr ? ws.w[l][(size_t)col * in + row] : ws.w[l][(size_t)row * in + col];
            packed[off + q] = v;
        }
        {   // fp16 pair q of the slot: block b = q / 512, lane, element j
            const uint32_t b = q >> 9, lane = (q >> 3) & 63, j = q & 7, kb = b % (2 * nkt), rt = b / (2 * nkt);
            const uint32_t row = rt * 32 + (lane & 31), col = (uint32_t)f16_col_from_frag((int)kb, (int)(lane >> 5), (int)j);
            float v = 0.0f;
            if (row < rows && col < cols) v = (tr ? ws.w[l][(size_t)col * in + row] : ws.w[l][(size_t)row * in + col]) * wscale;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            hpart[(size_t)b * 1024 + lane * 8 + j] = hi;
            hpart[(size_t)b * 1024 + 512 + lane * 8 + j] = lo;
        }
    }
}

// out[rt] = sum_kt W[rt][kt] . a[kt]  (feature-major fragments; NRT x NKT tiles of 32)
template <int NRT, int NKT>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ wp, const f32x16 (&a)[2], f32x16 (&out)[2], int lane) {
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        out[rt] = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; kt++) {
            const float* w = wp + (size_t)((rt * NKT + kt) * 16) * 64;
#pragma unroll
#ifdef PNR_MLP_FAKE   // timing builds only (WRONG results): every PNR_MLP_FAKE-th matrix instruction, the operand reads stay -- what is left of the kernel without its matrix work
            for (int r = 0; r < 16; r++) { const float wv = w[r * 64 + lane]; if (r % PNR_MLP_FAKE == 0) out[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, a[kt][r], out[rt], 0, 0, 0); else out[rt][r] += wv; }
#else
            for (int r = 0; r < 16; r++) out[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r * 64 + lane], a[kt][r], out[rt], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ELU without a divergent call: exp(min(z, 0)) - 1 is within 6e-8 absolute of expm1 (the activations are O(1)); selected by sign
__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == 0) return fmaxf(z, 0.0f);
    const float e = expf(fminf(z, 0.0f)) - 1.0f;
    return z > 0.0f ? z : e;
}
// derivative from the activation's OUTPUT (ReLU: h > 0; ELU, alpha 1: h > 0 ? 1 : h + 1)
__device__ __forceinline__ float act_grad(float h, int act) { return h > 0.0f ? 1.0f : (act == 0 ? 0.0f : h + 1.0f); }

template <int NT>
__device__ __forceinline__ void apply_act(f32x16 (&v)[2], int act) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) v[t][r] = act_fwd(v[t][r], act);
}
template <int NT>
__device__ __forceinline__ void mul_act_grad(f32x16 (&g)[2], const f32x16 (&h)[2], int act) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) g[t][r] *= act_grad(h[t][r], act);
}

// The staging tiles are private to a wave: its LDS instructions execute in issue order, so a later read sees an earlier write of any
// lane of the same wave; all that is needed is that the compiler keeps the program order (no workgroup barrier, waves run independently).
// A fence or __syncthreads() here would also drain vmcnt, i.e. wait for the next tile's prefetch right away.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// sample-major staging tile of one wave: buf[sample 0..31][feature 0..63] (row stride kStage), row 32 = trash.
// A wave's 32 x width tile is held raw in registers: the NEXT tile's global loads are issued before the current tile's matrix work and land while it runs
// (one wave per SIMD: nothing else would hide them).  The tile's 32 x width floats are ONE contiguous run that starts on a 16-byte boundary whatever the width
// (the arrays are 16-byte aligned: the entry points check), so it moves as 16-byte requests: registers 4 k4 .. 4 k4 + 3 = elements 4 (lane + 64 k4) ... + 3 of the
// row-major tile.  (Rounds 1-4 moved 4 bytes per request, every one with its own index clamp and bound select: a fifth of the vector instructions of a training
// launch.)  The fast path has no bound checks at all; the launch's one partial tile is staged / stored by the rolled loops below, straight from / to global memory.
template <int NT>
__device__ __forceinline__ void raw_load(float (&v)[16 * NT], const float* __restrict__ g, uint32_t row0, uint32_t B, uint32_t width, int lane) {
    const uint32_t total = B * width, base = row0 * width;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    const uint32_t last4 = (total - 4u) >> 2;     // (B >= 1 and width >= 4 or B >= 4 ... : total >= 4 is checked by the entry points)  requests beyond the tile -- or, in
                                                  // the partial tile, beyond the array -- re-read the last float4 that lies inside the array; nothing but the load here:
                                                  // any use of the value (even a select) would make the compiler wait for it on the spot and serialise the tile's loads
#pragma unroll
    for (int k4 = 0; k4 < 4 * NT; k4++) {
        const uint32_t i = (base >> 2) + (uint32_t)lane + 64u * k4;
        const float4 x = g4[i < last4 ? i : last4];
        v[4 * k4] = x.x; v[4 * k4 + 1] = x.y; v[4 * k4 + 2] = x.z; v[4 * k4 + 3] = x.w;
    }
}
template <int NT>
__device__ __forceinline__ void raw_to_stage(float* __restrict__ buf, const float (&v)[16 * NT], const float* __restrict__ g, uint32_t row0, uint32_t B, uint32_t width,
                                             uint32_t magic, int lane, const int kStage = pnr::kStage /* row stride of this tile (the X tile of the split-fp16 backward is narrower) */,
                                             const float* __restrict__ g_sig = nullptr /* the forward's sigmoid output: the staged values are dY (1 - y) y (the caller applied that to v) */) {
    if (row0 + 32u <= B) {       // (wave-uniform) a full tile: nothing to zero, and with a width that is a multiple of 4 a request's four values sit in one row
#pragma unroll
        for (int k4 = 0; k4 < 4 * NT; k4++) {
            const uint32_t f0 = 4u * ((uint32_t)lane + 64u * k4);
            if ((width & 3u) == 0) {
                uint32_t r = __umulhi(f0, magic);
                const uint32_t c = f0 - r * width;
                r = r < 32u ? r : 32u;
                float* d = buf + r * kStage + c;
                d[0] = v[4 * k4]; d[1] = v[4 * k4 + 1]; d[2] = v[4 * k4 + 2]; d[3] = v[4 * k4 + 3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t f = f0 + j;
                    uint32_t r = __umulhi(f, magic);
                    const uint32_t c = f - r * width;
                    r = r < 32u ? r : 32u;
                    buf[r * kStage + c] = v[4 * k4 + j];
                }
            }
        }
    } else {                     // the launch's partial tile (or a wave beyond the last row): element by element from global memory, zeros past the end
        const uint32_t total = B * width, base = row0 * width;
#pragma unroll 1
        for (uint32_t f = (uint32_t)lane; f < 32u * width; f += 64u) {
            const uint32_t r = __umulhi(f, magic), c = f - r * width;
            float x = 0.0f;
            if (base + f < total) {
                x = g[base + f];
                if (g_sig) { const float y = g_sig[base + f]; x = (x * (1.0f - y)) * y; }
            }
            buf[r * kStage + c] = x;
        }
    }
    // zero the padding columns (they are MFMA operands too): lane = (row, parity), columns width + parity, width + parity + 2, ...
    for (uint32_t c = width + ((uint32_t)lane >> 5); c < 32u * NT; c += 2) buf[((uint32_t)lane & 31u) * kStage + c] = 0.0f;
}
template <int NT>
__device__ __forceinline__ void stage_to_global(const float* __restrict__ buf, float* __restrict__ g, uint32_t row0, uint32_t B, uint32_t width, uint32_t magic, int lane,
                                                bool sigmoid = false) {
    const uint32_t total = B * width, base = row0 * width;
    if (row0 + 32u <= B) {
        float4* __restrict__ g4 = reinterpret_cast<float4*>(g);
#pragma unroll
        for (int k4 = 0; k4 < 4 * NT; k4++) {
            if (256u * k4 >= 32u * width) break;      // wave-uniform: requests wholly beyond the tile
            const uint32_t f0 = 4u * ((uint32_t)lane + 64u * k4);
            if (f0 < 32u * width) {
                float x[4];
                if ((width & 3u) == 0) {
                    const uint32_t r = __umulhi(f0, magic), c = f0 - r * width;
                    const float* sp = buf + r * kStage + c;
                    x[0] = sp[0]; x[1] = sp[1]; x[2] = sp[2]; x[3] = sp[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t f = f0 + j, r = __umulhi(f, magic), c = f - r * width;
                        x[j] = buf[r * kStage + c];
                    }
                }
                if (sigmoid) {
#pragma unroll
                    for (int j = 0; j < 4; j++) x[j] = 1.0f / (1.0f + expf(-x[j]));   // torch.sigmoid
                }
                g4[(base + f0) >> 2] = make_float4(x[0], x[1], x[2], x[3]);
            }
        }
    } else {
#pragma unroll 1
        for (uint32_t f = (uint32_t)lane; f < 32u * width; f += 64u) {
            if (base + f < total) {
                const uint32_t r = __umulhi(f, magic), c = f - r * width;
                float x = buf[r * kStage + c];
                if (sigmoid) x = 1.0f / (1.0f + expf(-x));
                g[base + f] = x;
            }
        }
    }
}

// X whose first 32 columns are a hash-grid encoder output in its native level-major layout enc [16][B][2] (no [B,32] copy is ever made) and
// whose remaining `tail` columns (NT == 2) are a row-major [B][tail] tensor.  A level's 32 samples x 2 channels of a tile are 64 contiguous floats that start on an
// 8-byte boundary (B may be odd): registers 2 k2, 2 k2 + 1 = both channels of sample lane & 31 on level 2 k2 + (lane >> 5) -- 8-byte requests, two levels per
// instruction; registers 16.. hold the tail like raw_load.  Rows beyond B are clamped here and zeroed when the tile is staged.
template <int NT>
__device__ __forceinline__ void raw_load_lm(float (&v)[16 * NT], const float* __restrict__ enc, const float* __restrict__ tail_src, uint32_t wt, uint32_t row0,
                                            uint32_t B, int lane) {
    const float2* __restrict__ e2 = reinterpret_cast<const float2*>(enc);
    const uint32_t s = (uint32_t)lane & 31u, hl = (uint32_t)lane >> 5;
    const uint32_t row = row0 + s < B ? row0 + s : B - 1u;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const float2 x = e2[(size_t)(2 * k2 + hl) * B + row];
        v[2 * k2] = x.x; v[2 * k2 + 1] = x.y;
    }
    if constexpr (NT == 2) {
        const uint32_t ttotal = B * wt, tbase = row0 * wt;
        const float4* __restrict__ t4 = reinterpret_cast<const float4*>(tail_src);
        const uint32_t last4 = ttotal >= 4u ? (ttotal - 4u) >> 2 : 0u;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const uint32_t i = (tbase >> 2) + (uint32_t)lane + 64u * k4;
            const float4 x = t4[i < last4 ? i : last4];
            v[16 + 4 * k4] = x.x; v[16 + 4 * k4 + 1] = x.y; v[16 + 4 * k4 + 2] = x.z; v[16 + 4 * k4 + 3] = x.w;
        }
    }
}
template <int NT>
__device__ __forceinline__ void raw_to_stage_lm(float* __restrict__ buf, const float (&v)[16 * NT], const float* __restrict__ tail_src, uint32_t wt, uint32_t tail_magic,
                                                uint32_t row0, uint32_t B, int lane, const int kStage = pnr::kStage) {
    const uint32_t s = (uint32_t)lane & 31u, hl = (uint32_t)lane >> 5;
    const bool full = row0 + 32u <= B;             // wave-uniform
    if (full) {
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { float* d = buf + s * kStage + 2 * (2 * k2 + hl); d[0] = v[2 * k2]; d[1] = v[2 * k2 + 1]; }
    } else {
        const bool live = row0 + s < B;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { float* d = buf + s * kStage + 2 * (2 * k2 + hl); d[0] = live ? v[2 * k2] : 0.0f; d[1] = live ? v[2 * k2 + 1] : 0.0f; }
    }
    if constexpr (NT == 2) {
        if (full) {
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t f = 4u * ((uint32_t)lane + 64u * k4) + j;
                    uint32_t r = __umulhi(f, tail_magic);
                    const uint32_t c = f - r * wt;
                    r = r < 32u ? r : 32u;
                    buf[r * kStage + 32 + c] = v[16 + 4 * k4 + j];
                }
        } else {
            const uint32_t ttotal = B * wt, tbase = row0 * wt;
#pragma unroll 1
            for (uint32_t f = (uint32_t)lane; f < 32u * wt; f += 64u) {
                const uint32_t r = __umulhi(f, tail_magic), c = f - r * wt;
                buf[r * kStage + 32 + c] = tbase + f < ttotal ? tail_src[tbase + f] : 0.0f;
            }
        }
        for (uint32_t c = 32u + wt + ((uint32_t)lane >> 5); c < 64u; c += 2) buf[((uint32_t)lane & 31u) * kStage + c] = 0.0f;
    }
}
// dX of the level-major columns back in level-major layout [16][B][2] (what the table-gradient kernels take)
__device__ __forceinline__ void stage_to_global_lm(const float* __restrict__ buf, float* __restrict__ denc, uint32_t row0, uint32_t B, int lane) {
    float2* __restrict__ d2 = reinterpret_cast<float2*>(denc);
    const uint32_t s = (uint32_t)lane & 31u, hl = (uint32_t)lane >> 5;
    if (row0 + s < B) {
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            const float* sp = buf + s * kStage + 2 * (2 * k2 + hl);
            d2[(size_t)(2 * k2 + hl) * B + row0 + s] = make_float2(sp[0], sp[1]);
        }
    }
}

template <int NT>
__device__ __forceinline__ void frag_from_stage(const float* __restrict__ buf, int lane, f32x16 (&a)[2], const int kStage = pnr::kStage) {
    const int s = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) a[t][r] = buf[s * kStage + t * 32 + frag_row(r, h)];
}
template <int NT>
__device__ __forceinline__ void frag_to_stage(float* __restrict__ buf, int lane, const f32x16 (&a)[2]) {
    const int s = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) buf[s * kStage + t * 32 + frag_row(r, h)] = t < NT ? a[t][r] : 0.0f;   // absent tiles: zero operands
}

// acc[rt][ct] += sum over the tile's 32 samples of G[s][rt*32 + i] * A[s][ct*32 + j]   (G, A: staged sample-major tiles)
template <int NRT, int NCT>
__device__ __forceinline__ void wgrad_accumulate(f32x16 (&acc)[2][2], const float* __restrict__ G, const float* __restrict__ A, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
#ifdef PNR_MLP_FAKE
            for (int p = 0; p < 16; p++) {
                const float gv = G[(2 * p + h) * kStage + rt * 32 + c], av = A[(2 * p + h) * kStage + ct * 32 + c];
                if (p % PNR_MLP_FAKE == 0) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, av, acc[rt][ct], 0, 0, 0); else acc[rt][ct][p] += gv * av;
            }
#else
            for (int p = 0; p < 16; p++)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(G[(2 * p + h) * kStage + rt * 32 + c], A[(2 * p + h) * kStage + ct * 32 + c], acc[rt][ct], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// tiles of 32 features at layer boundary d (0 = input ... NL = output); hidden widths always take 2 tiles (<= 64, zero padded)
template <int NL, int TI, int TO>
__host__ __device__ constexpr int tiles_at(int d) { return d == 0 ? TI : (d == NL ? TO : 2); }

// reduce the 4 waves' dW through LDS (`red`: the staging area, 8 tiles of 2080 floats >= 64 x 64 x 3), then one partial row per workgroup
template <int NL, int TI, int TO>
__device__ __forceinline__ void mlp_dw_to_partial(const MlpPlan& p, const f32x16 (&dw0)[2][2], const f32x16 (&dw1)[2][2], const f32x16 (&dw2)[2][2], float* __restrict__ red,
                                                  float* __restrict__ partial, int lane, int wave) {
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * p.dw_floats;
    const int c = lane & 31, hh = lane >> 5;
    for (int wv = 0; wv < kMlpWaves; wv++) {
        if (wave == wv) {
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const uint32_t in = p.dims[l], outd = p.dims[l + 1];
#pragma unroll
                for (int rt = 0; rt < tiles_at<NL, TI, TO>(l + 1); rt++)
#pragma unroll
                    for (int ct = 0; ct < tiles_at<NL, TI, TO>(l); ct++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const uint32_t row = rt * 32 + frag_row(r, hh), col = ct * 32 + c;
                            if (row < outd && col < in) {
                                float* q = red + p.dw_off[l] + row * in + col;
                                const float v = l == 0 ? dw0[rt][ct][r] : (l == 1 ? dw1[rt][ct][r] : dw2[rt][ct][r]);
                                *q = (wv == 0 ? 0.0f : *q) + v;
                            }
                        }
            }
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < p.dw_floats; e += kMlpThreads) out[e] = red[e];
}

template <int NL, int TI, int TO, int ACT>
__global__ void __launch_bounds__(kMlpThreads) k_mlp_fwd(MlpPlan p, const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ x_tail, uint32_t B,
                                                         float* __restrict__ y) {
    extern __shared__ float lds[];
    float* w = lds;
    const uint32_t wfloats = p.wt_off[0];   // forward slots only
    for (uint32_t i = threadIdx.x * 4; i < wfloats; i += kMlpThreads * 4) *reinterpret_cast<float4*>(&w[i]) = *reinterpret_cast<const float4*>(&packed[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = lds + wfloats + wave * kStageFloats;
    __syncthreads();
    const uint32_t nblock_tiles = (B + 32 * kMlpWaves - 1) / (32 * kMlpWaves);
    float xr[16 * TI];
    if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, (blockIdx.x * kMlpWaves + wave) * 32, B, lane);
    else raw_load<TI>(xr, x, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[0], lane);
    for (uint32_t bt = blockIdx.x; bt < nblock_tiles; bt += gridDim.x) {
        const uint32_t row0 = (bt * kMlpWaves + wave) * 32;
        if (p.lm) raw_to_stage_lm<TI>(stage, xr, x_tail, p.tail, p.tail_magic, row0, B, lane);
        else raw_to_stage<TI>(stage, xr, x, row0, B, p.dims[0], p.magic[0], lane);
        wave_sync();
        if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, ((bt + gridDim.x) * kMlpWaves + wave) * 32, B, lane);   // the next tile's loads fly during this one's matrix work
        else raw_load<TI>(xr, x, ((bt + gridDim.x) * kMlpWaves + wave) * 32, B, p.dims[0], lane);
        f32x16 a[2], o[2];
        frag_from_stage<TI>(stage, lane, a);
        mlp_layer<2, TI>(w + p.w_off[0], a, o, lane);
        apply_act<2>(o, ACT);
        if constexpr (NL == 3) {
            mlp_layer<2, 2>(w + p.w_off[1], o, a, lane);
            apply_act<2>(a, ACT);
            mlp_layer<TO, 2>(w + p.w_off[2], a, o, lane);
        } else {
            mlp_layer<TO, 2>(w + p.w_off[1], o, a, lane);
            o[0] = a[0];
            if constexpr (TO == 2) o[1] = a[1];
        }
        wave_sync();
        frag_to_stage<TO>(stage, lane, o);
        wave_sync();
        stage_to_global<TO>(stage, y, row0, B, p.dims[NL], p.magic[NL], lane, p.out_act != 0);   // + the colour heads' sigmoid, on the real outputs only
        wave_sync();
    }
}

template <int NL, int TI, int TO, int ACT>
__global__ void __launch_bounds__(kMlpThreads) k_mlp_bwd(MlpPlan p, const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ x_tail, const float* __restrict__ dy,
                                                         const float* __restrict__ yout /* the forward's output: out_act only */, uint32_t B, float* __restrict__ dx,
                                                         float* __restrict__ partial /* [gridDim.x][dw_floats] */) {
    extern __shared__ float lds[];
    float* w = lds;
    for (uint32_t i = threadIdx.x * 4; i < p.packed_floats; i += kMlpThreads * 4) *reinterpret_cast<float4*>(&w[i]) = *reinterpret_cast<const float4*>(&packed[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* GA = lds + p.packed_floats + wave * 2 * kStageFloats;   // gradient side of the weight-gradient products
    float* GB = GA + kStageFloats;                                 // activation side
    __syncthreads();
    f32x16 dw0[2][2], dw1[2][2], dw2[2][2];   // only the tiles a layer has are touched (the others are never materialised)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) { dw0[i][j] = zero16(); dw1[i][j] = zero16(); dw2[i][j] = zero16(); }
    const uint32_t nblock_tiles = (B + 32 * kMlpWaves - 1) / (32 * kMlpWaves);
    float xr[16 * TI], yn[16 * TO], sn[16 * TO];
    if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, (blockIdx.x * kMlpWaves + wave) * 32, B, lane);
    else raw_load<TI>(xr, x, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[0], lane);
    raw_load<TO>(yn, dy, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[NL], lane);
    if (yout) raw_load<TO>(sn, yout, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[NL], lane);
    for (uint32_t bt = blockIdx.x; bt < nblock_tiles; bt += gridDim.x) {
        const uint32_t row0 = (bt * kMlpWaves + wave) * 32;
        if (p.lm) raw_to_stage_lm<TI>(GB, xr, x_tail, p.tail, p.tail_magic, row0, B, lane);
        else raw_to_stage<TI>(GB, xr, x, row0, B, p.dims[0], p.magic[0], lane);
        if (yout) {   // dZ = dY (1 - y) y: sigmoid_backward, in the raw layout both tiles share
#pragma unroll
            for (int k = 0; k < 16 * TO; k++) yn[k] = (yn[k] * (1.0f - sn[k])) * sn[k];
        }
        raw_to_stage<TO>(GA, yn, dy, row0, B, p.dims[NL], p.magic[NL], lane, kStage, yout);
        wave_sync();
        const uint32_t next0 = ((bt + gridDim.x) * kMlpWaves + wave) * 32;
        raw_load<TO>(yn, dy, next0, B, p.dims[NL], lane);          // the next tile's dY flies during this tile's matrix work
        if (yout) raw_load<TO>(sn, yout, next0, B, p.dims[NL], lane);
        // recompute the hidden activations h1 (after layer 0) and h2 (after layer 1, NL == 3)
        f32x16 xin[2], h1[2], h2[2], g[2], t[2];
        frag_from_stage<TI>(GB, lane, xin);
        frag_from_stage<TO>(GA, lane, g);
        mlp_layer<2, TI>(w + p.w_off[0], xin, h1, lane);
        apply_act<2>(h1, ACT);
        if constexpr (NL == 3) {
            mlp_layer<2, 2>(w + p.w_off[1], h1, h2, lane);
            apply_act<2>(h2, ACT);
            // layer 2: GA = dY, GB <- h2
            wave_sync();
            frag_to_stage<2>(GB, lane, h2);
            wave_sync();
            wgrad_accumulate<TO, 2>(dw2, GA, GB, lane);
            mlp_layer<2, TO>(w + p.wt_off[2], g, t, lane);          // dH2 = W2^T dY
            mul_act_grad<2>(t, h2, ACT);
            g[0] = t[0]; g[1] = t[1];
            // layer 1: GA <- dZ2, GB <- h1
            wave_sync();
            frag_to_stage<2>(GA, lane, g);
            frag_to_stage<2>(GB, lane, h1);
            wave_sync();
            wgrad_accumulate<2, 2>(dw1, GA, GB, lane);
            mlp_layer<2, 2>(w + p.wt_off[1], g, t, lane);           // dH1 = W1^T dZ2
            mul_act_grad<2>(t, h1, ACT);
            g[0] = t[0]; g[1] = t[1];
        } else {
            // layer 1 (the last): GA = dY, GB <- h1
            wave_sync();
            frag_to_stage<2>(GB, lane, h1);
            wave_sync();
            wgrad_accumulate<TO, 2>(dw1, GA, GB, lane);
            mlp_layer<2, TO>(w + p.wt_off[1], g, t, lane);          // dH1 = W1^T dY
            mul_act_grad<2>(t, h1, ACT);
            g[0] = t[0]; g[1] = t[1];
        }
        // layer 0: GA <- dZ1, GB <- X
        wave_sync();
        frag_to_stage<2>(GA, lane, g);
        if (p.lm) raw_to_stage_lm<TI>(GB, xr, x_tail, p.tail, p.tail_magic, row0, B, lane);
        else raw_to_stage<TI>(GB, xr, x, row0, B, p.dims[0], p.magic[0], lane);
        wave_sync();
        if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, next0, B, lane);   // X is needed twice per tile: its prefetch starts after the second use
        else raw_load<TI>(xr, x, next0, B, p.dims[0], lane);
        wgrad_accumulate<2, TI>(dw0, GA, GB, lane);
        if (dx) {
            mlp_layer<TI, 2>(w + p.wt_off[0], g, t, lane);          // dX = W0^T dZ1
            wave_sync();
            frag_to_stage<TI>(GA, lane, t);
            wave_sync();
            if (p.lm) stage_to_global_lm(GA, dx, row0, B, lane);
            else stage_to_global<TI>(GA, dx, row0, B, p.dims[0], p.magic[0], lane);
        }
        wave_sync();
    }
    mlp_dw_to_partial<NL, TI, TO>(p, dw0, dw1, dw2, lds + p.packed_floats, partial, lane, wave);
}

// ------------------------------------------------------------------------------------------
// The launches on the fp16 matrix pipe ("f16x3", field_core.hpp: every operand v = hi + lo with hi = fp16(v), lo = fp16(v - hi), a product is
// a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- 22-bit products at 1/5.3 of the fp32 pipe's cycles).
//  * Range: encoder features start at 1e-4 (gridencoder/grid.py:107), below fp16's normal range, and nothing bounds a user's activations, so every operand
//    tile is multiplied by the power of two that puts its largest magnitude (wave reduction: 4 DPP steps + 4 readlanes) into [2^14, 2^15) before it is
//    split; the weights get one such power per layer at pack time.  A fragment is carried as (stored values, integer exponent): powers of two commute
//    with the products and with ReLU, so un-scaling costs integer arithmetic on the exponent, not an instruction per value; the scaling itself rides on
//    the conversions (v_fma_mix*: fp32 multiply-add, one rounding to f16).
//  * Measured (626 k rows): 32-64-64-16 ReLU 100 -> 55 us, 35-64-15 ELU 108 -> 79 us; outputs within 2e-6 of the float64 layer loop relative to the
//    largest output, as the fp32 launch (tests/test_gpu_ops.py::test_fused_mlp_forward_backward_match_float64).
//  The backward in the same arithmetic follows below (k_mlp_bwd_h); both have their exact fp32 twins above (pnr_set_option("mlp_f16x3", 0)).
// ------------------------------------------------------------------------------------------
// The row-of-16 maximum of a register, as unsigned integers: |x| >= 0, so the integer order is the float order and NaN patterns sort above Inf.  One assembly
// statement: the integer maximum takes its DPP operand directly (lanes without a source keep their own value: the destination is the operand), and a DPP read of
// a register a vector instruction has just written needs two wait states -- as builtins every step was a move, the wait, the DPP move, a canonicalising
// v_max_f32 and the maximum.  Lane 15 of every row holds the row's maximum afterwards.
__device__ __forceinline__ uint32_t dpp_row_max_u32(uint32_t b) {
    asm("s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(b));
    return b;
}
// Every fragment is carried as (stored values, exponent e) with true = stored * 2^e, e a wave-uniform integer: powers of two commute with the
// matrix products and with ReLU, so the unscaling of a layer's accumulators costs no instruction per value -- it is integer arithmetic on e.
// biased exponent (0 = the tile is all zeros, 255: an Inf / NaN somewhere) of the largest |stored value| of a tile; wave-uniform
template <int NT>
__device__ __forceinline__ int tile_max_exp(const f32x16 (&a)[2]) {
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) m = fmaxf(m, fabsf(a[t][r]));
    uint32_t b = __float_as_uint(m);     // non-negative floats order like their bit patterns
    b = dpp_row_max_u32(b);              // lane 15 of every row of 16: the row's maximum
    uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)b, 15);
    u = max(u, (uint32_t)__builtin_amdgcn_readlane((int)b, 31));
    u = max(u, (uint32_t)__builtin_amdgcn_readlane((int)b, 47));
    u = max(u, (uint32_t)__builtin_amdgcn_readlane((int)b, 63));
    return (int)((u >> 23) & 0xffu);
}
// k with 2^k * (largest |stored|) in [2^14, 2^15) (0 for an all-zero or non-finite tile).  k is clamped HERE to what pow2i can represent (a tile
// whose largest magnitude is below 2^-113 would ask for k > 127): the exponent bookkeeping (e = -k - kw, P, eg) is integer arithmetic on this very
// value, so the scale applied by pow2i(k) and the un-scale carried in e always agree -- such a tile is merely lifted less far
__device__ __forceinline__ int split_exp(int max_biased_exp) {
    if (max_biased_exp == 0 || max_biased_exp == 255) return 0;
    const int k = 141 - max_biased_exp;      // >= -113: never below pow2i's lower clamp
    return k > 127 ? 127 : k;
}
__device__ __forceinline__ float pow2i(int k) {   // 2^k, k clamped to the normal range
    k = k < -126 ? -126 : (k > 127 ? 127 : k);
    return __uint_as_float((uint32_t)(k + 127) << 23);
}
__device__ __forceinline__ int exp_of_pow2(float v) { return (int)((__float_as_uint(v) >> 23) & 0xffu) - 127; }
// Two value pairs split in ONE assembly statement, the two pairs' instructions interleaved: v_fma_mixhi_f16 writes half a register (op_sel), and on gfx950 the
// next instruction that reads such a register needs a wait state -- as separate statements the compiler put an s_nop between them (five per eight values, 190
// issue slots per tile of the backward at one wave per SIMD); here every consumer sits two instructions behind its producer.
__device__ __forceinline__ void split_scaled4(float a0, float a1, float b0, float b1, float s, uint32_t& ha, uint32_t& hb, uint32_t& la, uint32_t& lb) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(ha), "=&v"(hb), "=&v"(la), "=&v"(lb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(s));
}
// hi = fp16(v * s), lo = fp16(v * s - hi) for a power of two s: the scaling rides on the conversions (v_fma_mix*: fp32 fma, ONE rounding to f16) -- two
// instructions per value and half
__device__ __forceinline__ void split_scaled(const f32x16& a, int half_idx, float s, h8& hi, h8& lo) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hw, lw;
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
        uint32_t h0, h1, l0, l1;
        split_scaled4(a[half_idx * 8 + j], a[half_idx * 8 + j + 1], a[half_idx * 8 + j + 2], a[half_idx * 8 + j + 3], s, h0, h1, l0, l1);
        hw[j / 2] = h0; hw[j / 2 + 1] = h1; lw[j / 2] = l0; lw[j / 2 + 1] = l1;
    }
    asm volatile("s_nop 3" : "+v"(hw), "+v"(lw));   // inline-asm VALU writes -> MFMA reads: the compiler's hazard recogniser does not see them (field_core.hpp: split8)
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}
struct HOp { h8 hi[4], lo[4]; };        // a tile of up to 64 features (or, transposed, 2 x 32 features by 2 k-blocks of 16 samples) as MFMA operands
template <int NT>
__device__ __forceinline__ void split_tiles(const f32x16 (&a)[2], int k, HOp& o) {
    const float s = pow2i(k);
#pragma unroll
    for (int t = 0; t < NT; t++) { split_scaled(a[t], 0, s, o.hi[2 * t], o.lo[2 * t]); split_scaled(a[t], 1, s, o.hi[2 * t + 1], o.lo[2 * t + 1]); }
}
// out[rt] = sum_kb W[rt][kb] . b[kb].  The launch runs one wave per SIMD, so nothing hides an LDS read but the wave's own matrix instructions: the NEXT block's
// weights are requested in front of this block's three products (a 1-deep software pipeline; the scheduling barrier behind every block keeps the compiler
// from pulling later reads further up, which blew the register budget, and from sinking this one back to its use, where every block waited ~100 cycles).
template <int NRT, int NKB>
__device__ __forceinline__ void layer_h(const unsigned char* __restrict__ slot, const HOp& b, f32x16 (&out)[2], int lane, h8 ahi, h8 alo) {
    constexpr int N = NRT * NKB;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int rt = i / NKB, kb = i % NKB;
        h8 nhi = ahi, nlo = alo;
        if (i + 1 < N) {
            const unsigned char* nblk = slot + (size_t)(i + 1) * 2048;
            nhi = *reinterpret_cast<const h8*>(nblk + lane * 16);
            nlo = *reinterpret_cast<const h8*>(nblk + 1024 + lane * 16);
        }
        if (kb == 0) out[rt] = zero16();
        out[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, b.hi[kb], out[rt], 0, 0, 0);
        out[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, b.lo[kb], out[rt], 0, 0, 0);
        out[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, b.hi[kb], out[rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ahi = nhi; alo = nlo;
    }
}
// the operand split of a layer's input and the layer: the FIRST block's weights are requested in front of the split (some 150 vector instructions), not behind it
// where the layer's first product would wait a full LDS round trip for them
template <int NT, int NRT, int NKB>
__device__ __forceinline__ void split_layer(const unsigned char* __restrict__ slot, const f32x16 (&in)[2], int k, HOp& b, f32x16 (&out)[2], int lane) {
    const h8 ahi = *reinterpret_cast<const h8*>(slot + lane * 16), alo = *reinterpret_cast<const h8*>(slot + 1024 + lane * 16);
    __builtin_amdgcn_sched_barrier(0);
    split_tiles<NT>(in, k, b);
    layer_h<NRT, NKB>(slot, b, out, lane, ahi, alo);
}
// hidden activation of a layer's accumulators (stored, e): ReLU commutes with the power of two and keeps e; ELU needs the true value (e becomes 0)
template <int NT, int ACT>
__device__ __forceinline__ void act_stored(f32x16 (&v)[2], int e) {
    if constexpr (ACT == 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) v[t] = relu16(v[t]);
    } else {
        const float u = pow2i(e);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) v[t][r] = act_fwd(v[t][r] * u, ACT);
    }
}
template <int NL, int TI, int TO, int ACT>
__global__ void __launch_bounds__(kMlpThreads) k_mlp_fwd_h(MlpPlan p, const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ x_tail, uint32_t B,
                                                           float* __restrict__ y) {
    extern __shared__ float lds[];
    const uint32_t wfloats = p.wt_off[0];   // forward slots only
    {
        const float* src = packed + mlp_f16_part_floats(p);
        for (uint32_t i = threadIdx.x * 4; i < wfloats; i += kMlpThreads * 4) *reinterpret_cast<float4*>(&lds[i]) = *reinterpret_cast<const float4*>(&src[i]);
    }
    const unsigned char* w = reinterpret_cast<const unsigned char*>(lds);
    const int kw0 = -exp_of_pow2(packed[mlp_scales_floats(p)]), kw1 = -exp_of_pow2(packed[mlp_scales_floats(p) + 1]),
              kw2 = NL == 3 ? -exp_of_pow2(packed[mlp_scales_floats(p) + 2]) : 0;       // log2 of the layers' weight scales
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = lds + wfloats + wave * kStageFloats;
    __syncthreads();
    const uint32_t nblock_tiles = (B + 32 * kMlpWaves - 1) / (32 * kMlpWaves);
    float xr[16 * TI];
    if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, (blockIdx.x * kMlpWaves + wave) * 32, B, lane);
    else raw_load<TI>(xr, x, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[0], lane);
    for (uint32_t bt = blockIdx.x; bt < nblock_tiles; bt += gridDim.x) {
        const uint32_t row0 = (bt * kMlpWaves + wave) * 32;
        if (p.lm) raw_to_stage_lm<TI>(stage, xr, x_tail, p.tail, p.tail_magic, row0, B, lane);
        else raw_to_stage<TI>(stage, xr, x, row0, B, p.dims[0], p.magic[0], lane);
        wave_sync();
        if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, ((bt + gridDim.x) * kMlpWaves + wave) * 32, B, lane);
        else raw_load<TI>(xr, x, ((bt + gridDim.x) * kMlpWaves + wave) * 32, B, p.dims[0], lane);
        f32x16 a[2], o[2];
        HOp b;
        frag_from_stage<TI>(stage, lane, a);
        int k = split_exp(tile_max_exp<TI>(a)), e;
        split_layer<TI, 2, 2 * TI>(w + (size_t)p.w_off[0] * 4, a, k, b, o, lane);
        e = -k - kw0;
        act_stored<2, ACT>(o, e);
        if (ACT != 0) e = 0;
        k = split_exp(tile_max_exp<2>(o));
        if constexpr (NL == 3) {
            split_layer<2, 2, 4>(w + (size_t)p.w_off[1] * 4, o, k, b, a, lane);
            e = e - k - kw1;
            act_stored<2, ACT>(a, e);
            if (ACT != 0) e = 0;
            k = split_exp(tile_max_exp<2>(a));
            split_layer<2, TO, 4>(w + (size_t)p.w_off[2] * 4, a, k, b, o, lane);
            const float u = pow2i(e - k - kw2);
#pragma unroll
            for (int t = 0; t < TO; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] *= u;
        } else {
            split_layer<2, TO, 4>(w + (size_t)p.w_off[1] * 4, o, k, b, a, lane);
            const float u = pow2i(e - k - kw1);
#pragma unroll
            for (int t = 0; t < TO; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] = a[t][r] * u;
        }
        wave_sync();
        frag_to_stage<TO>(stage, lane, o);
        wave_sync();
        stage_to_global<TO>(stage, y, row0, B, p.dims[NL], p.magic[NL], lane, p.out_act != 0);
        wave_sync();
    }
}

// ---- the backward on the fp16 pipe ----------------------------------------------------------------------------------------------------
// Same tile walk as k_mlp_bwd (recompute h, stage a (gradient, activation) pair sample-major in the wave's LDS tiles per layer, weight gradient with the
// SAMPLE as the k dimension, propagate through W^T), every product split-fp16.  The weight gradient's operands are read from the staged fp32 tiles
// as "lane = feature, 8 consecutive samples" (conflict-free column walks), scaled and split on the way into the matrix instruction -- no transposed
// copies are kept in registers (a first version formed them with a second set of matrix instructions and spilled; profiles/EXPERIMENTS.md).
// The launch-long sum of a layer is (stored, P) with dW = stored * 2^-P per wave: a tile's two operands are scaled by 2^kg and 2^ka with
// kg + ka - (e_g + e_a) = P, so its matrix instructions accumulate straight into the sum; kg puts the gradient side at 2^14, ka is what P leaves for the
// activation side -- anywhere below its own 2^14 is fine (fp16 keeps an absolute 2^-25 there, and a tile whose gradients are that much smaller than the
// established scale contributes correspondingly little).  Only when ka would overflow the activation side is P lowered and the stored sum rescaled.
struct DwScale { int P; bool have; };
__device__ __forceinline__ void split_scaled8(const float (&v)[8], float s, h8& hi, h8& lo) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hw, lw;
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
        uint32_t h0, h1, l0, l1;
        split_scaled4(v[j], v[j + 1], v[j + 2], v[j + 3], s, h0, h1, l0, l1);
        hw[j / 2] = h0; hw[j / 2 + 1] = h1; lw[j / 2] = l0; lw[j / 2 + 1] = l1;
    }
    asm volatile("s_nop 3" : "+v"(hw), "+v"(lw));
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}
template <int NRT, int NCT>
__device__ __forceinline__ void wgrad_lds(f32x16 (&dw)[2][2], DwScale& st, const float* __restrict__ G, int e_g, int mg, const float* __restrict__ A, int e_a, int ma, int lane,
                                          const int a_stride = kStage) {
    if (mg == 0 || ma == 0) return;     // an all-zero side: nothing to add (wave-uniform)
    const int kg = split_exp(mg), ka_top = split_exp(ma);
    int ka = st.P + e_g + e_a - kg;
    if (!st.have || ka > ka_top) {
        const int Pn = kg + (ka_top - 4) - e_g - e_a;
        if (st.have) {
            const float f = Pn - st.P < -126 ? 0.0f : pow2i(Pn - st.P);
#pragma unroll
            for (int rt = 0; rt < NRT; rt++)
#pragma unroll
                for (int ct = 0; ct < NCT; ct++)
#pragma unroll
                    for (int r = 0; r < 16; r++) dw[rt][ct][r] *= f;
        }
        st.P = Pn; st.have = true;
        ka = ka_top - 4;
    }
    const int c = lane & 31, hh = lane >> 5;
    const float sg = pow2i(kg), sa = pow2i(ka);
    // The operands of the two 16-sample halves, in the order they are needed: A[0..NCT-1], G[0..NRT-1] per half.  One wave per SIMD: an operand's eight LDS
    // reads are requested while the PREVIOUS operand is being split and multiplied (raw values of one operand ahead: 8 registers), the scheduling
    // barrier pins them there -- left alone the compiler sank them to their use and every operand waited a full LDS round trip.
    constexpr int kPerHalf = NCT + NRT, kOps = 2 * kPerHalf;
    auto raw_of = [&](int i, float (&v)[8]) {
        const int kb = i / kPerHalf, j = i % kPerHalf;
        const float* __restrict__ buf = j < NCT ? A : G;
        const int stride = j < NCT ? a_stride : kStage, col = (j < NCT ? j : j - NCT) * 32 + c;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = buf[(kb * 16 + hh * 8 + q) * stride + col];
    };
    float cur[8], nxt[8];
    raw_of(0, cur);
    h8 ahi[2], alo[2];
#pragma unroll
    for (int i = 0; i < kOps; i++) {
        const int j = i % kPerHalf;
        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));   // this operand has arrived ...
        if (i + 1 < kOps) raw_of(i + 1, nxt);                                                                                                     // ... the next one is requested
        __builtin_amdgcn_sched_barrier(0);
        if (j < NCT) {
            split_scaled8(cur, sa, ahi[j], alo[j]);
        } else {
            const int rt = j - NCT;
            h8 ghi, glo;
            split_scaled8(cur, sg, ghi, glo);
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
                dw[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(glo, ahi[ct], dw[rt][ct], 0, 0, 0);
                dw[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ghi, alo[ct], dw[rt][ct], 0, 0, 0);
                dw[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ghi, ahi[ct], dw[rt][ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
    }
}
// g (stored) *= act'(h): the exponent of g is untouched (h: the stored activation; ReLU only needs its sign, ELU stores true values)
template <int NT, int ACT>
__device__ __forceinline__ void act_grad_stored(f32x16 (&g)[2], const f32x16 (&h)[2]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if constexpr (ACT == 0) g[t][r] = h[t][r] > 0.0f ? g[t][r] : 0.0f;
            else g[t][r] *= act_grad(h[t][r], ACT);
        }
}

template <int NL, int TI, int TO, int ACT>
__global__ void __launch_bounds__(kMlpThreads) k_mlp_bwd_h(MlpPlan p, const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ x_tail, const float* __restrict__ dy,
                                                           const float* __restrict__ yout, uint32_t B, float* __restrict__ dx, float* __restrict__ partial) {
    extern __shared__ float lds[];
    {
        const float* src = packed + mlp_f16_part_floats(p);
        for (uint32_t i = threadIdx.x * 4; i < p.packed_floats; i += kMlpThreads * 4) *reinterpret_cast<float4*>(&lds[i]) = *reinterpret_cast<const float4*>(&src[i]);
    }
    const unsigned char* w = reinterpret_cast<const unsigned char*>(lds);
    const int kw0 = -exp_of_pow2(packed[mlp_scales_floats(p)]), kw1 = -exp_of_pow2(packed[mlp_scales_floats(p) + 1]),
              kw2 = NL == 3 ? -exp_of_pow2(packed[mlp_scales_floats(p) + 2]) : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* GA = lds + p.packed_floats + wave * 2 * kStageFloats;   // gradient side of the weight-gradient products
    float* GB = GA + kStageFloats;                                 // activation side
    // the X tile keeps a (narrower) tile of its own: staged once per tile, read as fragments for the recomputation and as columns for layer 0's weight
    // gradient; the next tile's X is requested right behind the staging and its registers are free for the whole tile (k_mlp_bwd stages X twice and holds them)
    constexpr int kXStage = 32 * TI + 1;
    float* GX = lds + p.packed_floats + kMlpWaves * 2 * kStageFloats + wave * 33 * kXStage;
    __syncthreads();
    f32x16 dw0[2][2], dw1[2][2], dw2[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) { dw0[i][j] = zero16(); dw1[i][j] = zero16(); dw2[i][j] = zero16(); }
    DwScale st0{0, false}, st1{0, false}, st2{0, false};
    const uint32_t nblock_tiles = (B + 32 * kMlpWaves - 1) / (32 * kMlpWaves);
    float xr[16 * TI], yn[16 * TO], sn[16 * TO];
    if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, (blockIdx.x * kMlpWaves + wave) * 32, B, lane);
    else raw_load<TI>(xr, x, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[0], lane);
    raw_load<TO>(yn, dy, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[NL], lane);
    if (yout) raw_load<TO>(sn, yout, (blockIdx.x * kMlpWaves + wave) * 32, B, p.dims[NL], lane);
    for (uint32_t bt = blockIdx.x; bt < nblock_tiles; bt += gridDim.x) {
        const uint32_t row0 = (bt * kMlpWaves + wave) * 32;
        if (p.lm) raw_to_stage_lm<TI>(GX, xr, x_tail, p.tail, p.tail_magic, row0, B, lane, kXStage);
        else raw_to_stage<TI>(GX, xr, x, row0, B, p.dims[0], p.magic[0], lane, kXStage);
        if (yout) {   // dZ = dY (1 - y) y: sigmoid_backward, in the raw layout both tiles share
#pragma unroll
            for (int k = 0; k < 16 * TO; k++) yn[k] = (yn[k] * (1.0f - sn[k])) * sn[k];
        }
        raw_to_stage<TO>(GA, yn, dy, row0, B, p.dims[NL], p.magic[NL], lane, kStage, yout);
        wave_sync();
        const uint32_t next0 = ((bt + gridDim.x) * kMlpWaves + wave) * 32;
        f32x16 h1[2], h2[2], g[2], t[2];
        HOp b;
        // forward again: x -> h1 (-> h2)
        frag_from_stage<TI>(GX, lane, t, kXStage);
        const int mx = tile_max_exp<TI>(t), kx = split_exp(mx);
        split_layer<TI, 2, 2 * TI>(w + (size_t)p.w_off[0] * 4, t, kx, b, h1, lane);
        const int e1s = -kx - kw0;
        act_stored<2, ACT>(h1, e1s);
        const int e1 = ACT != 0 ? 0 : e1s;
        const int m1 = tile_max_exp<2>(h1);
        int mg, kg, eg = 0;
        if constexpr (NL == 3) {
            const int k1 = split_exp(m1);
            split_layer<2, 2, 4>(w + (size_t)p.w_off[1] * 4, h1, k1, b, h2, lane);
            const int e2s = e1 - k1 - kw1;
            act_stored<2, ACT>(h2, e2s);
            const int e2 = ACT != 0 ? 0 : e2s;
            const int m2 = tile_max_exp<2>(h2);
            frag_from_stage<TO>(GA, lane, g);
            mg = tile_max_exp<TO>(g); kg = split_exp(mg);
            // layer 2: GA = dY, GB <- h2
            wave_sync();
            frag_to_stage<2>(GB, lane, h2);
            wave_sync();
            wgrad_lds<TO, 2>(dw2, st2, GA, 0, mg, GB, e2, m2, lane);
            split_layer<TO, 2, 2 * TO>(w + (size_t)p.wt_off[2] * 4, g, kg, b, t, lane);          // dH2 = W2^T dY
            eg = -kg - kw2;
            act_grad_stored<2, ACT>(t, h2);
            mg = tile_max_exp<2>(t); kg = split_exp(mg);
            // h1 again (from the X tile, which is still staged): 12-24 matrix instructions against 32 registers that would otherwise sit through layer 2
            {
                f32x16 xin[2];
                HOp bx;
                frag_from_stage<TI>(GX, lane, xin, kXStage);
                split_layer<TI, 2, 2 * TI>(w + (size_t)p.w_off[0] * 4, xin, kx, bx, h1, lane);
                act_stored<2, ACT>(h1, e1s);
            }
            // layer 1: GA <- dZ2, GB <- h1
            wave_sync();
            frag_to_stage<2>(GA, lane, t);
            frag_to_stage<2>(GB, lane, h1);
            wave_sync();
            wgrad_lds<2, 2>(dw1, st1, GA, eg, mg, GB, e1, m1, lane);
            split_layer<2, 2, 4>(w + (size_t)p.wt_off[1] * 4, t, kg, b, g, lane);               // dH1 = W1^T dZ2
            eg = eg - kg - kw1;
            act_grad_stored<2, ACT>(g, h1);
        } else {
            frag_from_stage<TO>(GA, lane, g);
            mg = tile_max_exp<TO>(g); kg = split_exp(mg);
            // layer 1 (the last): GA = dY, GB <- h1
            wave_sync();
            frag_to_stage<2>(GB, lane, h1);
            wave_sync();
            wgrad_lds<TO, 2>(dw1, st1, GA, 0, mg, GB, e1, m1, lane);
            split_layer<TO, 2, 2 * TO>(w + (size_t)p.wt_off[1] * 4, g, kg, b, t, lane);          // dH1 = W1^T dY
            eg = -kg - kw1;
            act_grad_stored<2, ACT>(t, h1);
            g[0] = t[0]; g[1] = t[1];
        }
        // the next tile's dY and X: requested here, a layer and the dX store ahead of their use (earlier, their 32-48 registers sit through the tile's
        // register-tightest stretch)
        raw_load<TO>(yn, dy, next0, B, p.dims[NL], lane);
        if (yout) raw_load<TO>(sn, yout, next0, B, p.dims[NL], lane);
        if (p.lm) raw_load_lm<TI>(xr, x, x_tail, p.tail, next0, B, lane);
        else raw_load<TI>(xr, x, next0, B, p.dims[0], lane);
        // layer 0: GA <- dZ1, the activation side is the X tile
        mg = tile_max_exp<2>(g); kg = split_exp(mg);
        wave_sync();
        frag_to_stage<2>(GA, lane, g);
        wave_sync();
        wgrad_lds<2, TI>(dw0, st0, GA, eg, mg, GX, 0, mx, lane, kXStage);
        if (dx) {
            split_layer<2, TI, 4>(w + (size_t)p.wt_off[0] * 4, g, kg, b, t, lane);              // dX = W0^T dZ1
            const float u = pow2i(eg - kg - kw0);
#pragma unroll
            for (int tt = 0; tt < TI; tt++)
#pragma unroll
                for (int r = 0; r < 16; r++) t[tt][r] *= u;
            wave_sync();
            frag_to_stage<TI>(GA, lane, t);
            wave_sync();
            if (p.lm) stage_to_global_lm(GA, dx, row0, B, lane);
            else stage_to_global<TI>(GA, dx, row0, B, p.dims[0], p.magic[0], lane);
        }
        wave_sync();
    }
    {   // back to true units: dW = stored * 2^-P (per wave, per layer)
        const float f0 = st0.have ? pow2i(-st0.P) : 1.0f, f1 = st1.have ? pow2i(-st1.P) : 1.0f, f2 = st2.have ? pow2i(-st2.P) : 1.0f;
#pragma unroll
        for (int l = 0; l < NL; l++)     // (only the tiles a layer has: touching the others would keep 64 more accumulator registers alive through the loop)
#pragma unroll
            for (int i = 0; i < tiles_at<NL, TI, TO>(l + 1); i++)
#pragma unroll
                for (int j = 0; j < tiles_at<NL, TI, TO>(l); j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (l == 0) dw0[i][j][r] *= f0;
                        else if (l == 1) dw1[i][j][r] *= f1;
                        else dw2[i][j][r] *= f2;
                    }
    }
    mlp_dw_to_partial<NL, TI, TO>(p, dw0, dw1, dw2, lds + p.packed_floats, partial, lane, wave);
}

// dw[e] = sum over the workgroup partials in a fixed order (8 groups of 32 columns per workgroup)
__global__ void __launch_bounds__(256) k_mlp_dw_reduce(const float* __restrict__ partial, uint32_t nparts, MlpPlan p, MlpGrads gr) {
    __shared__ float red[8][32];
    const uint32_t c = threadIdx.x & 31u, g = threadIdx.x >> 5;
    const uint32_t e = blockIdx.x * 32 + c;
    float s = 0.0f;
    if (e < p.dw_floats)
        for (uint32_t q = g; q < nparts; q += 8) s += partial[(size_t)q * p.dw_floats + e];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && e < p.dw_floats) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; k++) t += red[k][c];
        for (uint32_t l = 0; l < p.n_layers; l++) {
            const uint32_t n = p.dims[l] * p.dims[l + 1];
            if (e >= p.dw_off[l] && e < p.dw_off[l] + n && gr.dw[l]) gr.dw[l][e - p.dw_off[l]] = t;
        }
    }
}

static uint32_t mlp_blocks(uint32_t B) {
    const uint32_t t = cdiv(B, 32 * kMlpWaves);
    return t < kMlpMaxBlocks ? (t ? t : 1) : kMlpMaxBlocks;
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_mlp_packed_bytes(const pnr_mlp_desc* desc) {
    MlpPlan p;
    return make_plan(desc, p) ? ((uint64_t)p.packed_floats * 2 + 4) * 4 : 0;
}

int pnr_mlp_pack(const pnr_mlp_desc* desc, const float* w0, const float* w1, const float* w2, float* packed, pnr_stream_t stream) {
    MlpPlan p;
    if (!make_plan(desc, p)) return PNR_ERR_UNSUPPORTED;
    if (!w0 || !w1 || (p.n_layers == 3 && !w2) || !packed) return PNR_ERR_INVALID;
    MlpWeights ws{{w0, w1, w2}};
    hipLaunchKernelGGL(k_mlp_pack, dim3(2 * p.n_layers, kPackChunks), dim3(256), 0, as_stream(stream), p, ws, packed);
    return check_launch();
}

#define PNR_MLP_DISPATCH(KERNEL, NLV, TIV, TOV, ...)                                                                                           \
    do {                                                                                                                                       \
        if (p.act == 0) {                                                                                                                      \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<NLV, TIV, TOV, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                    (int)lds) != hipSuccess)                                                                                   \
                return PNR_ERR_LAUNCH;                                                                                                         \
            hipLaunchKernelGGL((KERNEL<NLV, TIV, TOV, 0>), dim3(grid), dim3(kMlpThreads), lds, s, __VA_ARGS__);                                \
        } else {                                                                                                                               \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<NLV, TIV, TOV, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                    (int)lds) != hipSuccess)                                                                                   \
                return PNR_ERR_LAUNCH;                                                                                                         \
            hipLaunchKernelGGL((KERNEL<NLV, TIV, TOV, 1>), dim3(grid), dim3(kMlpThreads), lds, s, __VA_ARGS__);                                \
        }                                                                                                                                      \
    } while (0)
#ifdef PNR_MLP_MINI   /* compile-time experiments only: one instantiation, so that a resource-usage build takes seconds */
#define PNR_MLP_SWITCH(KERNEL, ...) do { hipLaunchKernelGGL((KERNEL<3, 1, 1, 0>), dim3(grid), dim3(kMlpThreads), lds, s, __VA_ARGS__); } while (0)
#else
#define PNR_MLP_SWITCH(KERNEL, ...)                                                                   \
    do {                                                                                              \
        const int ti = (int)tiles32(p.dims[0]), to = (int)tiles32(p.dims[p.n_layers]);                \
        if (p.n_layers == 2) {                                                                        \
            if (ti == 1 && to == 1) PNR_MLP_DISPATCH(KERNEL, 2, 1, 1, __VA_ARGS__);                   \
            else if (ti == 1) PNR_MLP_DISPATCH(KERNEL, 2, 1, 2, __VA_ARGS__);                         \
            else if (to == 1) PNR_MLP_DISPATCH(KERNEL, 2, 2, 1, __VA_ARGS__);                         \
            else PNR_MLP_DISPATCH(KERNEL, 2, 2, 2, __VA_ARGS__);                                      \
        } else {                                                                                      \
            if (ti == 1 && to == 1) PNR_MLP_DISPATCH(KERNEL, 3, 1, 1, __VA_ARGS__);                   \
            else if (ti == 1) PNR_MLP_DISPATCH(KERNEL, 3, 1, 2, __VA_ARGS__);                         \
            else if (to == 1) PNR_MLP_DISPATCH(KERNEL, 3, 2, 1, __VA_ARGS__);                         \
            else PNR_MLP_DISPATCH(KERNEL, 3, 2, 2, __VA_ARGS__);                                      \
        }                                                                                             \
    } while (0)
#endif

// lm_levels: 0 = x is row-major [B, dims[0]]; 16 = x is a level-major encoder output [16][B][2] followed by x_tail [B, dims[0] - 32]
// The launches move their tiles as 16-byte requests: every activation array must start on a 16-byte boundary (torch allocations do; a view that starts inside one may
// not -- palettenerf_amd.mlp copies those) and hold at least four floats.
static bool arrays_ok(std::initializer_list<const void*> ptrs, uint32_t B, const MlpPlan& p) {
    for (const void* q : ptrs) if (q && (reinterpret_cast<uintptr_t>(q) & 15u) != 0) return false;
    uint32_t wmin = p.dims[0] < p.dims[p.n_layers] ? p.dims[0] : p.dims[p.n_layers];
    if (p.lm && p.tail && p.tail < wmin) wmin = p.tail;
    return (uint64_t)B * wmin >= 4;
}

static int plan_sources(MlpPlan& p, uint32_t lm_levels, const float* x_tail) {
    if (lm_levels == 0) return PNR_OK;
    if (lm_levels != 16 || p.dims[0] < 32) return PNR_ERR_UNSUPPORTED;
    p.lm = 1;
    p.tail = p.dims[0] - 32;
    if (p.tail && !x_tail) return PNR_ERR_INVALID;
    p.tail_magic = p.tail ? (uint32_t)((1ull << 32) / p.tail) + 1u : 1u;
    return PNR_OK;
}

static int mlp_forward_impl(const pnr_mlp_desc* desc, const float* packed, const float* x, uint32_t lm_levels, const float* x_tail, uint32_t B, float* y,
                            pnr_stream_t stream) {
    MlpPlan p;
    if (!make_plan(desc, p)) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!packed || !x || !y) return PNR_ERR_INVALID;
    if ((uint64_t)B * 64 >= (1ull << 32)) return PNR_ERR_UNSUPPORTED;   // element indices are 32-bit
    if (int rc = plan_sources(p, lm_levels, x_tail)) return rc;
    const size_t lds = ((size_t)p.wt_off[0] + kMlpWaves * kStageFloats) * 4;
    if (lds > 160 * 1024) return PNR_ERR_UNSUPPORTED;
    if (!arrays_ok({x, x_tail, y}, B, p)) return PNR_ERR_ALIGNMENT;
    hipStream_t s = as_stream(stream);
    const uint32_t tiles = cdiv(B, 32 * kMlpWaves), grid = tiles < 2 * kMlpMaxBlocks ? tiles : 2 * kMlpMaxBlocks;   // two workgroups per CU
    if (g_opt_mlp_f16x3) PNR_MLP_SWITCH(k_mlp_fwd_h, p, packed, x, x_tail, B, y);
    else PNR_MLP_SWITCH(k_mlp_fwd, p, packed, x, x_tail, B, y);
    return check_launch();
}

int pnr_mlp_forward(const pnr_mlp_desc* desc, const float* packed, const float* x, uint32_t B, float* y, pnr_stream_t stream) {
    return mlp_forward_impl(desc, packed, x, 0, nullptr, B, y, stream);
}
int pnr_mlp_forward_lm(const pnr_mlp_desc* desc, const float* packed, const float* enc_level_major, uint32_t levels, const float* x_tail, uint32_t B, float* y,
                       pnr_stream_t stream) {
    return mlp_forward_impl(desc, packed, enc_level_major, levels, x_tail, B, y, stream);
}

uint64_t pnr_mlp_backward_workspace_bytes(const pnr_mlp_desc* desc, uint32_t B) {
    MlpPlan p;
    if (!make_plan(desc, p)) return 0;
    return (uint64_t)mlp_blocks(B) * p.dw_floats * 4;
}

static int mlp_backward_impl(const pnr_mlp_desc* desc, const float* packed, const float* x, uint32_t lm_levels, const float* x_tail, const float* y, const float* dy,
                             uint32_t B, float* dx, float* dw0, float* dw1, float* dw2, void* workspace, uint64_t workspace_bytes, pnr_stream_t stream) {
    MlpPlan p;
    if (!make_plan(desc, p)) return PNR_ERR_UNSUPPORTED;
    if (int rc = plan_sources(p, lm_levels, x_tail)) return rc;
    hipStream_t s = as_stream(stream);
    MlpGrads gr{{dw0, dw1, dw2}};
    if (B == 0) {
        for (uint32_t l = 0; l < p.n_layers; l++)
            if (gr.dw[l] && hipMemsetAsync(gr.dw[l], 0, (size_t)p.dims[l] * p.dims[l + 1] * 4, s) != hipSuccess) return PNR_ERR_LAUNCH;
        return PNR_OK;
    }
    if (!packed || !x || !dy || !workspace || (p.out_act && !y)) return PNR_ERR_INVALID;
    if (!p.out_act) y = nullptr;
    if ((uint64_t)B * 64 >= (1ull << 32)) return PNR_ERR_UNSUPPORTED;   // element indices are 32-bit
    if (workspace_bytes < pnr_mlp_backward_workspace_bytes(desc, B)) return PNR_ERR_INVALID;
    const size_t lds = ((size_t)p.packed_floats + kMlpWaves * 2 * kStageFloats) * 4;
    if (lds > 160 * 1024 || p.dw_floats > (uint32_t)(kMlpWaves * 2 * kStageFloats)) return PNR_ERR_UNSUPPORTED;
    const uint32_t blocks = mlp_blocks(B), grid = blocks;
    float* partial = static_cast<float*>(workspace);
    if (!arrays_ok({x, x_tail, y, dy, dx}, B, p)) return PNR_ERR_ALIGNMENT;
    // the split-fp16 backward where it holds its tile without (much) scratch: two layers unless both ends are 64 wide, three layers with 32-wide ends --
    // every stack of both fields; the wider instantiations spill twice what the fp32 ones do and stay on those
    const uint32_t ti = tiles32(p.dims[0]), to = tiles32(p.dims[p.n_layers]);
    const bool h_fits = p.n_layers == 2 ? !(ti == 2 && to == 2) : (ti == 1 && to == 1);
    const size_t lds_h = lds + (size_t)kMlpWaves * 33 * (32 * ti + 1) * 4;       // + the X tiles of k_mlp_bwd_h
    if (g_opt_mlp_f16x3 && h_fits && lds_h <= 160 * 1024) {
        const size_t lds_fp32 = lds;
        (void)lds_fp32;
        { const size_t lds = lds_h; PNR_MLP_SWITCH(k_mlp_bwd_h, p, packed, x, x_tail, dy, y, B, dx, partial); }
    } else PNR_MLP_SWITCH(k_mlp_bwd, p, packed, x, x_tail, dy, y, B, dx, partial);
    hipLaunchKernelGGL(k_mlp_dw_reduce, dim3(cdiv(p.dw_floats, 32)), dim3(256), 0, s, partial, grid, p, gr);
    return check_launch();
}

int pnr_mlp_backward(const pnr_mlp_desc* desc, const float* packed, const float* x, const float* y, const float* dy, uint32_t B, float* dx, float* dw0,
                     float* dw1, float* dw2, void* workspace, uint64_t workspace_bytes, pnr_stream_t stream) {
    return mlp_backward_impl(desc, packed, x, 0, nullptr, y, dy, B, dx, dw0, dw1, dw2, workspace, workspace_bytes, stream);
}
int pnr_mlp_backward_lm(const pnr_mlp_desc* desc, const float* packed, const float* enc_level_major, uint32_t levels, const float* x_tail, const float* dy,
                        uint32_t B, float* denc_level_major, float* dw0, float* dw1, float* dw2, void* workspace, uint64_t workspace_bytes,
                        pnr_stream_t stream) {
    if (desc && (desc->activation & PNR_MLP_OUT_SIGMOID)) return PNR_ERR_UNSUPPORTED;   // no stack behind an encoder ends in a sigmoid
    return mlp_backward_impl(desc, packed, enc_level_major, levels, x_tail, nullptr, dy, B, denc_level_major, dw0, dw1, dw2, workspace, workspace_bytes, stream);
}

}  // extern "C"
