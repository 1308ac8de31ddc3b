// gridencoder.hip -- multiresolution hash / tiled grid encoder for gfx950 (MI355X).
//
// Computes what the reference's gridencoder extension computes (cited per kernel).  Host-side
// differences: the per-level scale/resolution are evaluated once on the host (exp2f of the same
// libm the oracle uses) and passed by value, so no transcendental runs per thread and the cell a
// sample falls into cannot depend on the device's exp2 implementation.
//
// HBM layout: table [sum_l T_l, C] row-major (fp32 or fp16), offsets int32[L+1], inputs [B,D] fp32
// in [0,1], outputs [L,B,C] (level-major: a block column works on ONE level so that level's
// <= 4 MiB table stays resident in the XCD L2s while the samples stream through).
#include "pnr_common.hpp"
#include "grid_core.hpp"
#include <type_traits>

namespace pnr {

template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.0f; }
template <> __device__ __forceinline__ __half zero_of<__half>() { return __float2half(0.0f); }

// reference gridencoder.cu:75-223  kernel_grid
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_fwd(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                  const int32_t* __restrict__ offsets, T* __restrict__ outputs, uint32_t B, uint32_t L,
                                                  LevelParams lp, T* __restrict__ dy_dx, uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    grid += (size_t)off0 * C;
    inputs += (size_t)b * D;
    outputs += ((size_t)level * B + b) * C;
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];

    float in[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) { in[d] = inputs[d]; oob |= (in[d] < 0.0f) | (in[d] > 1.0f); }
    if (oob) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = zero_of<T>();
        if (dy_dx) {
            T* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dd[i] = zero_of<T>();
        }
        return;
    }

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }

    // issue all 2^D gathers first (independent loads in flight), then reduce in the reference's order
    uint32_t idxs[1u << D];
    float ws[1u << D];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        ws[idx] = w;
        idxs[idx] = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
    }
    T acc[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) acc[ch] = zero_of<T>();
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) corner_accumulate<C>(acc, ws[idx], grid + idxs[idx]);

    if constexpr (sizeof(T) == 4 && C == 2) {
        *reinterpret_cast<float2*>(outputs) = make_float2(acc[0], acc[1]);
    } else if constexpr (sizeof(T) == 2 && C == 2) {
        *reinterpret_cast<__half2*>(outputs) = *reinterpret_cast<__half2*>(acc);
    } else {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = acc[ch];
    }

    if (dy_dx) {  // reference gridencoder.cu:179-222
        T* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float ga[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) ga[ch] = 0.0f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                pl[gd] = pg[gd];
                const uint32_t il = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
                pl[gd] = pg[gd] + 1;
                const uint32_t ir = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    if constexpr (sizeof(T) == 4) ga[ch] = fmaf(w, to_f32(grid[ir + ch]) - to_f32(grid[il + ch]), ga[ch]);
                    else ga[ch] = __half2float(__float2half(ga[ch] + __half2float(__float2half(w * __half2float(__hsub(grid[ir + ch], grid[il + ch]))))));
                }
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                if constexpr (sizeof(T) == 4) dd[gd * C + ch] = ga[ch];
                else dd[gd * C + ch] = __float2half(ga[ch]);
            }
        }
    }
}

// The lookup every shipped configuration issues (D = 3, C = 2, no dy_dx; fp32 or fp16 table): the same cell, corner order and fmaf / half
// chains as k_grid_fwd<T, 3, 2> above -- bit for bit -- with what the frame loop's lookup (frame.hip: grid_row / gather8) found:
//   * all eight row addresses exist before the first load and every load returns into registers of its own (load8_fresh);
//   * the finest levels -- scattered rows, five times the time of a dense level -- are dispatched first (blockIdx.y = 0 is level L-1),
//     the dense ones fill the launch's tail;
//   * the row index is formed for the kind of level the block works on (block-uniform): dense (the index is below the level's size, no
//     modulo), hashed with a power-of-two size (a mask), anything else (the reference's `%`);
//   * ROWMAJOR: the sample-major [B, L*C] row GridEncoder.forward returns (gridencoder/grid.py:57) is written here, 8 bytes per (sample,
//     level) -- the reference and k_grid_fwd write [L, B, C] and leave the permute-copy to the caller.
template <typename T, bool ROWMAJOR, int NT = 0>   // NT (experiment knob, pnr_set_option "grid_nt"): bit 0 = non-temporal output stores, bit 1 = non-temporal input loads
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_grid_fwd_d3c2(const float* __restrict__ inputs, const T* __restrict__ grid, const int32_t* __restrict__ offsets, T* __restrict__ outputs, uint32_t B,
                uint32_t L, LevelParams lp, uint32_t gridtype, bool align_corners) {
    typedef typename std::conditional<sizeof(T) == 4, f32x2, uint32_t>::type RowT;   // one table row: float2 or half2
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = L - 1u - blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];
    const RowT* tab = reinterpret_cast<const RowT*>(grid) + off0;
    RowT* out = reinterpret_cast<RowT*>(outputs) + (ROWMAJOR ? (size_t)b * L + level : (size_t)level * B + b);

    float in[3];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        in[d] = (NT & 2) ? __builtin_nontemporal_load(inputs + (size_t)b * 3 + d) : inputs[(size_t)b * 3 + d];
        oob |= (in[d] < 0.0f) | (in[d] > 1.0f);
    }
    if (oob) {
        if constexpr (sizeof(T) == 4) *out = RowT{0.0f, 0.0f}; else *out = 0u;
        return;
    }
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    // get_grid_index (gridencoder.cu:49-72) per kind of level; all three forms give the generic form's index
    const uint32_t side = align_corners ? resolution : resolution + 1u;
    // the reference multiplies a uint32 stride up dimension by dimension while it is <= hashmap_size and hashes iff it ends up larger
    uint32_t stride = 1u;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++)
        if (stride <= hashmap_size) stride *= side;
    const bool hashed = gridtype == 0u && stride > hashmap_size;
    // (64-bit: implies every per-dimension test above and no uint32 wrap.)  Not with align_corners: there side == resolution and an input of exactly
    // 1.0 gives pg + 1 == side, an index that may reach side^3 >= hashmap_size -- the reference wraps it with `%` (gridencoder.cu:49-72), so those
    // levels take the generic form below
    const bool dense = !align_corners && (uint64_t)side * side * side <= (uint64_t)hashmap_size;
    uint32_t idxs[8];
    if (dense) {                                                       // index < side^3 <= size: the reference's `%` is the identity
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
            idxs[i] = (pg[0] + (i & 1u)) + (pg[1] + ((i >> 1) & 1u)) * side + (pg[2] + ((i >> 2) & 1u)) * side * side;
    } else if (hashed && (hashmap_size & (hashmap_size - 1u)) == 0u) {   // hashed, power-of-two size
        const uint32_t mask = hashmap_size - 1u;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
            idxs[i] = ((pg[0] + (i & 1u)) ^ ((pg[1] + ((i >> 1) & 1u)) * 2654435761u) ^ ((pg[2] + ((i >> 2) & 1u)) * 805459861u)) & mask;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t pl[3] = {pg[0] + (i & 1u), pg[1] + ((i >> 1) & 1u), pg[2] + ((i >> 2) & 1u)};
            idxs[i] = grid_index<3, 1>(gridtype, align_corners, hashmap_size, resolution, pl);
        }
    }
    const RowT* p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = tab + idxs[i];
    RowT v[8];
    load8_fresh(p, v);
    float ws[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {   // the reference's order of multiplications (gridencoder.cu:150-163)
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) w *= (i & (1u << d)) ? pos[d] : 1.0f - pos[d];
        ws[i] = w;
    }
    if constexpr (sizeof(T) == 4) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) { a0 = fmaf(ws[i], v[i].x, a0); a1 = fmaf(ws[i], v[i].y, a1); }
        if constexpr (NT & 1) __builtin_nontemporal_store(RowT{a0, a1}, out); else *out = RowT{a0, a1};
    } else {
        __half acc[2] = {__float2half(0.0f), __float2half(0.0f)};
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            __half hv[2];
            __builtin_memcpy(hv, &v[i], 4);
#pragma unroll
            for (int ch = 0; ch < 2; ch++)   // corner_accumulate<__half>: addend and sum rounded to fp16
                acc[ch] = __float2half(__half2float(acc[ch]) + __half2float(__float2half(ws[i] * __half2float(hv[ch]))));
        }
        uint32_t o;
        __builtin_memcpy(&o, acc, 4);
        *out = o;
    }
}

// Two tables of the same geometry looked up at the same points in one pass (PaletteNeRF training: `encoder` for the frozen density and `encoder_palette`
// for the colour basis, palette/network.py:156-262).  `pair` holds the two tables interleaved row by row, [row][table][2]: one 16-byte gather serves both
// (the inference loop's k_frame_grid_pair does the same: 1.33x the time of one lookup instead of 2x).  Each output is bit for bit what
// k_grid_fwd_d3c2 gives for its table (same cell, corner order and fmaf chains).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
k_grid_fwd_d3c2_pair(const float* __restrict__ inputs, const f32x4* __restrict__ pair, const int32_t* __restrict__ offsets, float* __restrict__ out0,
                     float* __restrict__ out1, uint32_t B, uint32_t L, LevelParams lp, uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = L - 1u - blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];
    const f32x4* tab = pair + off0;
    const size_t o = ((size_t)level * B + b) * 2;
    float in[3];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        in[d] = inputs[(size_t)b * 3 + d];
        oob |= (in[d] < 0.0f) | (in[d] > 1.0f);
    }
    if (oob) {
        *reinterpret_cast<f32x2*>(out0 + o) = f32x2{0.0f, 0.0f};
        *reinterpret_cast<f32x2*>(out1 + o) = f32x2{0.0f, 0.0f};
        return;
    }
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    uint32_t idxs[8];
    if (align_corners) {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t pl[3] = {pg[0] + (i & 1u), pg[1] + ((i >> 1) & 1u), pg[2] + ((i >> 2) & 1u)};
            idxs[i] = grid_index<3, 1>(gridtype, true, hashmap_size, resolution, pl);
        }
    } else {
        corner_rows_by_kind<1>(level_kind(gridtype, hashmap_size, resolution), gridtype, hashmap_size, resolution, pg, idxs);
    }
    const f32x4* p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = tab + idxs[i];
    f32x4 v[8];
    load8_fresh(p, v);
    float a0 = 0.0f, a1 = 0.0f, c0 = 0.0f, c1 = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {   // the reference's order of multiplications (gridencoder.cu:150-163)
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) w *= (i & (1u << d)) ? pos[d] : 1.0f - pos[d];
        a0 = fmaf(w, v[i].x, a0); a1 = fmaf(w, v[i].y, a1);
        c0 = fmaf(w, v[i].z, c0); c1 = fmaf(w, v[i].w, c1);
    }
    *reinterpret_cast<f32x2*>(out0 + o) = f32x2{a0, a1};
    *reinterpret_cast<f32x2*>(out1 + o) = f32x2{c0, c1};
}

// reference gridencoder.cu:226-313  kernel_grid_backward.  One thread scatters all C channels of
// one (sample, level) with hardware fp32 / packed-fp16 atomics (global_atomic_add_f32 /
// global_atomic_pk_add_f16) -- no CAS loops.
//
// COMBINE (coarse levels, fp32): samples arrive ordered along rays, so on a coarse level long RUNS of consecutive
// lanes scatter into the very same table rows (25 consecutive samples per cell on level 0) and the L2 serialises
// same-address atomics (11.8 ms for level 0 alone on a 627 k-sample training batch).  Per corner the wave does a
// segmented sum over runs of equal row index (ballot of run heads + 6 shuffle steps) and only the last lane of
// a run issues the atomic: same sums up to rounding order, ~25x fewer atomics where it matters.
template <typename T, uint32_t D, uint32_t C, bool COMBINE>
__global__ void __launch_bounds__(256) k_grid_bwd(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                  const int32_t* __restrict__ offsets, T* __restrict__ grad_grid, uint32_t B, uint32_t L,
                                                  LevelParams lp, uint32_t gridtype, bool align_corners, uint32_t level0) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (!COMBINE && b >= B) return;
    const uint32_t level = level0 + blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    grad_grid += (size_t)off0 * C;
    const float scale = lp.scale[level];
    const uint32_t resolution = lp.resolution[level];

    bool active = b < B;
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float v = active ? inputs[(size_t)b * D + d] : 0.0f;
        if (v < 0.0f || v > 1.0f) active = false;  // grad_grid is zero-initialised by the caller
        pos[d] = fmaf(v, scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    if (!COMBINE && !active) return;
    T g[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) g[ch] = active ? grad[((size_t)level * B + b) * C + ch] : T(0);

    const int lane = threadIdx.x & (PNR_WAVE - 1);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
        if constexpr (COMBINE && (sizeof(T) == 4 || C % 2 == 0)) {
            const uint32_t key = active ? index : 0xFFFFFFFFu;
            const uint32_t prev = __shfl_up(key, 1, PNR_WAVE);
            const bool head = lane == 0 || prev != key;
            const unsigned long long heads = __ballot(head);
            const int start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));   // first lane of this lane's run
            const bool tail = lane == PNR_WAVE - 1 || ((heads >> (lane + 1)) & 1ull);
            float acc[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) acc[ch] = active ? w * to_f32(g[ch]) : 0.0f;
#pragma unroll
            for (int off = 1; off < PNR_WAVE; off <<= 1) {
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float up = __shfl_up(acc[ch], off, PNR_WAVE);
                    if (lane - off >= start) acc[ch] += up;
                }
            }
            if (active && tail) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(reinterpret_cast<float*>(grad_grid) + index + ch, acc[ch]);
                } else {  // fp16 table: the run sum is formed in fp32 and rounded once (the reference rounds every addend)
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ch += 2)
                        unsafeAtomicAdd(reinterpret_cast<__half2*>(reinterpret_cast<__half*>(grad_grid) + index + ch),
                                        __halves2half2(__float2half(acc[ch]), __float2half(acc[ch + 1])));
                }
            }
        } else if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(reinterpret_cast<float*>(grad_grid) + index + ch, w * to_f32(g[ch]));
        } else if constexpr (C % 2 == 0) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch += 2) {
                const __half2 v = __halves2half2(__float2half(w * __half2float(g[ch])), __float2half(w * __half2float(g[ch + 1])));
                unsafeAtomicAdd(reinterpret_cast<__half2*>(reinterpret_cast<__half*>(grad_grid) + index + ch), v);
            }
        } else {
            // C == 1 with an fp16 table: the reference never takes this path either (grid.py:38 keeps
            // fp32 when C is odd); fall back to a 32-bit CAS on the containing word.
            __half* addr = reinterpret_cast<__half*>(grad_grid) + index;
            unsigned int* word = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(addr) & ~uintptr_t(3));
            const bool hi = reinterpret_cast<uintptr_t>(addr) & 2;
            unsigned int old = *word, assumed;
            do {
                assumed = old;
                __half2 cur = *reinterpret_cast<__half2*>(&assumed);
                const __half add = __float2half(w * __half2float(g[0]));
                if (hi) cur = __halves2half2(__low2half(cur), __hadd(__high2half(cur), add));
                else cur = __halves2half2(__hadd(__low2half(cur), add), __high2half(cur));
                old = atomicCAS(word, assumed, *reinterpret_cast<unsigned int*>(&cur));
            } while (assumed != old);
        }
    }
}

// reference gridencoder.cu:316-342  kernel_input_backward
template <typename T>
__global__ void __launch_bounds__(256) k_grid_input_bwd(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++) {
            const float a = to_f32(grad[((size_t)l * B + b) * C + ch]), v = to_f32(dd[l * D * C + d * C + ch]);
            if constexpr (sizeof(T) == 4) r = fmaf(a, v, r);
            else r = __half2float(__float2half(r + __half2float(__float2half(a * v))));
        }
    if constexpr (sizeof(T) == 4) grad_inputs[t] = r; else grad_inputs[t] = __float2half(r);
}

template <typename T, uint32_t D>
static int launch_fwd_c(const float* inputs, const T* emb, const int32_t* offsets, T* outputs, uint32_t B, uint32_t C, uint32_t L,
                        const LevelParams& lp, T* dy_dx, uint32_t gridtype, bool ac, hipStream_t s) {
    const dim3 grid(cdiv(B, 256), L), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_fwd<T, D, 1>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, dy_dx, gridtype, ac); break;
        case 2: hipLaunchKernelGGL((k_grid_fwd<T, D, 2>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, dy_dx, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((k_grid_fwd<T, D, 4>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, dy_dx, gridtype, ac); break;
        case 8: hipLaunchKernelGGL((k_grid_fwd<T, D, 8>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, dy_dx, gridtype, ac); break;
        default: return PNR_ERR_UNSUPPORTED;  // "GridEncoding: C must be 1, 2, 4, or 8." (gridencoder.cu:354)
    }
    return check_launch();
}
template <typename T>
static int launch_fwd(const float* inputs, const T* emb, const int32_t* offsets, T* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                      const LevelParams& lp, T* dy_dx, uint32_t gridtype, bool ac, int layout, hipStream_t s) {
    if (D == 3 && C == 2 && !dy_dx && g_opt_grid_fast) {
        const dim3 grid(cdiv(B, 256), L), block(256);
        if (layout == PNR_LAYOUT_ROWS) hipLaunchKernelGGL((k_grid_fwd_d3c2<T, true>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, gridtype, ac);
        else if (g_opt_grid_nt == 1) hipLaunchKernelGGL((k_grid_fwd_d3c2<T, false, 1>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, gridtype, ac);
        else if (g_opt_grid_nt == 2) hipLaunchKernelGGL((k_grid_fwd_d3c2<T, false, 2>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, gridtype, ac);
        else if (g_opt_grid_nt == 3) hipLaunchKernelGGL((k_grid_fwd_d3c2<T, false, 3>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, gridtype, ac);
        else hipLaunchKernelGGL((k_grid_fwd_d3c2<T, false>), grid, block, 0, s, inputs, emb, offsets, outputs, B, L, lp, gridtype, ac);
        return check_launch();
    }
    if (layout != PNR_LAYOUT_LEVELS) return PNR_ERR_UNSUPPORTED;
    switch (D) {
        case 1: return launch_fwd_c<T, 1>(inputs, emb, offsets, outputs, B, C, L, lp, dy_dx, gridtype, ac, s);
        case 2: return launch_fwd_c<T, 2>(inputs, emb, offsets, outputs, B, C, L, lp, dy_dx, gridtype, ac, s);
        case 3: return launch_fwd_c<T, 3>(inputs, emb, offsets, outputs, B, C, L, lp, dy_dx, gridtype, ac, s);
        case 4: return launch_fwd_c<T, 4>(inputs, emb, offsets, outputs, B, C, L, lp, dy_dx, gridtype, ac, s);
        case 5: return launch_fwd_c<T, 5>(inputs, emb, offsets, outputs, B, C, L, lp, dy_dx, gridtype, ac, s);
        default: return PNR_ERR_UNSUPPORTED;  // "GridEncoding: D must be 1, 2, 3, 4, or 5." (gridencoder.cu:372)
    }
}

// levels whose cells are wide compared with the sample spacing get the run-combining kernel (fp32 only)
static uint32_t count_coarse_levels(const LevelParams& lp, uint32_t L) {
    uint32_t n = 0;
    while (n < L && lp.scale[n] <= 384.0f) n++;   // scales grow monotonically with the level
    return n;
}

template <typename T, uint32_t D>
static int launch_bwd_c(const T* grad, const float* inputs, const int32_t* offsets, T* gg, uint32_t B, uint32_t C, uint32_t L,
                        const LevelParams& lp, uint32_t gridtype, bool ac, hipStream_t s) {
    const uint32_t nc = (sizeof(T) == 4 || C % 2 == 0) ? count_coarse_levels(lp, L) : 0;
    const dim3 block(256);
#define PNR_BWD(CV)                                                                                                                        \
    if (nc) hipLaunchKernelGGL((k_grid_bwd<T, D, CV, true>), dim3(cdiv(B, 256), nc), block, 0, s, grad, inputs, offsets, gg, B, L, lp, gridtype, ac, 0u); \
    if (nc < L) hipLaunchKernelGGL((k_grid_bwd<T, D, CV, false>), dim3(cdiv(B, 256), L - nc), block, 0, s, grad, inputs, offsets, gg, B, L, lp, gridtype, ac, nc)
    switch (C) {
        case 1: PNR_BWD(1); break;
        case 2: PNR_BWD(2); break;
        case 4: PNR_BWD(4); break;
        case 8: PNR_BWD(8); break;
        default: return PNR_ERR_UNSUPPORTED;
    }
#undef PNR_BWD
    return check_launch();
}
template <typename T>
static int launch_bwd(const T* grad, const float* inputs, const int32_t* offsets, T* gg, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                      const LevelParams& lp, uint32_t gridtype, bool ac, hipStream_t s) {
    switch (D) {
        case 1: return launch_bwd_c<T, 1>(grad, inputs, offsets, gg, B, C, L, lp, gridtype, ac, s);
        case 2: return launch_bwd_c<T, 2>(grad, inputs, offsets, gg, B, C, L, lp, gridtype, ac, s);
        case 3: return launch_bwd_c<T, 3>(grad, inputs, offsets, gg, B, C, L, lp, gridtype, ac, s);
        case 4: return launch_bwd_c<T, 4>(grad, inputs, offsets, gg, B, C, L, lp, gridtype, ac, s);
        case 5: return launch_bwd_c<T, 5>(grad, inputs, offsets, gg, B, C, L, lp, gridtype, ac, s);
        default: return PNR_ERR_UNSUPPORTED;
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_grid_encode_forward_pair(const float* inputs, const float* pair_embeddings, const int32_t* offsets, float* out0, float* out1, uint32_t B, uint32_t L, float S,
                                 uint32_t H, uint32_t gridtype, int align_corners, pnr_stream_t stream) {
    if (L == 0 || L > kMaxLevels) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!inputs || !pair_embeddings || !offsets || !out0 || !out1) return PNR_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(pair_embeddings) & 15u) != 0) return PNR_ERR_INVALID;    // 16-byte rows
    const LevelParams lp = make_level_params(L, S, H);
    hipLaunchKernelGGL(k_grid_fwd_d3c2_pair, dim3(cdiv(B, 256), L), dim3(256), 0, as_stream(stream), inputs, reinterpret_cast<const f32x4*>(pair_embeddings), offsets, out0,
                       out1, B, L, lp, gridtype, align_corners != 0);
    return check_launch();
}

int pnr_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D,
                            uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                            pnr_stream_t stream) {
    return pnr_grid_encode_forward_layout(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, dtype, PNR_LAYOUT_LEVELS, stream);
}

int pnr_grid_encode_forward_layout(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D,
                                   uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                   int layout, pnr_stream_t stream) {
    if (layout != PNR_LAYOUT_LEVELS && layout != PNR_LAYOUT_ROWS) return PNR_ERR_INVALID;
    if (L == 0 || L > kMaxLevels) return PNR_ERR_UNSUPPORTED;
    if (dtype != PNR_DTYPE_F32 && dtype != PNR_DTYPE_F16) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return PNR_ERR_INVALID;
    const LevelParams lp = make_level_params(L, S, H);
    if (dtype == PNR_DTYPE_F32)
        return launch_fwd<float>(inputs, static_cast<const float*>(embeddings), offsets, static_cast<float*>(outputs), B, D, C, L, lp,
                                 static_cast<float*>(dy_dx), gridtype, align_corners != 0, layout, as_stream(stream));
    return launch_fwd<__half>(inputs, static_cast<const __half*>(embeddings), offsets, static_cast<__half*>(outputs), B, D, C, L, lp,
                              static_cast<__half*>(dy_dx), gridtype, align_corners != 0, layout, as_stream(stream));
}

int pnr_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets, void* grad_embeddings,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void* dy_dx, void* grad_inputs,
                             uint32_t gridtype, int align_corners, int dtype, pnr_stream_t stream) {
    (void)embeddings;  // the reference passes it but its kernel never reads it (gridencoder.cu:230)
    if (L == 0 || L > kMaxLevels) return PNR_ERR_UNSUPPORTED;
    if (dtype != PNR_DTYPE_F32 && dtype != PNR_DTYPE_F16) return PNR_ERR_UNSUPPORTED;
    if (D < 1 || D > 5) return PNR_ERR_UNSUPPORTED;
    if (C != 1 && C != 2 && C != 4 && C != 8) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings) return PNR_ERR_INVALID;
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return PNR_ERR_INVALID;
    const LevelParams lp = make_level_params(L, S, H);
    hipStream_t s = as_stream(stream);
    int rc;
    if (dtype == PNR_DTYPE_F32) {
        rc = launch_bwd<float>(static_cast<const float*>(grad), inputs, offsets, static_cast<float*>(grad_embeddings), B, D, C, L, lp, gridtype,
                               align_corners != 0, s);
        if (rc == PNR_OK && dy_dx) {
            hipLaunchKernelGGL(k_grid_input_bwd<float>, dim3(cdiv(B * D, 256)), dim3(256), 0, s, static_cast<const float*>(grad),
                               static_cast<const float*>(dy_dx), static_cast<float*>(grad_inputs), B, D, C, L);
            rc = check_launch();
        }
    } else {
        rc = launch_bwd<__half>(static_cast<const __half*>(grad), inputs, offsets, static_cast<__half*>(grad_embeddings), B, D, C, L, lp,
                                gridtype, align_corners != 0, s);
        if (rc == PNR_OK && dy_dx) {
            hipLaunchKernelGGL(k_grid_input_bwd<__half>, dim3(cdiv(B * D, 256)), dim3(256), 0, s, static_cast<const __half*>(grad),
                               static_cast<const __half*>(dy_dx), static_cast<__half*>(grad_inputs), B, D, C, L);
            rc = check_launch();
        }
    }
    return rc;
}

}  // extern "C"
