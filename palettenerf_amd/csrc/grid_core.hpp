// grid_core.hpp -- hash-grid indexing / interpolation helpers shared by gridencoder.hip and frame.hip.
#pragma once
#include "pnr_common.hpp"
#include <math.h>

namespace pnr {

constexpr uint32_t kMaxLevels = 32;
struct LevelParams {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__half v) { return __half2float(v); }

// reference gridencoder.cu:35-72 (fast_hash + get_grid_index with ch = 0)
template <uint32_t D, uint32_t C>
__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                               const uint32_t pg[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pg[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) index ^= pg[d] * primes[d];
    }
    return (index % hashmap_size) * C;
}

// How a row index is formed on one level of a D = 3 grid (gridencoder.cu:49-72 evaluated once per level instead of once per corner):
//   0 the reference's general form (stride test per dimension, hash or tiled, `%`); 1 dense: side^3 <= size, the index is below the size and the
//   `%` is the identity; 2 hashed with a power-of-two size: a mask.  All three give grid_index's value (tests/test_gpu_ops.py: the D3C2 kernel and
//   the frame loops against the generic kernel and the oracle, incl. sizes that are neither).  align_corners = false.
#ifndef PNR_GRID_KIND
#define PNR_GRID_KIND 1     // 0: every level through the general form (the A/B of the specialised index forms)
#endif
__device__ __forceinline__ uint32_t level_kind(uint32_t gridtype, uint32_t hashmap_size, uint32_t resolution) {
    if (!PNR_GRID_KIND) return 0u;
    const uint32_t side = resolution + 1u;
    if ((uint64_t)side * side * side <= (uint64_t)hashmap_size) return 1u;
    uint32_t stride = 1u;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++)
        if (stride <= hashmap_size) stride *= side;
    return (gridtype == 0u && stride > hashmap_size && (hashmap_size & (hashmap_size - 1u)) == 0u) ? 2u : 0u;
}
// the eight corner rows (x CMUL) of the cell whose lower corner is pg
template <uint32_t CMUL>
__device__ __forceinline__ void corner_rows_by_kind(uint32_t kind, uint32_t gridtype, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg /* [3] */,
                                                    uint32_t* idxs /* [8] */) {
    if (kind == 1u) {
        const uint32_t side = resolution + 1u;
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++)
            idxs[idx] = ((pg[0] + (idx & 1u)) + (pg[1] + ((idx >> 1) & 1u)) * side + (pg[2] + ((idx >> 2) & 1u)) * side * side) * CMUL;
    } else if (kind == 2u) {
        const uint32_t mask = hashmap_size - 1u;
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++)
            idxs[idx] = (((pg[0] + (idx & 1u)) ^ ((pg[1] + ((idx >> 1) & 1u)) * 2654435761u) ^ ((pg[2] + ((idx >> 2) & 1u)) * 805459861u)) & mask) * CMUL;
    } else {
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            uint32_t pl[3];
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) pl[d] = pg[d] + ((idx >> d) & 1u);
            idxs[idx] = grid_index<3, CMUL>(gridtype, false, hashmap_size, resolution, pl);
        }
    }
}

// accumulate one corner: fp32 table -> fmaf chain; fp16 table -> half accumulator with the
// reference's two roundings (gridencoder.cu:142,165 with scalar_t = at::Half)
template <uint32_t C>
__device__ __forceinline__ void corner_accumulate(float acc[C], float w, const float* __restrict__ g) {
    if constexpr (C == 2) {
        const float2 v = *reinterpret_cast<const float2*>(g);
        acc[0] = fmaf(w, v.x, acc[0]); acc[1] = fmaf(w, v.y, acc[1]);
    } else if constexpr (C == 4) {
        const float4 v = *reinterpret_cast<const float4*>(g);
        acc[0] = fmaf(w, v.x, acc[0]); acc[1] = fmaf(w, v.y, acc[1]); acc[2] = fmaf(w, v.z, acc[2]); acc[3] = fmaf(w, v.w, acc[3]);
    } else {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) acc[ch] = fmaf(w, g[ch], acc[ch]);
    }
}
template <uint32_t C>
__device__ __forceinline__ void corner_accumulate(__half acc[C], float w, const __half* __restrict__ g) {
    __half v[C];
    if constexpr (C == 2) {
        *reinterpret_cast<__half2*>(v) = *reinterpret_cast<const __half2*>(g);
    } else if constexpr (C == 4) {
        *reinterpret_cast<uint2*>(v) = *reinterpret_cast<const uint2*>(g);
    } else if constexpr (C == 8) {
        *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(g);
    } else {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) v[ch] = g[ch];
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++)
        acc[ch] = __float2half(__half2float(acc[ch]) + __half2float(__float2half(w * __half2float(v[ch]))));
}

// 1 / v when v is a power of two (then x / v == x * (1 / v) bit for bit for every finite x that does not end subnormal), else 0: "divide"
static inline float exact_reciprocal_or_zero(float v) {
    int e2 = 0;
    const float mant = frexpf(v, &e2);
    return (mant == 0.5f && e2 > -100 && e2 < 100) ? 1.0f / v : 0.0f;
}

static inline LevelParams make_level_params(uint32_t L, float S, uint32_t H) {
    LevelParams lp;
    for (uint32_t l = 0; l < kMaxLevels; l++) { lp.scale[l] = 0; lp.resolution[l] = 0; }
    for (uint32_t l = 0; l < L; l++) {
        lp.scale[l] = exp2f((float)l * S) * (float)H - 1.0f;               // gridencoder.cu:125
        lp.resolution[l] = (uint32_t)ceil((double)lp.scale[l]) + 1;        // gridencoder.cu:126
    }
    return lp;
}

}  // namespace pnr
