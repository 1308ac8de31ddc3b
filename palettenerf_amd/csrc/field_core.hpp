// field_core.hpp -- MFMA fragment helpers and the packed-weight layout of the fused NeRF field (see field.hip).
#pragma once
#include "pnr_common.hpp"
#include "sh_eval.hpp"

namespace pnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// packed weight blob: offsets (in floats) of each layer, layout [row tile][step][64 lanes]
constexpr int kS0 = 0;                 // sigma_net[0]: 32 -> 64   : 2 row tiles x 16 steps
constexpr int kS1 = kS0 + 2 * 16 * 64; // sigma_net[1]: 64 -> 16   : 1 row tile  x 32 steps
constexpr int kC0 = kS1 + 1 * 32 * 64; // color_net[0]: 31 -> 64   : 2 row tiles x 16 steps
constexpr int kC1 = kC0 + 2 * 16 * 64; // color_net[1]: 64 -> 64   : 2 row tiles x 32 steps
constexpr int kC2 = kC1 + 2 * 32 * 64; // color_net[2]: 64 -> 3    : 1 row tile  x 32 steps
constexpr int kPackedFloats = kC2 + 1 * 32 * 64;  // 12288 floats = 48 KiB

// feature held by accumulator register r of a lane in half h (D layout of the 32x32 MFMA family)
__host__ __device__ constexpr int frag_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = fmaxf(v[i], 0.0f);
    return v;
}
// acc += W[tile] . act, act given as one accumulator fragment (16 k-steps)
__device__ __forceinline__ f32x16 mma_frag(f32x16 acc, const float* __restrict__ w /* [16][64] */, const f32x16& act, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r * 64 + lane], act[r], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);  // keep hipcc from hoisting every later weight read above this group (VGPR blow-up, spills)
    return acc;
}

}  // namespace pnr
