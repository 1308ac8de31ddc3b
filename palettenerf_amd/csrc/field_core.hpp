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

// lane-half select that stays in registers: hipcc turns `h ? a[8 + j] : a[j]` over a local array into an indexed
// scratch load; a bitfield insert cannot be rewritten that way and is exact.
__device__ __forceinline__ float select_half(int h, float lower, float upper) {
    const uint32_t m = 0u - (uint32_t)h;  // 0 or all-ones
    return __uint_as_float((__float_as_uint(lower) & ~m) | (__float_as_uint(upper) & m));
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    // max on the bit pattern as a signed integer: negative floats (sign bit set) are negative integers, non-negative floats keep
    // their order -- one v_max_i32 per element, where fmaxf costs an extra canonicalising v_max_f32 on MFMA results
    for (int i = 0; i < 16; i++) v[i] = __int_as_float(max(__float_as_int(v[i]), 0));
    return v;
}
// acc += W[tile] . act, act given as one accumulator fragment (16 k-steps)
__device__ __forceinline__ f32x16 mma_frag(f32x16 acc, const float* __restrict__ w /* [16][64] */, const f32x16& act, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r * 64 + lane], act[r], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);  // keep hipcc from hoisting every later weight read above this group (VGPR blow-up, spills)
    return acc;
}

// ------------------------------------------------------------------------------------------
// split-fp16 ("f16x3") variant: fp32-class accuracy at the fp16 matrix-core rate.
// Every fp32 operand v is split as v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (22 significant bits),
// and a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: 3 MFMAs of K = 16 at 32 cycles replace 8 fp32 MFMAs of K = 2 at 64 cycles (5.3x fewer
// matrix-pipe cycles); the dropped a_lo.b_lo term is ~2^-22 relative.  Weights are split once at pack time.
// Same register-resident chaining as the fp32 path: the D fragment of a layer is the B operand of the next,
// eight accumulator registers per K = 16 block.
// Blob layout: 24 blocks of 2 KiB, block = [hi: 64 lanes x 8 halfs][lo: 64 lanes x 8 halfs]
//   S0: q = rt*2 + kb (0..3) | S1: 4 + kb (kb 0..3) | C0: 8 + rt*2 + kb | C1: 12 + rt*4 + kb | C2: 20 + kb
// ------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int kF16Blocks = 24;
constexpr int kF16BlockBytes = 2048;
constexpr int kPackedBytesF16 = kF16Blocks * kF16BlockBytes;  // 48 KiB, same LDS footprint as the fp32 blob
static_assert(kPackedBytesF16 == kPackedFloats * 4, "both blobs are 48 KiB");

// weight column (input feature) that element j of lane-half h multiplies in k-block kb of each layer; -1 = zero pad
__host__ __device__ constexpr int f16_col_S0(int kb, int h, int j) { return 16 * kb + 8 * h + j; }
__host__ __device__ constexpr int f16_col_from_frag(int kb, int h, int j) { return (kb / 2) * 32 + frag_row((kb % 2) * 8 + j, h); }
__host__ __device__ constexpr int f16_col_C0(int kb, int h, int j) {
    if (kb == 0) return 8 * h + j;               // SH coefficient
    const int g = frag_row(j, h);                // geo feature g (row g of the sigma_net[1] tile); g == 0 is the sigma logit
    return g >= 1 ? 15 + g : -1;
}

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float v[8], h8& hi, h8& lo) {
    // hi = fp16(v) (round to nearest even, two per v_cvt_pk_f16_f32); lo = fp16(v - hi): v - hi is exact in fp32 (hi keeps v's leading 11 bits),
    // so v_fma_mixlo_f16 / v_fma_mixhi_f16 -- fma with the f16 source converted inside the instruction, result rounded ONCE to f16 and
    // written to the low / high half of the destination -- give the packed lo pair in two instructions: 3 VALU per pair of values
    // (it was 4 with v_fma_mix_f32 + a second v_cvt_pk_f16_f32)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hw, lw;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const h2v hp = {(_Float16)v[j], (_Float16)v[j + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, hp);
        uint32_t lb;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(v[j]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(v[j + 1]));
        hw[j / 2] = hb; lw[j / 2] = lb;
    }
    // The low halves come out of inline asm: the compiler's hazard recogniser does not see a VALU write there, and the next instruction may be
    // the MFMA that reads these registers as its B operand (VALU-write -> MFMA-read needs wait states on gfx9).  Four wait states behind the last write (s_nop 1 already made the 12-wave kernel, where this was first seen as run-to-run differences of 1e-6, bit-reproducible)
    // covers all four registers (the first three are at least two instructions old by then).
    asm volatile("s_nop 3" : "+v"(lw));
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}
__device__ __forceinline__ void split_frag(const f32x16& a, int half_idx, h8& hi, h8& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = a[half_idx * 8 + j];
    split8(v, hi, lo);
}
// Overflow watch of the split-fp16 path: fp16 holds magnitudes up to 65 504, a larger operand would be split into (inf, nan).  When the
// host cannot rule that out from the weights (fused.py: static activation bound), the CHECK = true instantiations remember the largest
// |operand| they split (integer max on the bit patterns, ~1 VALU per operand) and raise a flag; the frame is then rendered again on the
// exact fp32 path.  CHECK = false compiles to nothing.
constexpr int kF16MaxBits = 0x477FE000;   // 65504.0f
template <bool CHECK> struct SplitWatch;
template <> struct SplitWatch<false> {
    __device__ __forceinline__ void see(const float*) {}
    __device__ __forceinline__ bool overflowed() const { return false; }
};
template <> struct SplitWatch<true> {
    int mx = 0;
    __device__ __forceinline__ void see(const float v[8]) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) mx = max(max(mx, __float_as_int(v[j]) & 0x7fffffff), __float_as_int(v[j + 1]) & 0x7fffffff);   // NaN patterns are larger still
    }
    __device__ __forceinline__ bool overflowed() const { return mx > kF16MaxBits; }
};
template <bool CHECK>
__device__ __forceinline__ void split_frag_w(SplitWatch<CHECK>& sw, const f32x16& a, int half_idx, h8& hi, h8& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = a[half_idx * 8 + j];
    sw.see(v);
    split8(v, hi, lo);
}

__device__ __forceinline__ f32x16 mma3(f32x16 acc, const unsigned char* __restrict__ wblock, const h8& bhi, const h8& blo, int lane) {
    const h8 ahi = *reinterpret_cast<const h8*>(wblock + lane * 16);
    const h8 alo = *reinterpret_cast<const h8*>(wblock + 1024 + lane * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    return acc;
}

// "f16x2": the activation rounded ONCE to fp16 (no lo half: one v_cvt_pk_f16_f32 per pair instead of three instructions), the weight still
// split -- a_hi.b + a_lo.b, two MFMAs instead of three.  Used for the COLOUR layers only: sigma_net keeps the split form, so densities, alphas and
// the march are bit for bit those of f16x3 and a pixel's error is bounded by the per-sample colour error.  The rounding of b costs 2^-12 relative per activation: ~1e-5 on a colour, ~6e-5
// relative on sigma with unit-scale weights (tests/) -- inside the 1e-4 colour contract, four digits short of the split form.
__device__ __forceinline__ void round8(const float v[8], h8& hi) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hw;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const h2v hp = {(_Float16)v[j], (_Float16)v[j + 1]};
        hw[j / 2] = __builtin_bit_cast(uint32_t, hp);
    }
    hi = __builtin_bit_cast(h8, hw);
}
__device__ __forceinline__ f32x16 mma2(f32x16 acc, const unsigned char* __restrict__ wblock, const h8& bhi, int lane) {
    const h8 ahi = *reinterpret_cast<const h8*>(wblock + lane * 16);
    const h8 alo = *reinterpret_cast<const h8*>(wblock + 1024 + lane * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    return acc;
}
// the two forms behind one name: LO = true is the split form (f16x3), LO = false the rounded one (f16x2)
template <bool LO> __device__ __forceinline__ void splitx(const float v[8], h8& hi, h8& lo) {
    if constexpr (LO) split8(v, hi, lo); else round8(v, hi);
}
template <bool LO> __device__ __forceinline__ f32x16 mmax(f32x16 acc, const unsigned char* __restrict__ wblock, const h8& bhi, const h8& blo, int lane) {
    if constexpr (LO) return mma3(acc, wblock, bhi, blo, lane); else return mma2(acc, wblock, bhi, lane);
}
template <bool LO, bool CHECK>
__device__ __forceinline__ void splitx_frag_w(SplitWatch<CHECK>& sw, const f32x16& a, int half_idx, h8& hi, h8& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = a[half_idx * 8 + j];
    sw.see(v);
    splitx<LO>(v, hi, lo);
}

// ------------------------------------------------------------------------------------------
// One K = 16 weight block against one group of 8 activations per lane-half, in either arithmetic -- the PaletteNeRF field is written
// once over this interface.  A block is 2 KiB in both layouts: split-fp16 [hi: 64 lanes x 8 halfs][lo: same], or fp32 [8 steps][64 lanes]
// where step j multiplies the activation v[j] of the lane-half: the same (row, column) pairs meet in both, only the number format of
// the products differs (fp32 here: v_mfma_f32_32x32x2_f32, exact fp32 products and sums).
// ------------------------------------------------------------------------------------------
template <int PREC> struct BOp;
template <> struct BOp<1> { h8 hi, lo; };
template <> struct BOp<2> { h8 hi; };        // f16x2: the activation rounded once (mma2)
template <> struct BOp<0> { float v[8]; };

template <int PREC, bool CHECK = false> __device__ __forceinline__ BOp<PREC> make_op(const float v[8], SplitWatch<CHECK>* sw = nullptr) {
    BOp<PREC> o;
    if constexpr (PREC == 1) { if constexpr (CHECK) sw->see(v); split8(v, o.hi, o.lo); }
    else if constexpr (PREC == 2) { if constexpr (CHECK) sw->see(v); round8(v, o.hi); }
    else {
#pragma unroll
        for (int j = 0; j < 8; j++) o.v[j] = v[j];
    }
    return o;
}
template <int PREC, bool CHECK = false> __device__ __forceinline__ BOp<PREC> frag_op(const f32x16& a, int half_idx, SplitWatch<CHECK>* sw = nullptr) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = a[half_idx * 8 + j];
    return make_op<PREC, CHECK>(v, sw);
}
template <int PREC> __device__ __forceinline__ f32x16 mma_blk(f32x16 acc, const unsigned char* __restrict__ wblock, const BOp<PREC>& b, int lane) {
    if constexpr (PREC == 1) return mma3(acc, wblock, b.hi, b.lo, lane);
    else if constexpr (PREC == 2) return mma2(acc, wblock, b.hi, lane);
    else {
        const float* w = reinterpret_cast<const float*>(wblock);
#pragma unroll
        for (int j = 0; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j * 64 + lane], b.v[j], acc, 0, 0, 0);
        return acc;
    }
}

// One 64 -> 3 head on the vector unit (palette_field.hip explains why: as a 32-row MFMA tile 29 of the 32 output rows are padding): `vec` = this lane-half's 96 weights in LDS ([output][tile][register]), u0 / u1 = the lane's two
// (ReLU'd) accumulator tiles.  out[o] = (sum over the lower half-wave's 32 features) + (sum over the upper half-wave's), the same bits in both lanes of a sample.
__device__ __forceinline__ void head3_valu(const unsigned char* __restrict__ vec, const f32x16& u0, const f32x16& u1, float out[3]) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    const lds_f32x4* wv = reinterpret_cast<const lds_f32x4*>(reinterpret_cast<uintptr_t>(vec));
    // (the three outputs' chains advance together: a packed fp32 instruction whose result the next instruction reads costs a wait state on gfx950 -- three
    // independent accumulators leave none; each accumulator still sees its own terms in the same order)
    f32x2v acc[3] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x16& u = t ? u1 : u0;
            f32x4 ww[3];
#pragma unroll
            for (int o = 0; o < 3; o++) ww[o] = wv[(o * 2 + t) * 4 + r4];
#pragma unroll
            for (int o = 0; o < 3; o++) acc[o] = __builtin_elementwise_fma(f32x2v{ww[o].x, ww[o].y}, f32x2v{u[4 * r4], u[4 * r4 + 1]}, acc[o]);
#pragma unroll
            for (int o = 0; o < 3; o++) acc[o] = __builtin_elementwise_fma(f32x2v{ww[o].z, ww[o].w}, f32x2v{u[4 * r4 + 2], u[4 * r4 + 3]}, acc[o]);
        }
#pragma unroll
    for (int o = 0; o < 3; o++) {
        const float p = acc[o].x + acc[o].y;
        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);   // ([0]: the lower half-wave's p in every lane, [1]: the upper's)
        out[o] = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
    }
}

constexpr uint32_t kVecHeadHalfBytes = 3 * 2 * 16 * 4;     // one lane-half's 96 weights
// w [3][64] row-major (nn.Linear) -> [half-wave h][output o][tile t][register r] = w[o][32 t + frag_row(r, h)]: what lane-half h multiplies its registers by
static __global__ void k_pack_vec_head(const float* __restrict__ w, float* __restrict__ out) {
    const int i = threadIdx.x;     // 192 threads
    if (i >= 192) return;
    const int r = i & 15, t = (i >> 4) & 1, o = (i >> 5) % 3, h = i / 96;
    out[i] = w[o * 64 + 32 * t + frag_row(r, h)];
}


struct FieldOut { float sigma_logit, o0, o1, o2; };

// One wave-tile (32 samples) of the NeRF field, split-fp16 matrix path.  w: the 48 KiB blob in LDS.
// enc_scale: a power of two (1 = none) the encoder features are multiplied by before they are split into fp16 pairs, and sigma_net's 16
// outputs divided by afterwards -- exact, since the bias-free ReLU stack is positively homogeneous.  It keeps the `lo` halves of very small
// features (hash tables at the reference's initialisation scale U(-1e-4, 1e-4), gridencoder/grid.py:107) out of the fp16 subnormal range.
// the 8 encoder rows of a lane (levels 4h..4h+3 and 8+4h..8+4h+3; 32 samples x 2 channels = 256 contiguous bytes per level and half-wave),
// issued together into fresh registers (load8_fresh, pnr_common.hpp).  `row` is a row that exists (callers clamp it); dead lanes get zeros.
__device__ __forceinline__ void load_enc_rows8(const float* __restrict__ enc, size_t level_stride, uint32_t row, bool valid, int h, float x[2][8]) {
    const f32x2* p[8];
    f32x2 v[8];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) p[4 * kb + q] = reinterpret_cast<const f32x2*>(enc + ((size_t)(8 * kb + 4 * h + q) * level_stride + row) * 2);
    load8_fresh(p, v);
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) { x[kb][2 * q] = valid ? v[4 * kb + q].x : 0.0f; x[kb][2 * q + 1] = valid ? v[4 * kb + q].y : 0.0f; }
}
#ifndef PNR_NERF_SPLIT_PER_BLOCK
#define PNR_NERF_SPLIT_PER_BLOCK 1
#endif
template <bool CHECK, bool LO = true>
__device__ __forceinline__ FieldOut nerf_field_tile_f16x3(const unsigned char* __restrict__ w, int lane, bool valid, const float* __restrict__ enc,
                                                          size_t level_stride, uint32_t row, float dx, float dy, float dz, float enc_scale,
                                                          SplitWatch<CHECK>& sw) {
    const int h = lane >> 5;
    h8 bh[4], bl[4];
    {   // sigma_net[0] inputs: k-block kb, element j  <-  encoder feature 16 kb + 8 h + j  = (level 8 kb + 4 h + j/2, channel j&1)
        float x[2][8];
        load_enc_rows8(enc, level_stride, row, valid, h, x);
        if (enc_scale != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; j++) { x[0][j] *= enc_scale; x[1][j] *= enc_scale; }
        }
        sw.see(x[0]); sw.see(x[1]);
        split8(x[0], bh[0], bl[0]);
        split8(x[1], bh[1], bl[1]);
    }
    f32x16 h0 = zero16(), h1 = zero16();
    h0 = mma3(h0, w + 0 * kF16BlockBytes, bh[0], bl[0], lane);
    h0 = mma3(h0, w + 1 * kF16BlockBytes, bh[1], bl[1], lane);
    h1 = mma3(h1, w + 2 * kF16BlockBytes, bh[0], bl[0], lane);
    h1 = mma3(h1, w + 3 * kF16BlockBytes, bh[1], bl[1], lane);
    __builtin_amdgcn_sched_barrier(0);
    h0 = relu16(h0); h1 = relu16(h1);

    // sigma_net[1]: 64 -> 16.  (PNR_NERF_SPLIT_PER_BLOCK: each k-block's activations are split right in front of its products, so that hipcc threads
    // block k + 1's split between block k's matrix instructions -- profiles/micro/mfma_interleave.hip: 134 against 158 cycles per block for
    // "threaded" against "all splits, then the chain"; same products in the same order per accumulator: same bits.)
    f32x16 g = zero16();
#if PNR_NERF_SPLIT_PER_BLOCK
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        split_frag_w(sw, kb < 2 ? h0 : h1, kb & 1, bh[0], bl[0]);
        g = mma3(g, w + (4 + kb) * kF16BlockBytes, bh[0], bl[0], lane);
    }
#else
    split_frag_w(sw, h0, 0, bh[0], bl[0]); split_frag_w(sw, h0, 1, bh[1], bl[1]);
    split_frag_w(sw, h1, 0, bh[2], bl[2]); split_frag_w(sw, h1, 1, bh[3], bl[3]);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) g = mma3(g, w + (4 + kb) * kF16BlockBytes, bh[kb], bl[kb], lane);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (enc_scale != 1.0f) {
        const float inv = 1.0f / enc_scale;
#pragma unroll
        for (int r = 0; r < 8; r++) g[r] *= inv;   // rows 0..15 live in registers 0..7 of both half-waves
    }
    FieldOut out;
    out.sigma_logit = g[0];

    // color_net[0]: [SH16 ; geo15] -> 64
    float sh[16];
    sh_eval<4>(dx, dy, dz, sh);
    {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = select_half(h, sh[j], sh[8 + j]);
        splitx<LO>(v, bh[0], bl[0]);
    }
    splitx_frag_w<LO>(sw, g, 0, bh[1], bl[1]);
    f32x16 c0 = zero16(), c1 = zero16();
    c0 = mmax<LO>(c0, w + 8 * kF16BlockBytes, bh[0], bl[0], lane);
    c0 = mmax<LO>(c0, w + 9 * kF16BlockBytes, bh[1], bl[1], lane);
    c1 = mmax<LO>(c1, w + 10 * kF16BlockBytes, bh[0], bl[0], lane);
    c1 = mmax<LO>(c1, w + 11 * kF16BlockBytes, bh[1], bl[1], lane);
    __builtin_amdgcn_sched_barrier(0);
    c0 = relu16(c0); c1 = relu16(c1);

    // color_net[1]: 64 -> 64
    f32x16 d0 = zero16(), d1 = zero16();
#if PNR_NERF_SPLIT_PER_BLOCK
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        splitx_frag_w<LO>(sw, kb < 2 ? c0 : c1, kb & 1, bh[0], bl[0]);
        d0 = mmax<LO>(d0, w + (12 + kb) * kF16BlockBytes, bh[0], bl[0], lane);
        d1 = mmax<LO>(d1, w + (16 + kb) * kF16BlockBytes, bh[0], bl[0], lane);
    }
#else
    splitx_frag_w<LO>(sw, c0, 0, bh[0], bl[0]); splitx_frag_w<LO>(sw, c0, 1, bh[1], bl[1]);
    splitx_frag_w<LO>(sw, c1, 0, bh[2], bl[2]); splitx_frag_w<LO>(sw, c1, 1, bh[3], bl[3]);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) d0 = mmax<LO>(d0, w + (12 + kb) * kF16BlockBytes, bh[kb], bl[kb], lane);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) d1 = mmax<LO>(d1, w + (16 + kb) * kF16BlockBytes, bh[kb], bl[kb], lane);
#endif
    __builtin_amdgcn_sched_barrier(0);
    d0 = relu16(d0); d1 = relu16(d1);

    // color_net[2]: 64 -> 3 -- fp32 dot products on the vector unit (round 5; its weights sit where block 20 was: [half-wave][output][tile][register])
    float o[3];
    head3_valu(w + 20 * kF16BlockBytes + (uint32_t)h * kVecHeadHalfBytes, d0, d1, o);
    out.o0 = o[0]; out.o1 = o[1]; out.o2 = o[2];
    return out;
}

// One wave-tile of the NeRF field, exact-fp32 matrix path.  w: the 48 KiB fp32 blob in LDS.
__device__ __forceinline__ FieldOut nerf_field_tile_f32(const float* __restrict__ w, int lane, bool valid, const float* __restrict__ enc,
                                                        size_t level_stride, uint32_t row, float dx, float dy, float dz) {
    const int h = lane >> 5;
    // sigma_net[0]: B operand of step s = encoder level s, channel h  (coalesced 256-byte rows)
    float x[16];
#pragma unroll
    for (int s = 0; s < 16; s++) x[s] = valid ? enc[((size_t)s * level_stride + row) * 2 + h] : 0.0f;
    f32x16 h0 = zero16(), h1 = zero16();
#pragma unroll
    for (int s = 0; s < 16; s++) {
        h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kS0 + s * 64 + lane], x[s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kS0 + (16 + s) * 64 + lane], x[s], h1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    h0 = relu16(h0); h1 = relu16(h1);
    // sigma_net[1]: 64 -> 16 (rows 16..31 of the tile are zero padding)
    f32x16 g = zero16();
    g = mma_frag(g, &w[kS1], h0, lane);
    g = mma_frag(g, &w[kS1 + 16 * 64], h1, lane);
    FieldOut out;
    out.sigma_logit = g[0];  // row 0 lives in register 0 of the lower half-wave
    // color_net[0]: [SH16 ; geo15] -> 64
    float sh[16];
    sh_eval<4>(dx, dy, dz, sh);
    f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const float b = select_half(h, sh[s], sh[8 + s]);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kC0 + s * 64 + lane], b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kC0 + (16 + s) * 64 + lane], b, c1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; r++) {  // geo features: accumulator registers 0..7 of g (rows 0..15); row 0 carries a zero weight
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kC0 + (8 + r) * 64 + lane], g[r], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kC0 + (24 + r) * 64 + lane], g[r], c1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    c0 = relu16(c0); c1 = relu16(c1);
    // color_net[1]: 64 -> 64
    f32x16 d0 = zero16(), d1 = zero16();
    d0 = mma_frag(d0, &w[kC1], c0, lane);
    d0 = mma_frag(d0, &w[kC1 + 16 * 64], c1, lane);
    d1 = mma_frag(d1, &w[kC1 + 32 * 64], c0, lane);
    d1 = mma_frag(d1, &w[kC1 + 48 * 64], c1, lane);
    d0 = relu16(d0); d1 = relu16(d1);
    // color_net[2]: 64 -> 3
    f32x16 o = zero16();
    o = mma_frag(o, &w[kC2], d0, lane);
    o = mma_frag(o, &w[kC2 + 16 * 64], d1, lane);
    out.o0 = o[0]; out.o1 = o[1]; out.o2 = o[2];
    return out;
}

// sigma_net alone (32 -> 64 ReLU -> 16): the 16 outputs of a sample as the lower 8 accumulator registers of its two half-wave lanes
// (row frag_row(r, h): logit = row 0, geo feature k = row k).  Same arithmetic as the first half of the field tiles above.
template <int PREC>
__device__ __forceinline__ f32x16 nerf_density_tile(const float* __restrict__ wf, int lane, bool valid, const float* __restrict__ enc, size_t level_stride,
                                                    uint32_t row, float enc_scale = 1.0f) {
    const int h = lane >> 5;
    f32x16 h0 = zero16(), h1 = zero16(), g = zero16();
    if constexpr (PREC == 0) {
        float x[16];
#pragma unroll
        for (int s = 0; s < 16; s++) x[s] = valid ? enc[((size_t)s * level_stride + row) * 2 + h] : 0.0f;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[kS0 + s * 64 + lane], x[s], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[kS0 + (16 + s) * 64 + lane], x[s], h1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        h0 = relu16(h0); h1 = relu16(h1);
        g = mma_frag(g, &wf[kS1], h0, lane);
        g = mma_frag(g, &wf[kS1 + 16 * 64], h1, lane);
    } else {
        const unsigned char* w = reinterpret_cast<const unsigned char*>(wf);
        h8 bh[4], bl[4];
        float x[2][8];
        load_enc_rows8(enc, level_stride, row, valid, h, x);
        if (enc_scale != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; j++) { x[0][j] *= enc_scale; x[1][j] *= enc_scale; }
        }
        split8(x[0], bh[0], bl[0]);
        split8(x[1], bh[1], bl[1]);
        h0 = mma3(h0, w + 0 * kF16BlockBytes, bh[0], bl[0], lane);
        h0 = mma3(h0, w + 1 * kF16BlockBytes, bh[1], bl[1], lane);
        h1 = mma3(h1, w + 2 * kF16BlockBytes, bh[0], bl[0], lane);
        h1 = mma3(h1, w + 3 * kF16BlockBytes, bh[1], bl[1], lane);
        __builtin_amdgcn_sched_barrier(0);
        h0 = relu16(h0); h1 = relu16(h1);
        split_frag(h0, 0, bh[0], bl[0]); split_frag(h0, 1, bh[1], bl[1]);
        split_frag(h1, 0, bh[2], bl[2]); split_frag(h1, 1, bh[3], bl[3]);
#pragma unroll
        for (int kb = 0; kb < 4; kb++) g = mma3(g, w + (4 + kb) * kF16BlockBytes, bh[kb], bl[kb], lane);
        if (enc_scale != 1.0f) {
            const float inv = 1.0f / enc_scale;
#pragma unroll
            for (int r = 0; r < 8; r++) g[r] *= inv;
        }
    }
    return g;
}

template <int PREC, bool CHECK = false>
__device__ __forceinline__ FieldOut nerf_field_tile(const float* __restrict__ w, int lane, bool valid, const float* __restrict__ enc,
                                                    size_t level_stride, uint32_t row, float dx, float dy, float dz, float enc_scale, SplitWatch<CHECK>& sw) {
    if constexpr (PREC == 0) return nerf_field_tile_f32(w, lane, valid, enc, level_stride, row, dx, dy, dz);   // exact fp32: no scaling needed
    else return nerf_field_tile_f16x3<CHECK, PREC == 1>(reinterpret_cast<const unsigned char*>(w), lane, valid, enc, level_stride, row, dx, dy, dz, enc_scale, sw);
}
template <int PREC>
__device__ __forceinline__ FieldOut nerf_field_tile(const float* __restrict__ w, int lane, bool valid, const float* __restrict__ enc,
                                                    size_t level_stride, uint32_t row, float dx, float dy, float dz, float enc_scale = 1.0f) {
    SplitWatch<false> sw;
    return nerf_field_tile<PREC, false>(w, lane, valid, enc, level_stride, row, dx, dy, dz, enc_scale, sw);
}

}  // namespace pnr
