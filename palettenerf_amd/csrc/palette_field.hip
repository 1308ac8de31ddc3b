// palette_field.hip -- fused evaluation of the PaletteNeRF field AND the palette colour-basis composite on the
// gfx950 matrix cores (MI355X-first; the reference runs ~14 GEMMs + ~25 elementwise launches per iteration).
//
// Per sample (palette/network.py:156-280, palette/renderer.py:470-500, inference branch without edit / stylizer):
//   h      = sigma_net(enc)                       32 -> 64 -> 16      sigma = exp(h0), geo = h[1:16]
//   clip   = clip_net(enc_clip)                   32 -> 64 -> clip_dim            (only with pred_clip; else zeros)
//   diff   = sigmoid(diff_net(geo))               15 -> 64 -> 64 -> 3
//   vd     = sigmoid(color_net([SH16(d) ; geo]))  31 -> 64 -> 64 -> 3
//   p      = basis_net([enc_palette ; diff])      35 -> 64 (ELU) -> 15
//   o_r    = offsets_radiance_net(p)              15 -> 3 nb + 1 (with bias)
//   omega  = normalise(softplus(omega_net(p)) + 0.05)                  15 -> nb
//   final_b = softplus(radiance) * (clamp(P_b,0,1) + k_off * offset_b) ; basis_rgb_b = omega_b * final_b
//   rgb    = sum_b basis_rgb_b + k_vd * vd
// Outputs: sigma * density_scale, rgb, and ONE packed auxiliary row
//   aux = [direct_rgb 3 | view_dep 3 | omega nb | basis_rgb 3nb | unscaled_basis_rgb 3nb | clip clip_dim | 0-pad to x4]
// so that the reference's six composite_rays_flex launches collapse into a single one over the packed row.
//
// Matrix path: split-fp16 (field_core.hpp), activations chained through the register file exactly as in the NeRF
// kernel: 50 (58 with clip) K=16 blocks of 2 KiB in LDS, 3 MFMAs each.  Layers whose outputs feed scalar math
// (rgb heads, offsets/radiance, omega, clip) place their rows in the lower half-wave so no cross-lane traffic is needed.
#include "pnr_common.hpp"
#include "field_core.hpp"

namespace pnr {

// ---- block table -------------------------------------------------------------------------------------------------
enum { COL_LINEAR = 0, COL_FRAG = 1, COL_SH_GEO = 2, COL_GEO = 3, COL_ENC_DIFF = 4, COL_FRAG15 = 5 };
enum { ROW_ID = 0, ROW_HALF0 = 1 };
// first block of every layer
enum {
    PB_S0 = 0, PB_S1 = 4, PB_D0 = 8, PB_D1 = 10, PB_D2 = 18, PB_C0 = 22, PB_C1 = 26, PB_C2 = 34, PB_B0 = 38, PB_B1 = 44, PB_OR = 48, PB_OM = 49,
    PB_CL0 = 50, PB_CL1 = 54, PB_END_NOCLIP = 50, PB_END_CLIP = 58
};
constexpr int kPalMaxBlocks = 58;

struct PackBlock { const float* W; int ld, nrows, rt, colkind, kb, rowkind; };
struct PackTable { PackBlock b[kPalMaxBlocks]; int n; };

// tile row -> slot among the 16 rows a lower-half-wave lane holds (registers 0..15), -1 for upper-half rows
__host__ __device__ constexpr int half0_slot(int r) { return ((r >> 2) & 1) ? -1 : ((r >> 3) * 4 + (r & 3)); }

__device__ __forceinline__ int pack_col(int kind, int kb, int h, int j) {
    switch (kind) {
        case COL_LINEAR: return 16 * kb + 8 * h + j;
        case COL_FRAG: return (kb / 2) * 32 + frag_row((kb % 2) * 8 + j, h);
        case COL_SH_GEO: { if (kb == 0) return 8 * h + j; const int g = frag_row(j, h); return g >= 1 ? 15 + g : -1; }
        case COL_GEO: { const int g = frag_row(j, h); return g >= 1 ? g - 1 : -1; }                 // diff_net input = geo_feat (h[1:16])
        case COL_ENC_DIFF: { if (kb < 2) return 16 * kb + 8 * h + j; return (h == 0 && j < 3) ? 32 + j : -1; }  // [enc_palette(32) ; diffuse(3)]
        default: { const int f = frag_row(j, h); return f < 15 ? f : -1; }                        // COL_FRAG15: the 15 basis features
    }
}

__global__ void __launch_bounds__(256) k_pack_blocks_f16x3(PackTable t, unsigned char* __restrict__ packed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.n * 512) return;
    const int q = e / 512, lane = (e / 8) & 63, j = e & 7, i = lane & 31, h = lane >> 5;
    const PackBlock b = t.b[q];
    const int row = b.rowkind == ROW_ID ? b.rt * 32 + i : half0_slot(i);
    const int col = pack_col(b.colkind, b.kb, h, j);
    float v = 0.0f;
    if (row >= 0 && row < b.nrows && col >= 0 && col < b.ld) v = b.W[(size_t)row * b.ld + col];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16* blk = reinterpret_cast<_Float16*>(packed + (size_t)q * kF16BlockBytes);
    blk[lane * 8 + j] = hi;
    blk[512 + lane * 8 + j] = lo;
}

// ---- device helpers ----------------------------------------------------------------------------------------------
// Epilogue transcendentals on the hardware units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1e-7 absolute on these bounded arguments)
// instead of the libm-accurate expansions (15-30 VALU each, ~80 calls per sample row with the ELU layer): the field's outputs
// stay inside the 5e-6 parity band of tests/test_gpu_ops.py and far inside the 1e-4 colour tolerance.
__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : __logf(1.0f + __expf(x)); }  // F.softplus, beta 1, threshold 20
__device__ __forceinline__ float sigmoidf(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ f32x16 elu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = v[i] > 0.0f ? v[i] : __expf(v[i]) - 1.0f;  // F.elu, alpha 1
    return v;
}
// the 8 encoder rows of a lane (levels 4h..4h+3 and 8+4h..8+4h+3), raw: issued at the top of a tile for every table so that the
// loads of the later networks are in flight while the earlier ones compute (the workgroup has registers to spare: LDS, not VGPRs,
// limits it to two waves per SIMD)
__device__ __forceinline__ void load_enc_raw(const float* __restrict__ enc, size_t level_stride, uint32_t row, bool valid, int h, float x[2][8]) {
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int level = 8 * kb + 4 * h + q;
            const float2 v = valid ? *reinterpret_cast<const float2*>(enc + ((size_t)level * level_stride + row) * 2) : make_float2(0.0f, 0.0f);
            x[kb][2 * q] = v.x; x[kb][2 * q + 1] = v.y;
        }
}
__device__ __forceinline__ void load_enc_blocks(const float* __restrict__ enc, size_t level_stride, uint32_t row, bool valid, int h, h8 bh[2], h8 bl[2]) {
    float x[2][8];
    load_enc_raw(enc, level_stride, row, valid, h, x);
    split8(x[0], bh[0], bl[0]);
    split8(x[1], bh[1], bl[1]);
}
// 64 -> N layer from two activation tiles: 4 k-blocks starting at block q0
__device__ __forceinline__ f32x16 dense64(f32x16 acc, const unsigned char* __restrict__ w, int q0, const f32x16& a0, const f32x16& a1, int lane) {
    h8 bh, bl;
    split_frag(a0, 0, bh, bl); acc = mma3(acc, w + (q0 + 0) * kF16BlockBytes, bh, bl, lane);
    split_frag(a0, 1, bh, bl); acc = mma3(acc, w + (q0 + 1) * kF16BlockBytes, bh, bl, lane);
    split_frag(a1, 0, bh, bl); acc = mma3(acc, w + (q0 + 2) * kF16BlockBytes, bh, bl, lane);
    split_frag(a1, 1, bh, bl); acc = mma3(acc, w + (q0 + 3) * kF16BlockBytes, bh, bl, lane);
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

struct PaletteParams {
    float basis_color[5][3];   // already clamped to [0,1]
    float or_bias[16];         // offsets_radiance_net.bias (3 nb + 1 entries)
    float density_scale, offsets_weight, view_dep_weight;
    int nb, clip_dim, pred_clip, aux_stride;
};

constexpr int kPalThreads = 512;

// ctl == nullptr: rows = B (stand-alone op).  Otherwise rows = n_alive * n_step of the frame control block and
// dead slots (delta == 0) are skipped.
struct FrameCtlView { int32_t n_alive, n_step, step, done; };

__global__ void __launch_bounds__(kPalThreads) k_palette_field_fwd(const FrameCtlView* __restrict__ ctl, uint32_t B_arg, const float* __restrict__ enc,
                                                                   const float* __restrict__ enc_pal, const float* __restrict__ enc_clip,
                                                                   uint32_t level_stride, const float* __restrict__ dirs,
                                                                   const float* __restrict__ deltas, const unsigned char* __restrict__ packed,
                                                                   uint32_t packed_bytes, PaletteParams pp, float* __restrict__ sigmas,
                                                                   float* __restrict__ rgbs, float* __restrict__ aux, uint32_t stage_stride,
                                                                   const int32_t* __restrict__ rays_alive, const float* __restrict__ weights_sum,
                                                                   float* __restrict__ aux_map, float T_thresh) {
    if (ctl && ctl->done) return;
    const uint32_t B = ctl ? (uint32_t)ctl->n_alive * (uint32_t)ctl->n_step : B_arg;
    const uint32_t ntiles = (B + 255) / 256;
    if (blockIdx.x >= ntiles) return;
    extern __shared__ unsigned char w[];
    for (uint32_t i = threadIdx.x * 16; i < packed_bytes; i += kPalThreads * 16)
        *reinterpret_cast<uint4*>(&w[i]) = *reinterpret_cast<const uint4*>(&packed[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int nb = pp.nb;
    // rows of a ray are consecutive; with 1, 2, 4 or 8 samples per ray they sit inside one 32-row wave tile and the aux composite
    // can run here (fstep = samples per ray), otherwise the composite launch does it
    const uint32_t fstep = (stage_stride && ctl && aux_map && ctl->n_step <= 8 && (32 % ctl->n_step) == 0) ? (uint32_t)ctl->n_step : 0u;
    const bool fuse_composite = fstep != 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t n = tile * 256 + wave * 32 + (lane & 31);
        const bool valid = n < B && (!deltas || deltas[(size_t)(n < B ? n : 0) * 2] != 0.0f);
        if (!__any(valid)) continue;
        const uint32_t row = n < B ? n : (B - 1);
        h8 bh[2], bl[2];

        // all global reads of the tile up front
        float xs[2][8], xp[2][8];
        load_enc_raw(enc, level_stride, row, valid, h, xs);
        load_enc_raw(enc_pal, level_stride, row, valid, h, xp);
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if (valid) { dx = dirs[(size_t)row * 3]; dy = dirs[(size_t)row * 3 + 1]; dz = dirs[(size_t)row * 3 + 2]; }

        // ---------------- sigma_net
        split8(xs[0], bh[0], bl[0]);
        split8(xs[1], bh[1], bl[1]);
        f32x16 t0 = zero16(), t1 = zero16();
        t0 = mma3(t0, w + (PB_S0 + 0) * kF16BlockBytes, bh[0], bl[0], lane);
        t0 = mma3(t0, w + (PB_S0 + 1) * kF16BlockBytes, bh[1], bl[1], lane);
        t1 = mma3(t1, w + (PB_S0 + 2) * kF16BlockBytes, bh[0], bl[0], lane);
        t1 = mma3(t1, w + (PB_S0 + 3) * kF16BlockBytes, bh[1], bl[1], lane);
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        const f32x16 g = dense64(zero16(), w, PB_S1, t0, t1, lane);   // rows 0..15: sigma logit, geo_feat 1..15
        const float sigma_logit = g[0];
        h8 gh, gl;
        split_frag(g, 0, gh, gl);                                      // the geo k-block, shared by diff_net and color_net

        // ---------------- diff_net: 15 -> 64 -> 64 -> 3
        t0 = mma3(zero16(), w + (PB_D0 + 0) * kF16BlockBytes, gh, gl, lane);
        t1 = mma3(zero16(), w + (PB_D0 + 1) * kF16BlockBytes, gh, gl, lane);
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        f32x16 u0 = dense64(zero16(), w, PB_D1, t0, t1, lane);
        f32x16 u1 = dense64(zero16(), w, PB_D1 + 4, t0, t1, lane);
        u0 = relu16(u0); u1 = relu16(u1);
        f32x16 dif = dense64(zero16(), w, PB_D2, u0, u1, lane);        // rows 0..2
        const float diffuse[3] = {sigmoidf(dif[0]), sigmoidf(dif[1]), sigmoidf(dif[2])};

        // ---------------- color_net (view dependent): [SH16 ; geo15] -> 64 -> 64 -> 3
        {
            float sh[16], v[8];
            sh_eval<4>(dx, dy, dz, sh);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = select_half(h, sh[j], sh[8 + j]);
            split8(v, bh[0], bl[0]);
        }
        t0 = mma3(zero16(), w + (PB_C0 + 0) * kF16BlockBytes, bh[0], bl[0], lane);
        t0 = mma3(t0, w + (PB_C0 + 1) * kF16BlockBytes, gh, gl, lane);
        t1 = mma3(zero16(), w + (PB_C0 + 2) * kF16BlockBytes, bh[0], bl[0], lane);
        t1 = mma3(t1, w + (PB_C0 + 3) * kF16BlockBytes, gh, gl, lane);
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        u0 = dense64(zero16(), w, PB_C1, t0, t1, lane);
        u1 = dense64(zero16(), w, PB_C1 + 4, t0, t1, lane);
        u0 = relu16(u0); u1 = relu16(u1);
        const f32x16 vdt = dense64(zero16(), w, PB_C2, u0, u1, lane);
        const float view_dep[3] = {sigmoidf(vdt[0]), sigmoidf(vdt[1]), sigmoidf(vdt[2])};

        // ---------------- basis_net: [enc_palette(32) ; diffuse(3)] -> 64 (ELU) -> 15
        split8(xp[0], bh[0], bl[0]);
        split8(xp[1], bh[1], bl[1]);
        h8 dh, dl;
        {
            float v[8] = {diffuse[0], diffuse[1], diffuse[2], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // read by the lower half-wave only (zero weights elsewhere)
            split8(v, dh, dl);
        }
        t0 = mma3(zero16(), w + (PB_B0 + 0) * kF16BlockBytes, bh[0], bl[0], lane);
        t0 = mma3(t0, w + (PB_B0 + 1) * kF16BlockBytes, bh[1], bl[1], lane);
        t0 = mma3(t0, w + (PB_B0 + 2) * kF16BlockBytes, dh, dl, lane);
        t1 = mma3(zero16(), w + (PB_B0 + 3) * kF16BlockBytes, bh[0], bl[0], lane);
        t1 = mma3(t1, w + (PB_B0 + 4) * kF16BlockBytes, bh[1], bl[1], lane);
        t1 = mma3(t1, w + (PB_B0 + 5) * kF16BlockBytes, dh, dl, lane);
        __builtin_amdgcn_sched_barrier(0);
        t0 = elu16(t0); t1 = elu16(t1);
        const f32x16 p = dense64(zero16(), w, PB_B1, t0, t1, lane);   // rows 0..14

        // ---------------- offsets_radiance_net (bias) and omega_net, rows placed in the lower half-wave
        h8 ph, pl;
        split_frag(p, 0, ph, pl);
        f32x16 orr = zero16();
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) orr[j] = pp.or_bias[j];
        }
        orr = mma3(orr, w + PB_OR * kF16BlockBytes, ph, pl, lane);
        const f32x16 om = mma3(zero16(), w + PB_OM * kF16BlockBytes, ph, pl, lane);
        __builtin_amdgcn_sched_barrier(0);

        // ---------------- clip_net (optional): 32 -> 64 -> clip_dim, output rows in the lower half-wave
        f32x16 clip = zero16();
        if (pp.pred_clip) {
            load_enc_blocks(enc_clip, level_stride, row, valid, h, bh, bl);
            t0 = mma3(zero16(), w + (PB_CL0 + 0) * kF16BlockBytes, bh[0], bl[0], lane);
            t0 = mma3(t0, w + (PB_CL0 + 1) * kF16BlockBytes, bh[1], bl[1], lane);
            t1 = mma3(zero16(), w + (PB_CL0 + 2) * kF16BlockBytes, bh[0], bl[0], lane);
            t1 = mma3(t1, w + (PB_CL0 + 3) * kF16BlockBytes, bh[1], bl[1], lane);
            __builtin_amdgcn_sched_barrier(0);
            t0 = relu16(t0); t1 = relu16(t1);
            clip = dense64(zero16(), w, PB_CL1, t0, t1, lane);
        }

        // ---------------- scalar epilogue on the lower half-wave: the palette colour-basis composite
        if (valid && h == 0) {
            float omega[5], osum = 0.0f;
#pragma unroll
            for (int b = 0; b < 5; b++) if (b < nb) { omega[b] = softplusf(om[b]) + 0.05f; osum += omega[b]; }
            const float sp = softplusf(orr[3 * nb]);  // radiance is the LAST of the 3 nb + 1 outputs (palette/renderer.py:471)
            float rgb[3] = {0.0f, 0.0f, 0.0f};
            // aux row: straight to global (one 4-byte store per channel and lane, rows aux_stride apart), or -- when the LDS has room --
            // into this wave's staging slab, from where the whole 32-row tile (contiguous in memory) goes out as 16-byte stores
            float* a = stage_stride ? reinterpret_cast<float*>(w + packed_bytes) + ((size_t)wave * 32 + (lane & 31)) * stage_stride
                                    : aux + (size_t)n * pp.aux_stride;
#pragma unroll
            for (int k = 0; k < 3; k++) { a[k] = diffuse[k] + view_dep[k]; a[3 + k] = view_dep[k]; }   // direct_rgb, view_dep_rgb
#pragma unroll
            for (int b = 0; b < 5; b++) if (b < nb) { omega[b] = omega[b] / osum; a[6 + b] = omega[b]; }
#pragma unroll
            for (int b = 0; b < 5; b++) if (b < nb) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float off = orr[3 * b + k];
                    const float fin = sp * (pp.basis_color[b][k] + pp.offsets_weight * off);
                    const float brgb = omega[b] * fin;
                    rgb[k] += brgb;
                    a[6 + nb + 3 * b + k] = brgb;                                    // basis_rgb
                    a[6 + 4 * nb + 3 * b + k] = pp.basis_color[b][k] + off;          // unscaled_basis_rgb
                }
            }
            int c = 6 + 7 * nb;
#pragma unroll
            for (int k = 0; k < 16; k++) if (k < pp.clip_dim) a[c + k] = pp.pred_clip ? clip[k] : 0.0f;
            for (c += pp.clip_dim; c < pp.aux_stride; c++) a[c] = 0.0f;
            const float sigma = pp.density_scale * __expf(sigma_logit);
            sigmas[n] = sigma;
            if (fuse_composite) a[pp.aux_stride] = 1.0f - __expf(-sigma * deltas[(size_t)n * 2]);   // alpha, exactly as k_frame_composite forms it
#pragma unroll
            for (int k = 0; k < 3; k++) rgbs[(size_t)n * 3 + k] = rgb[k] + pp.view_dep_weight * view_dep[k];
        }
        if (stage_stride) {   // same wave wrote the slab: DS operations of a wave complete in order
            float* slab = reinterpret_cast<float*>(w + packed_bytes) + (size_t)wave * 32 * stage_stride;
            const uint32_t n0 = tile * 256 + wave * 32, nq = (uint32_t)pp.aux_stride / 4, S = (uint32_t)pp.aux_stride;
            const unsigned long long live = __ballot(valid && h == 0);     // rows of dead / out-of-range slots hold stale slab data: skip them
            if (fuse_composite) {
                // aux_map[ray] += sum_k weight_k * row_k: the recurrence of raymarching.cu:1114-1185 (weights from the weights_sum of
                // BEFORE this iteration, stop at a dead row, stop after the sample that sees T < T_thresh), same fmaf order.
                // Leader lane of a ray: weights of its rows, how many count, and the ray id, left in the slab's spare columns.
                if (lane < 32 && (lane % fstep) == 0) {
                    const uint32_t slot = (n0 + lane) / fstep;
                    int cnt = 0, index = 0;
                    if (slot < (uint32_t)ctl->n_alive && ((live >> lane) & 1ull)) {
                        index = rays_alive[slot];
                        float ws = weights_sum[index];
                        for (uint32_t k = 0; k < fstep; k++) {
                            if (!((live >> (lane + k)) & 1ull)) break;
                            float* r = slab + (lane + k) * stage_stride;
                            const float T = 1.0f - ws;
                            const float wgt = r[S] * T;
                            ws += wgt;
                            r[S] = wgt;
                            cnt++;
                            if (T < T_thresh) break;
                        }
                    }
                    slab[lane * stage_stride + S + 1] = __int_as_float(index);
                    slab[lane * stage_stride + S + 2] = __int_as_float(cnt);
                }
                const uint32_t rays_in_tile = 32 / fstep;
                for (uint32_t i = (uint32_t)lane; i < rays_in_tile * nq; i += 64) {
                    const uint32_t ray = i / nq, q = i - ray * nq, base = ray * fstep;
                    const int cnt = __float_as_int(slab[base * stage_stride + S + 2]);
                    if (cnt == 0) continue;
                    const int index = __float_as_int(slab[base * stage_stride + S + 1]);
                    float4* dst = reinterpret_cast<float4*>(aux_map + (size_t)index * pp.aux_stride) + q;
                    float4 acc = *dst;
                    for (int k = 0; k < cnt; k++) {
                        const float* r = slab + (base + k) * stage_stride;
                        const float wgt = r[S];
                        const float4 v = *reinterpret_cast<const float4*>(r + q * 4);
                        acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
                    }
                    *dst = acc;
                }
            } else {
                for (uint32_t i = (uint32_t)lane; i < 32 * nq; i += 64) {
                    const uint32_t row = i / nq, q = i - row * nq;
                    if ((live >> row) & 1ull)
                        *reinterpret_cast<float4*>(aux + (size_t)(n0 + row) * pp.aux_stride + q * 4) = *reinterpret_cast<const float4*>(slab + row * stage_stride + q * 4);
                }
            }
        }
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_palette_field_packed_bytes(int pred_clip) { return (uint64_t)(pred_clip ? PB_END_CLIP : PB_END_NOCLIP) * kF16BlockBytes; }
uint32_t pnr_palette_aux_channels(uint32_t num_basis, uint32_t clip_dim) { return (6 + 7 * num_basis + clip_dim + 3) & ~3u; }

int pnr_palette_field_pack(const pnr_palette_weights* pw, void* packed, pnr_stream_t stream) {
    if (!pw || !packed) return PNR_ERR_INVALID;
    if (pw->num_basis < 1 || pw->num_basis > 5 || pw->clip_dim > 16) return PNR_ERR_UNSUPPORTED;
    if (!pw->sigma0 || !pw->sigma1 || !pw->diff0 || !pw->diff1 || !pw->diff2 || !pw->color0 || !pw->color1 || !pw->color2 || !pw->basis0 || !pw->basis1 ||
        !pw->offsets_radiance || !pw->omega)
        return PNR_ERR_INVALID;
    if (pw->pred_clip && (!pw->clip0 || !pw->clip1)) return PNR_ERR_INVALID;
    PackTable t;
    int q = 0;
    auto add = [&](const float* W, int ld, int nrows, int nrt, int nkb, int colkind, int rowkind) {
        for (int rt = 0; rt < nrt; rt++)
            for (int kb = 0; kb < nkb; kb++) t.b[q++] = PackBlock{W, ld, nrows, rt, colkind, kb, rowkind};
    };
    const int nb = (int)pw->num_basis;
    add(pw->sigma0, 32, 64, 2, 2, COL_LINEAR, ROW_ID);        // PB_S0
    add(pw->sigma1, 64, 16, 1, 4, COL_FRAG, ROW_ID);          // PB_S1
    add(pw->diff0, 15, 64, 2, 1, COL_GEO, ROW_ID);            // PB_D0
    add(pw->diff1, 64, 64, 2, 4, COL_FRAG, ROW_ID);           // PB_D1
    add(pw->diff2, 64, 3, 1, 4, COL_FRAG, ROW_ID);            // PB_D2
    add(pw->color0, 31, 64, 2, 2, COL_SH_GEO, ROW_ID);        // PB_C0
    add(pw->color1, 64, 64, 2, 4, COL_FRAG, ROW_ID);          // PB_C1
    add(pw->color2, 64, 3, 1, 4, COL_FRAG, ROW_ID);           // PB_C2
    add(pw->basis0, 35, 64, 2, 3, COL_ENC_DIFF, ROW_ID);      // PB_B0
    add(pw->basis1, 64, 15, 1, 4, COL_FRAG, ROW_ID);          // PB_B1
    add(pw->offsets_radiance, 15, 3 * nb + 1, 1, 1, COL_FRAG15, ROW_HALF0);  // PB_OR
    add(pw->omega, 15, nb, 1, 1, COL_FRAG15, ROW_HALF0);      // PB_OM
    if (pw->pred_clip) {
        add(pw->clip0, 32, 64, 2, 2, COL_LINEAR, ROW_ID);     // PB_CL0
        add(pw->clip1, 64, (int)pw->clip_dim, 1, 4, COL_FRAG, ROW_HALF0);  // PB_CL1
    }
    t.n = q;
    hipLaunchKernelGGL(k_pack_blocks_f16x3, dim3(cdiv((uint32_t)q * 512, 256)), dim3(256), 0, as_stream(stream), t, static_cast<unsigned char*>(packed));
    return check_launch();
}

int pnr_palette_field_stages_aux(uint32_t aux_stride, int pred_clip) {
    const uint32_t packed_bytes = (uint32_t)pnr_palette_field_packed_bytes(pred_clip);
    return (aux_stride & 3u) == 0 && packed_bytes + (kPalThreads / 64) * 32 * (aux_stride + 4) * 4 <= 160 * 1024;
}

int pnr_palette_field_forward(const pnr_palette_field_args* a, pnr_stream_t stream) {
    if (!a) return PNR_ERR_INVALID;
    if (a->num_basis < 1 || a->num_basis > 5 || a->clip_dim > 16) return PNR_ERR_UNSUPPORTED;
    if (a->aux_stride < 6 + 7 * a->num_basis + a->clip_dim || a->aux_stride > 64) return PNR_ERR_INVALID;
    if (a->B == 0 && !a->ctl) return PNR_OK;
    if (!a->enc || !a->enc_palette || !a->dirs || !a->packed || !a->sigmas || !a->rgbs || !a->aux || !a->basis_color || !a->or_bias) return PNR_ERR_INVALID;
    if (a->pred_clip && !a->enc_clip) return PNR_ERR_INVALID;
    PaletteParams pp;
    for (int b = 0; b < 5; b++)
        for (int k = 0; k < 3; k++) pp.basis_color[b][k] = b < (int)a->num_basis ? fminf(1.0f, fmaxf(0.0f, a->basis_color[b * 3 + k])) : 0.0f;
    for (int j = 0; j < 16; j++) pp.or_bias[j] = j < (int)(3 * a->num_basis + 1) ? a->or_bias[j] : 0.0f;
    pp.density_scale = a->density_scale; pp.offsets_weight = a->offsets_weight; pp.view_dep_weight = a->view_dep_weight;
    pp.nb = (int)a->num_basis; pp.clip_dim = (int)a->clip_dim; pp.pred_clip = a->pred_clip ? 1 : 0; pp.aux_stride = (int)a->aux_stride;
    const uint32_t packed_bytes = (uint32_t)pnr_palette_field_packed_bytes(pp.pred_clip);
    const uint32_t rows_ub = a->B;
    const uint32_t ntiles = cdiv(rows_ub ? rows_ub : 1, 256);
    const uint32_t grid = ntiles < 256u ? ntiles : 256u;  // one persistent 512-thread workgroup per CU (100-116 KiB of LDS)
    constexpr uint32_t kLdsLimit = 160 * 1024;
    static bool attr_set[kMaxDevices] = {};
    if (!ensure_dynamic_lds(k_palette_field_fwd, kLdsLimit, attr_set)) return PNR_ERR_LAUNCH;
    // staging slab for coalesced aux rows: 8 waves x 32 rows x (aux_stride + 4) floats, when it fits next to the weights
    uint32_t stage_stride = pnr_palette_field_stages_aux(a->aux_stride, pp.pred_clip) ? a->aux_stride + 4 : 0;
    const uint32_t lds = packed_bytes + (kPalThreads / 64) * 32 * stage_stride * 4;
    const bool fuse = a->ctl && a->rays_alive && a->weights_sum && a->aux_map;
    hipLaunchKernelGGL(k_palette_field_fwd, dim3(grid), dim3(kPalThreads), lds, as_stream(stream), static_cast<const FrameCtlView*>(a->ctl), a->B,
                       a->enc, a->enc_palette, a->enc_clip, a->level_stride, a->dirs, a->deltas, static_cast<const unsigned char*>(a->packed), packed_bytes, pp,
                       a->sigmas, a->rgbs, a->aux, stage_stride, fuse ? a->rays_alive : nullptr, fuse ? a->weights_sum : nullptr,
                       fuse ? a->aux_map : nullptr, a->T_thresh);
    return check_launch();
}

}  // extern "C"
