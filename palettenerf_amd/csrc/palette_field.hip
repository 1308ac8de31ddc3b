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
//   final_b = softplus(radiance) * (clamp(P_b,0,1) + k_off * offset_b) ; [RegionEdit] ; basis_rgb_b = omega_b * final_b
//   rgb    = sum_b basis_rgb_b + k_vd * vd                              (or the Stylizer's closed form)
// Outputs: sigma * density_scale, rgb, and ONE packed auxiliary row
//   aux = [direct_rgb 3 | view_dep 3 | omega nb | basis_rgb 3nb | unscaled_basis_rgb 3nb | clip clip_dim | 0-pad to x4]
// so that the reference's six composite_rays_flex launches collapse into a single one over the packed row.
//
// Matrix path: written once over field_core.hpp's block interface, instantiated for split-fp16 (3 MFMAs per K = 16 block) and exact
// fp32 (8 v_mfma_f32_32x32x2_f32 per block); activations are chained through the register file exactly as in the NeRF kernel.
// 42 blocks of 2 KiB in LDS (+ the two vector heads, 1.5 KiB) for up to 5 palette bases without a clip head; +1 block for 6..10 bases (offsets_radiance then has up to
// 31 outputs: a second tile), +8 with the clip head (+4 more for clip_dim > 16).  Layers whose outputs feed scalar math (rgb heads,
// offsets/radiance, omega, clip) place their rows in the lower half-wave so no cross-lane traffic is needed.
#include "pnr_common.hpp"
#include "field_core.hpp"
#include "hsv_core.hpp"
#include <string.h>
#include <stdlib.h>
#include <mutex>

namespace pnr {

// ---- block table -------------------------------------------------------------------------------------------------
enum { COL_LINEAR = 0, COL_FRAG = 1, COL_SH_GEO = 2, COL_GEO = 3, COL_ENC_DIFF = 4, COL_FRAG15 = 5 };
enum { ROW_ID = 0, ROW_HALF0 = 1, ROW_HALF0_B = 2 };   // HALF0_B: output rows 16..31 in the lower half-wave's 16 slots (a second tile)
// first block of every layer
enum {
    PB_S0 = 0, PB_S1 = 4, PB_D0 = 8, PB_D1 = 10, PB_C0 = 18, PB_C1 = 22, PB_B0 = 30, PB_B1 = 36, PB_OR = 40, PB_OM = 41,
    PB_OR2 = 42, PB_CL0 = 43, PB_CL1 = 47, PB_CL1B = 51, PB_END = 55
};
// The two 64 -> 3 colour heads (diff_net[2], color_net[2]) are NOT matrix blocks (round 5): as a 32-row MFMA tile 29 of their 32 output rows are padding
// -- 12 matrix instructions (384 matrix-pipe cycles) and four activation splits for three numbers.  They are fp32 dot products on the vector unit instead:
// a lane holds 32 of its sample's 64 hidden features (its two accumulator tiles), multiplies them by its half of the three weight rows with v_pk_fma_f32
// (48 instructions) and v_permlane32_swap adds the two half-waves' partial sums.  Exact fp32 products and sums (the split-fp16 form dropped a 2^-22 term),
// one fixed order in every instantiation.  Their weights follow the blocks: per head [half-wave 2][output 3][tile 2][register 16] floats.
constexpr uint32_t kVecHeadBytes = 2 * 3 * 2 * 16 * 4;     // 768
constexpr uint32_t kVecBytes = 2 * kVecHeadBytes;          // diff_net[2], color_net[2]
constexpr int kPalMaxBlocks = PB_END;
// blocks a model shape needs (= what is packed and staged in LDS)
__host__ __device__ constexpr int pal_blocks(int nb, int clip_dim, int pred_clip) {
    return pred_clip ? (clip_dim > 16 ? PB_END : PB_CL1B) : (nb > 5 ? PB_CL0 : PB_OR2);
}

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

// PREC 1: block = [hi: 64 lanes x 8 halfs][lo: same]; PREC 0: block = [8 steps][64 lanes] fp32 (field_core.hpp: mma_blk)
template <int PREC>
__global__ void __launch_bounds__(256) k_pack_blocks(PackTable t, unsigned char* __restrict__ packed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.n * 512) return;
    const int q = e / 512, lane = (e / 8) & 63, j = e & 7, i = lane & 31, h = lane >> 5;
    const PackBlock b = t.b[q];
    int row = b.rt * 32 + i;
    if (b.rowkind == ROW_HALF0) row = half0_slot(i);
    else if (b.rowkind == ROW_HALF0_B) row = half0_slot(i) < 0 ? -1 : 16 + half0_slot(i);
    const int col = pack_col(b.colkind, b.kb, h, j);
    float v = 0.0f;
    if (b.W && row >= 0 && row < b.nrows && col >= 0 && col < b.ld) v = b.W[(size_t)row * b.ld + col];   // W == NULL: block unused by this shape (zeros)
    if constexpr (PREC == 1) {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* blk = reinterpret_cast<_Float16*>(packed + (size_t)q * kF16BlockBytes);
        blk[lane * 8 + j] = hi;
        blk[512 + lane * 8 + j] = lo;
    } else {
        reinterpret_cast<float*>(packed + (size_t)q * kF16BlockBytes)[j * 64 + lane] = v;
    }
}

__global__ void k_pack_tables(const float* __restrict__ basis_color, const float* __restrict__ or_bias, int nb, float* __restrict__ out /* 64 floats */) {
    const int i = threadIdx.x;
    float v = 0.0f;
    if (i < 30) { if (i < 3 * nb) v = fminf(1.0f, fmaxf(0.0f, basis_color[i])); }
    else if (i < 62) { if (i - 30 < 3 * nb + 1) v = or_bias[i - 30]; }
    out[i] = v;
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
// Addresses are base + a 32-bit byte offset (16 levels x level_stride rows x 8 bytes < 4 GiB: checked by the host): with 64-bit address
// arithmetic the compiler keeps one loop-invariant 64-bit `level * level_stride` base per level and table in scalar registers -- 64 of the
// kernel's ~100 -- and spills them to vector-register lanes (v_readlane in the tile loop).
__device__ __forceinline__ void load_enc_raw(const float* __restrict__ enc, uint32_t level_stride, uint32_t row, bool valid, int h, float x[2][8]) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(enc);
    // the eight loads go out together and return into registers of their own (pnr_common.hpp: load8_fresh explains why); `row` exists (clamped by the caller)
    uint32_t off[8];
    f32x2 v[8];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) off[4 * kb + q] = ((uint32_t)(8 * kb + 4 * h + q) * level_stride + row) * 8u;
    asm volatile("" : "+v"(off[0]), "+v"(off[1]), "+v"(off[2]), "+v"(off[3]), "+v"(off[4]), "+v"(off[5]), "+v"(off[6]), "+v"(off[7]));
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = *(GlobalPtr<f32x2>::type)(uintptr_t)(base + off[i]);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]));
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) { x[kb][2 * q] = valid ? v[4 * kb + q].x : 0.0f; x[kb][2 * q + 1] = valid ? v[4 * kb + q].y : 0.0f; }
}
// 64 -> N layer from two activation tiles: 4 k-blocks starting at block q0
template <int PREC, bool CHECK>
__device__ __forceinline__ f32x16 dense64(f32x16 acc, const unsigned char* __restrict__ w, int q0, const f32x16& a0, const f32x16& a1, int lane, SplitWatch<CHECK>* sw) {
    acc = mma_blk<PREC>(acc, w + (q0 + 0) * kF16BlockBytes, frag_op<PREC, CHECK>(a0, 0, sw), lane);
    acc = mma_blk<PREC>(acc, w + (q0 + 1) * kF16BlockBytes, frag_op<PREC, CHECK>(a0, 1, sw), lane);
    acc = mma_blk<PREC>(acc, w + (q0 + 2) * kF16BlockBytes, frag_op<PREC, CHECK>(a1, 0, sw), lane);
    acc = mma_blk<PREC>(acc, w + (q0 + 3) * kF16BlockBytes, frag_op<PREC, CHECK>(a1, 1, sw), lane);
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

// 64 -> 64 layer: BOTH output tiles per k-block, so that a block's activation split (8 registers) is formed once and is dead before the next one --
// two dense64 calls either keep all four splits alive across both (32 registers: the difference between three and four waves per SIMD) or form them
// twice.  Per accumulator the blocks arrive in the same order as in dense64: same bits.
template <int PREC, bool CHECK>
__device__ __forceinline__ void dense64x2(f32x16& o0, f32x16& o1, const unsigned char* __restrict__ w, int q0, const f32x16& a0, const f32x16& a1, int lane, SplitWatch<CHECK>* sw) {
    o0 = zero16(); o1 = zero16();
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const BOp<PREC> b = frag_op<PREC, CHECK>(kb < 2 ? a0 : a1, kb & 1, sw);
        o0 = mma_blk<PREC>(o0, w + (q0 + kb) * kF16BlockBytes, b, lane);
        o1 = mma_blk<PREC>(o1, w + (q0 + 4 + kb) * kF16BlockBytes, b, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void scale8x2(float x[2][8], float s) {
#pragma unroll
    for (int j = 0; j < 8; j++) { x[0][j] *= s; x[1][j] *= s; }
}
__device__ __forceinline__ f32x16 scale16(f32x16 v, float s) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] *= s;
    return v;
}

constexpr int kMaxNb = PNR_MAX_BASIS;
// per-model constants the epilogue reads.  They live at the end of the packed blob (written by pnr_palette_field_pack from the model's
// device parameters) and are staged into LDS with the weights: as kernel arguments these 62 values sat in scalar registers for the whole
// tile loop, were spilled to vector-register lanes and read back with v_readlane in the hot loop
struct PaletteTables {
    float basis_color[kMaxNb][3];   // clamped to [0,1] (palette/renderer.py:480)
    float or_bias[32];              // offsets_radiance_net.bias (3 nb + 1 entries, zero padded)
    float pad[2];
};
static_assert(sizeof(PaletteTables) == 64 * 4, "PaletteTables is 64 floats");
constexpr uint32_t kTablesBytes = sizeof(PaletteTables);
struct PaletteParams {
    float density_scale, offsets_weight, view_dep_weight;
    int nb, clip_dim, pred_clip, aux_stride;
    float enc_scale[3];             // power-of-two prescales of enc / enc_palette / enc_clip (split-fp16 path only; 1 = none)
    float enc_scale_inv[3];         // their exact reciprocals, formed on the host (a division in the kernel is a dozen instructions the compiler hoists and keeps)
};
// RegionEdit / Stylizer parameters (pnr_palette_edit) as the kernel reads them from device memory
struct EditParams {
    float delta_hsv[kMaxNb][3];
    float mean_xyz[3], std_xyz;
    float mean_clip[PNR_MAX_CLIP], std_clip;
    int has_mean_xyz, has_mean_clip, weight_mode;
    float dI[kMaxNb], dP[kMaxNb][3], ddelta[kMaxNb][3][3];
};

constexpr int kPalThreads = 512;
// Wide kernels: a request a tile used to wait for where it needed it is issued at the top of the tile --
//   PNR_PAL_EARLY_RAY  a ray leader's slot -> ray id (and then its weights_sum) for the compositing step.
// (Round 4 also tried the second table's rows through global_load_lds into the then-idle staging slab, PNR_PAL_EARLY_PAL: slower, EXPERIMENTS.md; the slab
// is gone since round 5 and that code with it.)
#ifndef PNR_PAL_WAVE_MAJOR
#define PNR_PAL_WAVE_MAJOR 1    // wave tiles dealt wave-major (see the tile loop); 0 = workgroup-major, for the A/B
#endif
#ifndef PNR_PAL_EARLY_RAY
#define PNR_PAL_EARLY_RAY 1     // garden 13.2 -> 13.0 ms (162 registers, no scratch)
#endif
// Round 5.  What the parts of this kernel cost alone (profiles/r05_pal_parts.txt, stand-alone op, 1.09 M rows, timing-only PNR_PAL_FAKE builds): the matrix
// phase 171 us of 213, everything behind it (scalar epilogue, staging, stores) ~40 us; and what they need: the matrix phase 118 registers, the old tail 157
// plus a 4.6 KB staging slab per wave -- the tail, not the layers, held the kernel at three waves per SIMD.  The specialised 4-basis kernels ("wide":
// NB == 4, PNR_PAL_WIDE_WAVES waves, no clip head, rows of 36 floats) therefore take a tail of their own that stages NOTHING in LDS:
//   * a sample's 34-value aux row stays in its lane's registers; with one sample per ray (every heavy launch) the lane itself does the ray's
//     aux_map read-modify-write, nine 16-byte loads in flight at once; with 2 .. 8 samples per ray the ray's leader lane pulls the following lanes'
//     rows through ds_bpermute and runs the same fmaf chain in the same order (bit-identical to the staged form either way);
//   * the ray state (rays_t, depth, image) is requested when the matrix phase ends and used behind the scalar epilogue;
//   * PNR_PAL_EARLY_ENC: the tile's encoder rows and directions are requested BEFORE the slot's `delta` has decided whether the row is alive (one trip
//     to memory instead of two dependent ones; a dead lane's loads are wasted), and the delta pair stays in two registers for the epilogue's alpha and
//     the ray-state step (it was loaded three times).
// LDS: the weights + 384 bytes per wave, so 16 waves (four per SIMD) fit where 12 slabs did not leave room for a thirteenth.
#ifndef PNR_PAL_EARLY_ENC
#define PNR_PAL_EARLY_ENC 1
#endif
#ifndef PNR_PAL_FAKE
#define PNR_PAL_FAKE 0          // timing-only builds (WRONG results): 1 = no matrix phase (the loads stay), 2 = nothing behind the matrix phase (no epilogue, composite, stores), 4 = nothing behind it on every second tile
#endif
#ifndef PNR_PAL_WIDE_WAVES
#define PNR_PAL_WIDE_WAVES 16   // waves per workgroup of the specialised 4-basis kernels ("wide": no clip head, rows of 36 floats, slabs in LDS).  Experiment builds: 8
#endif
#ifndef PNR_PAL_EARLY_ENC
#define PNR_PAL_EARLY_ENC 1
#endif


// ctl == nullptr: rows = B (stand-alone op).  Otherwise rows = n_alive * n_step of the frame control block and
// dead slots (delta == 0) are skipped.
struct FrameCtlView { int32_t n_alive, n_step, step, done; };

// frame loop, optional: the ray state the iteration's compositing step updates.  With it (and the aux rows staged) the kernel does
// k_frame_composite's second phase itself -- weights_sum / depth / image / rays_t of every ray, the alive list's holes and the per-chunk
// survivor counts -- for any number of samples per ray: a wave tile then holds floor(32 / n_step) whole rays.  All NULL: as before.
struct RayState { float* rays_t; float* weights_sum; float* depth; float* image; int32_t* rays_alive; int32_t* counts_cur; };

// torch.lerp(start, end, weight) for fp32 (ATen/native/Lerp.h): the branch keeps both ends exact
__device__ __forceinline__ float torch_lerp(float a, float b, float w) { const float d = b - a; return w < 0.5f ? a + w * d : b - d * (1.0f - w); }

// EDIT: 0 plain composite, 1 RegionEdit, 2 Stylizer (pnr_palette_edit.mode); separate instantiations keep the HSV / double-fmod code
// and the extra parameters out of the plain kernel.  CHECK: watch the split operands for magnitudes beyond fp16's range (SplitWatch)
// NB: 0 = any number of bases (loops predicated up to PNR_MAX_BASIS), otherwise exactly NB bases (the shipped default 4: no predication,
// no second offsets_radiance tile)
// WAVES: waves per workgroup (8, or 12 = three per SIMD where registers and LDS allow it); a workgroup tile is WAVES x 32 samples
// All arguments in one struct.  The kernel takes the ones its matrix phase needs from the parameter itself; the fifteen pointers only the epilogue
// uses are read from the kernel-argument segment where the epilogue starts, through a pointer the compiler cannot see through -- held from the
// kernel's entry they were 30 scalar registers that lived through the whole tile loop in a kernel that spills 120 of them to vector-register lanes
// (a v_readlane per use, ~160 per tile).
struct PalArgs {
    const FrameCtlView* ctl; uint32_t B; const float* enc; const float* enc_pal; const float* enc_clip; uint32_t level_stride; const float* dirs;
    const float* deltas; const unsigned char* packed; uint32_t packed_bytes; PaletteParams pp; float* sigmas; float* rgbs; float* aux; uint32_t stage_stride;
    const int32_t* rays_alive; const float* weights_sum; float* aux_map; float T_thresh; const float* xyzs; const EditParams* ep; int32_t* overflow_flag;
    uint32_t* tile_counter; RayState rs;
};
typedef const __attribute__((address_space(4))) PalArgs* PalArgsK;
#ifndef PNR_PAL_KARGS
#define PNR_PAL_KARGS 1     // 0: the loads are visible to the compiler again, which hoists them to the kernel's entry (the A/B of this change)
#endif
__device__ __forceinline__ PalArgsK launder(PalArgsK p) {
#if PNR_PAL_KARGS
    asm volatile("" : "+s"(p));
#endif
    return p;
}

// PNR_PAL_TIMING (diagnostic build, profiles/pal_timing.py): wave-level time per phase of a tile, summed over all waves and tiles of every launch since the
// last reset.  Stamps are the 100 MHz wall clock; the stamp behind a load phase first waits for the loads, so "wait" phases hold the exposed latency.
#ifdef PNR_PAL_TIMING
enum { PT_TOP = 0, PT_ENC_WAIT, PT_SIGMA, PT_DIFF, PT_COLOR, PT_PAL_WAIT, PT_BASIS, PT_EPILOGUE, PT_LEADER, PT_AUXMAP, PT_RAYSTATE, PT_SKIP, PT_SETUP, PT_PREFETCH, PT_ACC_WAIT, PT_ACC, PT_N };
__device__ unsigned long long g_pal_timing[PT_N + 3];   // + tiles, waves, kernel wall (first wave start .. last wave end is not tracked: wave residence sum)
#define PAL_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PAL_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PAL_T(i) do {} while (0)
#define PAL_WAIT_VM() do {} while (0)
#endif

template <int PREC, int EDIT, bool CHECK, int NB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_palette_field_fwd(PalArgs A_) {
#ifdef PNR_PAL_TIMING
    unsigned long long tacc[PT_N] = {};
    unsigned long long tlast = wall_clock64();
    const unsigned long long tstart = tlast;
    unsigned long long ntiles = 0;
#endif
    const PalArgsK ka = (PalArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const FrameCtlView* __restrict__ ctl = A_.ctl;
    const uint32_t B_arg = A_.B;
    const float* __restrict__ enc = A_.enc;
    const float* __restrict__ enc_pal = A_.enc_pal;
    const float* __restrict__ enc_clip = A_.enc_clip;
    const uint32_t level_stride = A_.level_stride;
    const float* __restrict__ dirs = A_.dirs;
    const float* __restrict__ deltas = A_.deltas;
    const unsigned char* __restrict__ packed = A_.packed;
    const uint32_t packed_bytes = A_.packed_bytes;
    const PaletteParams& pp = A_.pp;
    const uint32_t stage_stride = A_.stage_stride;
    const int32_t* __restrict__ rays_alive = A_.rays_alive;
    const float* __restrict__ weights_sum = A_.weights_sum;
    uint32_t* __restrict__ tile_counter = A_.tile_counter;
    const bool has_aux_map = A_.aux_map != nullptr, has_ray_state = A_.rs.rays_t != nullptr;
    if (ctl && ctl->done) return;
    // the control block does not change while this kernel runs: read once (a use inside the tile loop would be a global load per use)
    const uint32_t n_alive_k = ctl ? (uint32_t)__builtin_amdgcn_readfirstlane(ctl->n_alive) : 0u, n_step_k = ctl ? (uint32_t)__builtin_amdgcn_readfirstlane(ctl->n_step) : 0u;
    const uint32_t B = ctl ? n_alive_k * n_step_k : B_arg;
    // rows of a ray are consecutive.  Whole rays per wave tile (rpw rows of its 32) when the kernel also composites the ray state; otherwise 32 rows,
    // and the aux composite runs here only with 1, 2, 4 or 8 samples per ray (a ray's rows then sit inside one wave tile anyway)
    const bool ray_tiles = stage_stride && ctl && has_aux_map && has_ray_state;
    const uint32_t rpw = ray_tiles ? (32u / n_step_k) * n_step_k : 32u;
    const uint32_t nwt = (B + rpw - 1) / rpw;    // wave tiles of this launch
    if (blockIdx.x >= nwt) return;               // (wave 0 of workgroup b takes wave tile b first: a workgroup beyond nwt has nothing at all)
    extern __shared__ unsigned char w[];
    for (uint32_t i = threadIdx.x * 16; i < packed_bytes; i += WAVES * 64 * 16) lds_copy16(&packed[i], &w[i]);   // weights + the PaletteTables behind them
    lds_copy_wait();
    __syncthreads();
    PAL_T(PT_SETUP);
    const int lane_k = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = NB ? NB : pp.nb;
    constexpr int kLoopNb = NB ? NB : kMaxNb;
    // rows of a ray are consecutive; with 1, 2, 4 or 8 samples per ray they sit inside one 32-row wave tile and the aux composite
    // can run here (fstep = samples per ray), otherwise the composite launch does it
    const uint32_t fstep = (stage_stride && ctl && has_aux_map && n_step_k <= 8 && (ray_tiles || (32 % n_step_k) == 0)) ? n_step_k : 0u;
    const bool fuse_composite = fstep != 0;
    // x / fstep for the small x of a tile (lane numbers, rows per tile) as (x * fM) >> 16 -- exact for x < 8192, fstep <= 8 -- and the per-tile quantities that
    // do not depend on the lane as scalars: the compiler's own expansion of an integer division is ~20 instructions whose lane-dependent intermediate values
    // it hoists out of the tile loop and then keeps (or spills) across the whole matrix phase
    const uint32_t fM = fstep ? 65536u / fstep + 1u : 0u;
    const uint32_t rays_pt = fstep ? (rpw * fM) >> 16 : 0u;      // whole rays per wave tile
    // Work is handed out per 32-sample wave tile: the static persistent schedule (workgroup b takes tiles b, b + grid, ...), or -- an experiment
    // kept behind pnr_set_option("dynamic_tiles") -- every wave fetching its next wave tile from a device counter (tile_counter, zeroed by the
    // iteration's march launch).  The idea: a launch of 11.08 workgroup tiles per CU takes the time of 12 with the static schedule.  Measured:
    // garden frame 14.6 -> 20.2 ms -- 34 k waves queue on one counter and a workgroup's waves no longer read neighbouring rows.  Off.
    for (uint32_t it = 0;; it++) {
        uint32_t wt;
        if (tile_counter) {
            uint32_t got = 0;
            if (lane_k == 0) got = atomicAdd(tile_counter, 1u);
            wt = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        } else {
            // wave-major: round `it` deals wave tile (it * WAVES + wave) * grid + b to wave `wave` of workgroup b, so the LAST, partial round of a
            // launch is spread one tile per workgroup (and, past `grid` tiles, one per SIMD: wave w sits on SIMD w % 4) instead of filling all
            // WAVES waves of its first few workgroups -- garden: 11.09 rounds cost 11 + 1/3 tile times instead of 12; an eighth of a frame:
            // 1.4 rounds cost 1 + 2/3 instead of 2.  (Workgroup-major, rounds 1-3: (b + it * grid) * WAVES + wave.)
            if constexpr (PNR_PAL_WAVE_MAJOR != 0) wt = (it * WAVES + (uint32_t)wave) * gridDim.x + blockIdx.x;
            else wt = (blockIdx.x + it * gridDim.x) * WAVES + wave;
        }
        if (wt >= nwt) break;
        // (the lane number goes through an opaque move once per tile: what is derived from it below -- row numbers, leader flags, LDS addresses, bit masks -- is
        // then formed where it is used, a few instructions, instead of being hoisted out of the loop into registers that live through the matrix phase)
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5;
        const uint32_t lane_q = ((uint32_t)(lane & 31) * fM) >> 16, lane_k_in_ray = (uint32_t)(lane & 31) - lane_q * fstep;   // (lane & 31) / fstep, % fstep
        const uint32_t n = wt * rpw + (lane & 31);
        const bool mine = (uint32_t)(lane & 31) < rpw && n < B;
        // the slot's (delta_0, delta_1): delta_0 == 0 marks a dead slot; kept for the epilogue's alpha and the ray-state step's t advance
        f32x2 dl = {1.0f, 0.0f};
        if (deltas && mine) dl = *reinterpret_cast<const f32x2*>(deltas + (size_t)n * 2);
        constexpr bool kEarlyEnc = PNR_PAL_EARLY_ENC != 0 && WAVES == PNR_PAL_WIDE_WAVES;
        float xs[2][8], xp[2][8];
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if constexpr (kEarlyEnc) {   // requested before `dl` has arrived: rows of dead lanes are read too and replaced by zeros below
            const uint32_t row_e = n < B ? n : (B - 1);
            load_enc_raw(enc, level_stride, row_e, true, h, xs);
            dx = dirs[(size_t)row_e * 3]; dy = dirs[(size_t)row_e * 3 + 1]; dz = dirs[(size_t)row_e * 3 + 2];
        }
        const bool valid = mine && dl.x != 0.0f;
        if (!__any(valid)) {
            // nothing to evaluate -- but with the ray state composited here, the rays of this wave tile (all their rows dead) must still leave the alive list
            if (has_ray_state) {
                const uint32_t l = (uint32_t)(lane & 31);      // (has_ray_state implies the fused tail: fstep == n_step)
                if (h == 0 && l < rpw && lane_k_in_ray == 0 && wt * rays_pt + lane_q < n_alive_k) launder(ka)->rs.rays_alive[wt * rays_pt + lane_q] = -1;
            }
            PAL_T(PT_SKIP);
            continue;
        }
#ifdef PNR_PAL_TIMING
        ntiles++;
#endif
        const uint32_t row = n < B ? n : (B - 1);

        SplitWatch<CHECK> sw_, *sw = &sw_;
        // all global reads of the tile up front
        // (the 12-wave variant has three waves per SIMD to hide a load behind and 168 registers: it fetches the second table's features where they are used)
        if constexpr (kEarlyEnc) {
            // Dead lanes keep what they loaded (rows that exist: the caller-clamped row): a sample is a COLUMN of every matrix product and the cross-lane steps
            // pair the two half-wave lanes of ONE sample, so a dead lane's values reach nobody, and nothing it computes is stored.  Only the operand watch of
            // the CHECK instantiations looks at every lane: there they are zeroed as before.  (35 selects per tile.)
            if constexpr (CHECK) {
#pragma unroll
                for (int j = 0; j < 8; j++) { xs[0][j] = valid ? xs[0][j] : 0.0f; xs[1][j] = valid ? xs[1][j] : 0.0f; }
                dx = valid ? dx : 0.0f; dy = valid ? dy : 0.0f; dz = valid ? dz : 0.0f;
            }
        } else {
            load_enc_raw(enc, level_stride, row, valid, h, xs);
        }
        if constexpr (WAVES != PNR_PAL_WIDE_WAVES) load_enc_raw(enc_pal, level_stride, row, valid, h, xp);
        if constexpr (!kEarlyEnc) {
            if (valid) { dx = dirs[(size_t)row * 3]; dy = dirs[(size_t)row * 3 + 1]; dz = dirs[(size_t)row * 3 + 2]; }
        }
        // the compositing step's first dependent load (slot -> ray id), requested now; its second (the ray's weights_sum) once sigma_net is done
        const bool early_ray = PNR_PAL_EARLY_RAY && WAVES == PNR_PAL_WIDE_WAVES && fuse_composite;
        const uint32_t slot_t = wt * rays_pt + lane_q;      // the alive-list slot of this lane's ray
        const bool lead_lane = early_ray && (uint32_t)lane < rpw && lane_k_in_ray == 0 && slot_t < n_alive_k;
        int early_index = 0;
        float early_ws = 0.0f;
        if (lead_lane) early_index = rays_alive[slot_t];

        PAL_T(PT_TOP);
        PAL_WAIT_VM();
        PAL_T(PT_ENC_WAIT);
#if PNR_PAL_FAKE & 1
        // (timing only) stand-ins formed from the loaded rows; the second table's rows and the slab request stay
        constexpr bool kWide = NB == 4 && WAVES == PNR_PAL_WIDE_WAVES;
        if (lead_lane) early_ws = weights_sum[early_index];
        if constexpr (WAVES == PNR_PAL_WIDE_WAVES) load_enc_raw(enc_pal, level_stride, row, valid, h, xp);
        const float sigma_logit = xs[0][0] + xp[0][0];
        const float diffuse[3] = {xs[0][1], xs[0][2], xs[0][3]}, view_dep[3] = {xp[0][1] + dx, xp[0][2] + dy, xp[0][3] + dz};
        f32x16 orr = zero16(), orr2 = zero16(), om = zero16(), clip = zero16(), clip2 = zero16();
#pragma unroll
        for (int j = 0; j < 8; j++) { orr[j] = xs[1][j]; orr[8 + j] = xp[1][j]; om[j] = xs[0][j]; }
        uint32_t toff = packed_bytes - kTablesBytes;
        asm volatile("" : "+s"(toff));
        const bool pred_clip = false;
#else
        // ---------------- sigma_net (prescaled by a power of two when the table's entries are tiny: undone exactly on its 16 outputs)
        const bool pre_s = PREC != 0 && pp.enc_scale[0] != 1.0f, pre_p = PREC != 0 && pp.enc_scale[1] != 1.0f, pre_c = PREC != 0 && pp.enc_scale[2] != 1.0f;
        if (pre_s) scale8x2(xs, pp.enc_scale[0]);
        // F16X2 rounds the activations of the COLOUR heads only: sigma_net keeps the split form (PS), so densities, alphas, the transmittance cut-off and
        // with them the march are those of F16X3 bit for bit, and a pixel's error is bounded by the per-sample colour error whatever the ray's length
        constexpr int PS = PREC == 2 ? 1 : PREC;
        f32x16 t0 = zero16(), t1 = zero16();
        {
            const BOp<PS> b0 = make_op<PS, CHECK>(xs[0], sw), b1 = make_op<PS, CHECK>(xs[1], sw);
            t0 = mma_blk<PS>(t0, w + (PB_S0 + 0) * kF16BlockBytes, b0, lane);
            t0 = mma_blk<PS>(t0, w + (PB_S0 + 1) * kF16BlockBytes, b1, lane);
            t1 = mma_blk<PS>(t1, w + (PB_S0 + 2) * kF16BlockBytes, b0, lane);
            t1 = mma_blk<PS>(t1, w + (PB_S0 + 3) * kF16BlockBytes, b1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        f32x16 g = dense64<PS, CHECK>(zero16(), w, PB_S1, t0, t1, lane, sw);   // rows 0..15: sigma logit, geo_feat 1..15
        if (pre_s) g = scale16(g, pp.enc_scale_inv[0]);
        const float sigma_logit = g[0];
        const BOp<PREC> geo = frag_op<PREC, CHECK>(g, 0, sw);                           // the geo k-block, shared by diff_net and color_net
        PAL_T(PT_SIGMA);
        if (lead_lane) early_ws = weights_sum[early_index];
        constexpr bool kWide = NB == 4 && WAVES == PNR_PAL_WIDE_WAVES;   // the slab-free tail (see the top of the file)
        PAL_T(PT_PREFETCH);

        // ---------------- diff_net: 15 -> 64 -> 64 -> 3
        t0 = mma_blk<PREC>(zero16(), w + (PB_D0 + 0) * kF16BlockBytes, geo, lane);
        t1 = mma_blk<PREC>(zero16(), w + (PB_D0 + 1) * kF16BlockBytes, geo, lane);
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        f32x16 u0, u1;
        dense64x2<PREC, CHECK>(u0, u1, w, PB_D1, t0, t1, lane, sw);
        u0 = relu16(u0); u1 = relu16(u1);
        // (the heads' weights sit behind the blocks; the address goes through the per-tile lane number so that the 24 reads are not hoisted out of the tile loop)
        const unsigned char* vec_heads = w + (packed_bytes - kTablesBytes - kVecBytes) + (uint32_t)h * (kVecHeadBytes / 2);
        float dif[3];
        head3_valu(vec_heads, u0, u1, dif);
        const float diffuse[3] = {sigmoidf(dif[0]), sigmoidf(dif[1]), sigmoidf(dif[2])};

        PAL_T(PT_DIFF);
        // ---------------- color_net (view dependent): [SH16 ; geo15] -> 64 -> 64 -> 3
        {
            float sh[16], v[8];
            sh_eval<4>(dx, dy, dz, sh);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = select_half(h, sh[j], sh[8 + j]);
            const BOp<PREC> shb = make_op<PREC, CHECK>(v, sw);
            t0 = mma_blk<PREC>(zero16(), w + (PB_C0 + 0) * kF16BlockBytes, shb, lane);
            t0 = mma_blk<PREC>(t0, w + (PB_C0 + 1) * kF16BlockBytes, geo, lane);
            t1 = mma_blk<PREC>(zero16(), w + (PB_C0 + 2) * kF16BlockBytes, shb, lane);
            t1 = mma_blk<PREC>(t1, w + (PB_C0 + 3) * kF16BlockBytes, geo, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        t0 = relu16(t0); t1 = relu16(t1);
        dense64x2<PREC, CHECK>(u0, u1, w, PB_C1, t0, t1, lane, sw);
        u0 = relu16(u0); u1 = relu16(u1);
        float vdt[3];
        head3_valu(vec_heads + kVecHeadBytes, u0, u1, vdt);
        const float view_dep[3] = {sigmoidf(vdt[0]), sigmoidf(vdt[1]), sigmoidf(vdt[2])};

        PAL_T(PT_COLOR);
        // ---------------- basis_net: [enc_palette(32) ; diffuse(3)] -> 64 (ELU) -> 15
        {
            if constexpr (WAVES == PNR_PAL_WIDE_WAVES) {
                {
                    load_enc_raw(enc_pal, level_stride, row, valid || (kEarlyEnc && !CHECK), h, xp);     // (dead lanes: see the first table's rows above)
                }
            }
            PAL_WAIT_VM();
            PAL_T(PT_PAL_WAIT);
            const float ps = pre_p ? pp.enc_scale[1] : 1.0f;   // the whole 35-wide input row is scaled; the ELU needs the true pre-activations back
            if (pre_p) scale8x2(xp, ps);
            const BOp<PREC> b0 = make_op<PREC, CHECK>(xp[0], sw), b1 = make_op<PREC, CHECK>(xp[1], sw);
            const float v[8] = {diffuse[0] * ps, diffuse[1] * ps, diffuse[2] * ps, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // read by the lower half-wave only (zero weights elsewhere)
            const BOp<PREC> db = make_op<PREC, CHECK>(v, sw);
            t0 = mma_blk<PREC>(zero16(), w + (PB_B0 + 0) * kF16BlockBytes, b0, lane);
            t0 = mma_blk<PREC>(t0, w + (PB_B0 + 1) * kF16BlockBytes, b1, lane);
            t0 = mma_blk<PREC>(t0, w + (PB_B0 + 2) * kF16BlockBytes, db, lane);
            t1 = mma_blk<PREC>(zero16(), w + (PB_B0 + 3) * kF16BlockBytes, b0, lane);
            t1 = mma_blk<PREC>(t1, w + (PB_B0 + 4) * kF16BlockBytes, b1, lane);
            t1 = mma_blk<PREC>(t1, w + (PB_B0 + 5) * kF16BlockBytes, db, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pre_p) { t0 = scale16(t0, pp.enc_scale_inv[1]); t1 = scale16(t1, pp.enc_scale_inv[1]); }
        t0 = elu16(t0); t1 = elu16(t1);
        const f32x16 p = dense64<PREC, CHECK>(zero16(), w, PB_B1, t0, t1, lane, sw);   // rows 0..14

        // ---------------- offsets_radiance_net (bias) and omega_net, rows placed in the lower half-wave (outputs 16.. in a second tile)
        const BOp<PREC> pb = frag_op<PREC, CHECK>(p, 0, sw);
        f32x16 orr = zero16(), orr2 = zero16();
        // the tables are read from LDS where they are used: the offset goes through an opaque asm so that the 62 loads are not hoisted out of
        // the tile loop (where they would occupy 62 vector registers for the whole kernel)
        uint32_t toff = packed_bytes - kTablesBytes;
        asm volatile("" : "+s"(toff));
        const PaletteTables& T = *reinterpret_cast<const PaletteTables*>(w + toff);
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) { orr[j] = T.or_bias[j]; orr2[j] = T.or_bias[16 + j]; }
        }
        orr = mma_blk<PREC>(orr, w + PB_OR * kF16BlockBytes, pb, lane);
        if (NB ? NB > 5 : nb > 5) orr2 = mma_blk<PREC>(orr2, w + PB_OR2 * kF16BlockBytes, pb, lane);
        const f32x16 om = mma_blk<PREC>(zero16(), w + PB_OM * kF16BlockBytes, pb, lane);
        __builtin_amdgcn_sched_barrier(0);

        // ---------------- clip_net (optional): 32 -> 64 -> clip_dim, output rows in the lower half-wave
        f32x16 clip = zero16(), clip2 = zero16();
        // (the 12-wave kernels never carry a clip head -- its eight blocks and 16 more columns per staged row do not fit next to 12 slabs, see the launcher --
        // so there the head's two accumulator tiles and every branch on it go at compile time)
        constexpr bool kClipHead = WAVES != PNR_PAL_WIDE_WAVES;
        const bool pred_clip = kClipHead && pp.pred_clip;
        if (pred_clip) {
            float xc[2][8];
            load_enc_raw(enc_clip, level_stride, row, valid, h, xc);
            if (pre_c) scale8x2(xc, pp.enc_scale[2]);
            const BOp<PREC> b0 = make_op<PREC, CHECK>(xc[0], sw), b1 = make_op<PREC, CHECK>(xc[1], sw);
            t0 = mma_blk<PREC>(zero16(), w + (PB_CL0 + 0) * kF16BlockBytes, b0, lane);
            t0 = mma_blk<PREC>(t0, w + (PB_CL0 + 1) * kF16BlockBytes, b1, lane);
            t1 = mma_blk<PREC>(zero16(), w + (PB_CL0 + 2) * kF16BlockBytes, b0, lane);
            t1 = mma_blk<PREC>(t1, w + (PB_CL0 + 3) * kF16BlockBytes, b1, lane);
            __builtin_amdgcn_sched_barrier(0);
            t0 = relu16(t0); t1 = relu16(t1);
            clip = dense64<PREC, CHECK>(zero16(), w, PB_CL1, t0, t1, lane, sw);
            if (pp.clip_dim > 16) clip2 = dense64<PREC, CHECK>(zero16(), w, PB_CL1B, t0, t1, lane, sw);
            if (pre_c) { clip = scale16(clip, pp.enc_scale_inv[2]); clip2 = scale16(clip2, pp.enc_scale_inv[2]); }
        }

#endif
#if PNR_PAL_FAKE & 2
        {   // (timing only) keep the matrix phase's results alive, store nothing
            float chk = sigma_logit + diffuse[0] + diffuse[1] + diffuse[2] + view_dep[0] + view_dep[1] + view_dep[2] + early_ws;
#pragma unroll
            for (int j = 0; j < 16; j++) chk += orr[j] + om[j] + orr2[j] + clip[j] + clip2[j];
            if (chk == 123456.789f) launder(ka)->sigmas[n] = chk + dl.y;
        }
#else
#if PNR_PAL_FAKE & 4
        if (it & 1u) {   // (timing only) every second tile of a wave skips everything behind the matrix phase: what half the tail's work would buy
            float chk = sigma_logit + diffuse[0] + diffuse[1] + diffuse[2] + view_dep[0] + view_dep[1] + view_dep[2] + early_ws;
#pragma unroll
            for (int j = 0; j < 16; j++) chk += orr[j] + om[j];
            if (chk == 123456.789f) launder(ka)->sigmas[n] = chk + dl.y;
            continue;
        }
#endif
        PAL_T(PT_BASIS);
        // ---------------- scalar epilogue on the lower half-wave: the palette colour-basis composite
        asm volatile("" : "+s"(toff));
        // the epilogue's arguments, from the kernel-argument segment (scalar loads here, not registers held since the kernel's entry)
        const PalArgsK kc = launder(ka);
        float* __restrict__ const sigmas = kc->sigmas;
        float* __restrict__ const rgbs = kc->rgbs;
        float* __restrict__ const aux = kc->aux;
        float* __restrict__ const aux_map = kc->aux_map;
        const float T_thresh = kc->T_thresh;
        const float* __restrict__ const xyzs = kc->xyzs;
        const EditParams* __restrict__ const ep = kc->ep;
        int32_t* __restrict__ const overflow_flag = kc->overflow_flag;
        const RayState rs = {kc->rs.rays_t, kc->rs.weights_sum, kc->rs.depth, kc->rs.image, kc->rs.rays_alive, kc->rs.counts_cur};
        // Leader lane of a ray (the compositing step's weights): the recurrence of raymarching.cu:1114-1185 over the ray's rows -- weights from the
        // weights_sum of BEFORE this iteration, stop at a dead row, stop after the sample that sees T < T_thresh -- leaving per row the weight and per
        // leader the ray id and the number of rows that count in the wave's spare LDS columns (ex).  It needs the rows' alphas only, so the wide kernels run
        // it IN FRONT of the scalar epilogue: the 34 values of a row then never live across this piece of control flow.
        float* const ex = reinterpret_cast<float*>(w + packed_bytes) + (kWide ? (size_t)0 : (size_t)WAVES * 32 * stage_stride) + (size_t)wave * 96;
        const uint32_t live = (uint32_t)__ballot(valid && h == 0);     // (bits 0 .. 31: the rows) rows of dead / out-of-range slots hold no row
        const uint32_t slot = wt * rays_pt + lane_q;
        const bool leader = (uint32_t)lane < rpw && lane_k_in_ray == 0 && slot < n_alive_k;
        int cnt = 0, index = 0;
        float ws = 0.0f;
        bool stopped = false;   // the last row that counts saw T < T_thresh
        auto leader_phase = [&]() {
            if (lane < 32 && lane_k_in_ray == 0) {
                if (leader && ((live >> lane) & 1u)) {
                    if (early_ray) { index = early_index; ws = early_ws; }
                    else { index = rays_alive[slot]; ws = weights_sum[index]; }
                    for (uint32_t k = 0; k < fstep; k++) {
                        if (!((live >> (lane + k)) & 1u)) break;
                        const float T = 1.0f - ws;
                        const float wgt = ex[(lane + k) * 3] * T;
                        ws += wgt;
                        ex[(lane + k) * 3] = wgt;
                        cnt++;
                        if (T < T_thresh) { stopped = true; break; }
                    }
                } else if (leader && rs.rays_t) {
                    if (early_ray) { index = early_index; ws = early_ws; }
                    else { index = rays_alive[slot]; ws = weights_sum[index]; }
                }
                ex[lane * 3 + 1] = __int_as_float(index);
                ex[lane * 3 + 2] = __int_as_float(cnt);
            }
        };
        if constexpr (kWide) {
            if (fuse_composite) {
                if (valid && h == 0) ex[(lane & 31) * 3] = 1.0f - __expf(-(pp.density_scale * __expf(sigma_logit)) * dl.x);   // alpha, exactly as k_frame_composite forms it
                leader_phase();
            }
        }
        const int ray_cnt_w = (kWide && fstep > 1u) ? __shfl(cnt, (lane & 31) - (int)lane_k_in_ray) : cnt;   // the count of this lane's ray (held by its leader lane)
        float rgb_out[3] = {0.0f, 0.0f, 0.0f};   // this row's final colour, kept for the ray-state composite below
        float rowv[36];                          // (4-basis 12-wave kernels: the row's values; dead rows never count)
#pragma unroll
        for (int q = 32; q < 36; q++) rowv[q] = 0.0f;
        // the ray state of the tile's rays, requested here (wide kernels) and used behind the scalar epilogue.  Issued by every lane without a branch --
        // early_index is 0 outside the leader lanes, and without a ray state the loads read `dirs` (always there) -- because behind a branch the
        // compiler merges the loaded registers with the other path's zeros at once: a copy, and in front of it the wait this placement exists to avoid.
        float rs_t = 0.0f, rs_d = 0.0f, rs_r = 0.0f, rs_g = 0.0f, rs_b = 0.0f;
        if constexpr (kWide) {
            const bool rs_early = early_ray && rs.rays_t != nullptr;
            const float* __restrict__ p_t = rs_early ? rs.rays_t : dirs;
            const float* __restrict__ p_d = rs_early ? rs.depth : dirs;
            const float* __restrict__ p_i = rs_early ? rs.image : dirs;
            rs_t = p_t[early_index]; rs_d = p_d[early_index];
            rs_r = p_i[early_index * 3]; rs_g = p_i[early_index * 3 + 1]; rs_b = p_i[early_index * 3 + 2];
        }
        if (valid && h == 0) {
            const PaletteTables& T = *reinterpret_cast<const PaletteTables*>(w + toff);
            float omega[kMaxNb], osum = 0.0f;
#pragma unroll
            for (int b = 0; b < kLoopNb; b++) if (NB || b < nb) { omega[b] = softplusf(om[b]) + 0.05f; osum += omega[b]; }
            // offsets_radiance outputs by index (compile-time after unrolling); radiance is the LAST of the 3 nb + 1 outputs (palette/renderer.py:471)
            auto orv = [&](int idx) -> float { return idx < 16 ? orr[idx & 15] : orr2[idx & 15]; };
            float radiance = 0.0f;
            if constexpr (NB != 0) radiance = orv(3 * NB);
            else {
#pragma unroll
                for (int b = 1; b <= kMaxNb; b++) if (nb == b) radiance = orv(3 * b);
            }
            const float sp = softplusf(radiance);
            float rgb[3] = {0.0f, 0.0f, 0.0f};
            // aux row: straight to global (one 4-byte store per channel and lane, rows aux_stride apart), or -- when the LDS has room --
            // into this wave's staging slab, from where the whole 32-row tile (contiguous in memory) goes out as 16-byte stores
            // staging layout behind the weights: WAVES slabs of 32 rows x stage_stride floats, then WAVES x 32 x 3 spare floats (per row: the
            // compositing weight; per ray leader: ray id and number of rows that count)
            float* a = (stage_stride && !kWide) ? reinterpret_cast<float*>(w + packed_bytes) + ((size_t)wave * 32 + (lane & 31)) * stage_stride
                                                : aux + (size_t)n * pp.aux_stride;
            // `a` is LDS or global, so its stores would be flat_store (57 per tile through both the vector-memory and the LDS queue); the 12-wave
            // kernels always stage (the launcher sees to it): there the row is written with ds_write
            // ... and, with the shipped 4 bases, the first 32 floats of the row (compile-time positions) leave as eight 16-byte writes
            auto put_tail = [&](int idx, float v) { a[idx] = v; };   // run-time positions (clip columns, padding); never reached by the wide kernels
            auto put = [&](int idx, float v) {
                if constexpr (kWide) rowv[idx] = v;      // (compile-time positions, all below 34: the row stays in registers)
                else put_tail(idx, v);
            };
            if constexpr (EDIT == 3) {
                // "network heads" (pnr_palette_edit.mode 3): the row is what PaletteNetwork.forward returns per sample (palette/network.py:156-190) --
                // [omega nb (normalised) | offsets_radiance 3 nb + 1 (raw, bias added) | view_dep 3 | diffuse 3 | clip_feat clip_dim | 0-pad] -- and the
                // colour-basis composite is left to the caller (the reference's renderer does it with torch ops).  sigmas = density_scale * exp(logit).
#pragma unroll
                for (int b = 0; b < kLoopNb; b++) if (NB || b < nb) put(b, omega[b] / osum);
#pragma unroll
                for (int j = 0; j < 3 * kLoopNb + 1; j++) if (NB || j < 3 * nb + 1) put(nb + j, orv(j));
#pragma unroll
                for (int k = 0; k < 3; k++) { put(4 * nb + 1 + k, view_dep[k]); put(4 * nb + 4 + k, diffuse[k]); }
                int c = 4 * nb + 7;
                if constexpr (kWide) {      // (4 bases: 23 values; the rest of the row is padding)
#pragma unroll
                    for (int q = 23; q < 36; q++) rowv[q] = 0.0f;
                    c = pp.aux_stride;
                }
                if (pred_clip) {
                    const int c0 = 4 * nb + 7;
#pragma unroll
                    for (int k = 0; k < PNR_MAX_CLIP; k++) if (k < pp.clip_dim) put_tail(c0 + k, k < 16 ? clip[k & 15] : clip2[k & 15]);
                    if (c < c0 + pp.clip_dim) c = c0 + pp.clip_dim;
                }
                for (; c < pp.aux_stride; c++) put_tail(c, 0.0f);
                sigmas[n] = pp.density_scale * __expf(sigma_logit);
#pragma unroll
                for (int k = 0; k < 3; k++) rgbs[(size_t)n * 3 + k] = 0.0f;
            } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { put(k, diffuse[k] + view_dep[k]); put(3 + k, view_dep[k]); }   // direct_rgb, view_dep_rgb
#pragma unroll
            for (int b = 0; b < kLoopNb; b++) if (NB || b < nb) { omega[b] = omega[b] / osum; put(6 + b, omega[b]); }
            float edit_w = 1.0f;
            if constexpr (EDIT == 1) {   // RegionEdit's window (palette/renderer.py:126-134)
                if (ep->has_mean_xyz) {
                    float d2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const float d = xyzs[(size_t)n * 3 + k] - ep->mean_xyz[k]; d2 += d * d; }
                    edit_w *= expf(-d2 / ep->std_xyz);
                }
                if (ep->has_mean_clip) {
                    float d2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < PNR_MAX_CLIP; k++) if (k < ep->has_mean_clip) {   // without a clip head clip_feat is zeros (palette/network.py:179)
                        const float c = (pred_clip && k < pp.clip_dim) ? (k < 16 ? clip[k & 15] : clip2[k & 15]) : 0.0f;
                        const float d = c - ep->mean_clip[k];
                        d2 += d * d;
                    }
                    edit_w *= expf(-d2 / ep->std_clip);
                }
            }
#pragma unroll
            for (int b = 0; b < kLoopNb; b++) if (NB || b < nb) {
                float fin[3], off[3];
#pragma unroll
                for (int k = 0; k < 3; k++) off[k] = orv(3 * b + k);
                if constexpr (EDIT == 2) {   // Stylizer.forward (palette/renderer.py:166-183)
                    const float inten = fmaxf(sp + ep->dI[b], 0.0f);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float o2 = 0.0f;
#pragma unroll
                        for (int i = 0; i < 3; i++) o2 = fmaf(off[i], ep->ddelta[b][i][k], o2);
                        fin[k] = fminf(fmaxf(inten * ((T.basis_color[b][k] + ep->dP[b][k]) + o2), 0.0f), 1.0f);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) fin[k] = sp * (T.basis_color[b][k] + pp.offsets_weight * off[k]);
                }
                if constexpr (EDIT == 1) {   // RegionEdit.forward (palette/renderer.py:121-147)
                    if (ep->weight_mode) { fin[0] = fin[1] = fin[2] = edit_w; }
                    else {
                        float hh, ss, vv, er, eg, eb;
                        rgb_to_hsv_px(fin[0], fin[1], fin[2], hh, ss, vv);
                        hh = fmodf((hh + ep->delta_hsv[b][0]) + 360.0f, 360.0f);
                        ss = fmaxf(ss * ep->delta_hsv[b][1], 0.0f);
                        vv = fmaxf(vv * ep->delta_hsv[b][2], 0.0f);
                        hsv_to_rgb_px(hh, ss, vv, er, eg, eb);
                        fin[0] = torch_lerp(fin[0], er, edit_w); fin[1] = torch_lerp(fin[1], eg, edit_w); fin[2] = torch_lerp(fin[2], eb, edit_w);
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float brgb = omega[b] * fin[k];
                    rgb[k] += brgb;
                    put(6 + nb + 3 * b + k, brgb);                                    // basis_rgb
                    put(6 + 4 * nb + 3 * b + k, T.basis_color[b][k] + off[k]);       // unscaled_basis_rgb
                }
            }
            int c = 6 + 7 * nb;
            if constexpr (!kWide) {     // (wide: the row stays in rowv)
            if (pred_clip) {   // the clip head's outputs sit in registers: compile-time indices
#pragma unroll
                for (int k = 0; k < PNR_MAX_CLIP; k++) if (k < pp.clip_dim) put_tail(c + k, k < 16 ? clip[k & 15] : clip2[k & 15]);
                c += pp.clip_dim;
            }
            for (; c < pp.aux_stride; c++) put_tail(c, 0.0f);   // (without a clip head the clip_dim columns are zeros too, as the reference's torch.zeros clip_feat)
            }
            const float sigma = pp.density_scale * __expf(sigma_logit);
            if (!rs.rays_t) sigmas[n] = sigma;   // (with the ray state composited here nobody reads sigmas / rgbs)
            if constexpr (!kWide) { if (fuse_composite) ex[(lane & 31) * 3] = 1.0f - __expf(-sigma * dl.x); }   // alpha, exactly as k_frame_composite forms it (dl.x = deltas[2 n]); wide: formed above
            const float kvd = EDIT == 2 ? 1.0f : pp.view_dep_weight;   // the Stylizer adds view_dep unscaled (palette/renderer.py:181)
#pragma unroll
            for (int k = 0; k < 3; k++) { rgb_out[k] = rgb[k] + kvd * view_dep[k]; if (!rs.rays_t) rgbs[(size_t)n * 3 + k] = rgb_out[k]; }
            }   // EDIT != 3
            if constexpr (kWide) {
                // The row goes where it belongs from inside this block (round 5): its 34 values then never cross a merge of the control flow -- carried
                // through the leader phase and the composite's branches they were what kept a 128-register build of this kernel spilling.
                // acc = fmaf(weight_k, value_k, acc) for k = 0, 1, ...: the chain the staged form of the 8-wave kernels runs, in the same order.
                if (fuse_composite) {
                        const bool lead_acc = lane < 32 && lane_k_in_ray == 0 && cnt > 0;      // (cnt > 0 only on leader lanes with a live first row)
                        f32x4* __restrict__ arow = reinterpret_cast<f32x4*>(aux_map + (size_t)index * 36);
                        if (fstep == 1u) {     // (wave-uniform) one sample per ray -- every heavy launch: the row's own lane does the ray's read-modify-write
                            if (lead_acc) {
                                f32x4 acc[9];
    #pragma unroll
                                for (int q = 0; q < 9; q++) acc[q] = arow[q];                          // nine requests in flight, one trip
                                const float w0 = ex[lane * 3];
    #pragma unroll
                                for (int c = 0; c < 34; c++) acc[c >> 2][c & 3] = fmaf(w0, rowv[c], acc[c >> 2][c & 3]);
    #pragma unroll
                                for (int q = 0; q < 9; q++) arow[q] = acc[q];
                            }
                        } else {
                            // 2 .. 8 samples per ray (at most 16 rays in the tile): the rays' rows go through a small LDS image (16 x 144 bytes per wave) -- the
                            // leader lane fetches its ray's row, round k lets the lane that holds row k add it (the wave's DS operations complete in order, so
                            // round k + 1 reads what round k wrote), the leader writes the row back.  Same fmaf chain, same order.
                            typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
                            float* img = reinterpret_cast<float*>(w + packed_bytes) + (size_t)WAVES * 96 + (size_t)wave * (16 * 36);
                            lds_f32x4* srow = reinterpret_cast<lds_f32x4*>(reinterpret_cast<uintptr_t>(img + (lane_q & 15u) * 36u));
                                                    if (lead_acc) {
                                f32x4 acc[9];
    #pragma unroll
                                for (int q = 0; q < 9; q++) acc[q] = arow[q];
    #pragma unroll
                                for (int q = 0; q < 9; q++) srow[q] = acc[q];
                            }
                            // The rounds hand the row from lane to lane through LDS: between two of them the compiler must neither keep a lane's own copy of `srow` nor move
                            // an access across (a wave barrier + a compiler fence: no instruction on the device, where a wave's DS operations complete in order anyway).
                            auto round_fence = [] { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };
                            round_fence();
                            const bool counts = lane < 32 && (int)lane_k_in_ray < ray_cnt_w;
                            const float my_w = counts ? ex[lane * 3] : 0.0f;
    #pragma unroll 1
                            for (uint32_t k = 0; k < fstep; k++) {     // wave-uniform
                                if (counts && lane_k_in_ray == k) {
    #pragma unroll
                                    for (int q = 0; q < 9; q++) {
                                        f32x4 a4 = srow[q];
                                        a4.x = fmaf(my_w, rowv[4 * q], a4.x); a4.y = fmaf(my_w, rowv[4 * q + 1], a4.y);
                                        if (q < 8) { a4.z = fmaf(my_w, rowv[4 * q + 2], a4.z); a4.w = fmaf(my_w, rowv[4 * q + 3], a4.w); }   // (columns 34, 35 are padding)
                                        srow[q] = a4;
                                    }
                                }
                                round_fence();
                            }
                            if (lead_acc) {
    #pragma unroll
                                for (int q = 0; q < 9; q++) arow[q] = srow[q];
                            }
                        }

                } else {     // rows straight from the registers to the caller's aux buffer (stand-alone op; frames whose composite is a launch of its own)
                    f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(aux + (size_t)n * 36);
#pragma unroll
                    for (int q = 0; q < 9; q++) dst[q] = f32x4{rowv[4 * q], rowv[4 * q + 1], rowv[4 * q + 2], rowv[4 * q + 3]};
                }
            }
        }
        if constexpr (CHECK) { if (overflow_flag && sw_.overflowed()) *overflow_flag = 1; }
        PAL_T(PT_EPILOGUE);
        if (stage_stride) {   // (non-wide: the same wave wrote the slab -- DS operations of a wave complete in order)
            float* slab = reinterpret_cast<float*>(w + packed_bytes) + (size_t)wave * 32 * stage_stride;                          // (not in the wide kernels)
            const uint32_t n0 = wt * rpw, nq = (uint32_t)pp.aux_stride / 4;
            if (fuse_composite) {
                // aux_map[ray] += sum_k weight_k * row_k: the recurrence of raymarching.cu:1114-1185 (weights from the weights_sum of
                // BEFORE this iteration, stop at a dead row, stop after the sample that sees T < T_thresh), same fmaf order.
                // Leader lane of a ray: weights of its rows, how many count, and the ray id, left in the wave's spare LDS columns.
                if constexpr (!kWide) leader_phase();      // (the wide kernels have run it in front of the scalar epilogue)
                PAL_T(PT_LEADER);
                float t = 0.0f, d = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
                float dl1_0 = dl.y;
                if constexpr (kWide) {
                    t = rs_t; d = rs_d; r = rs_r; g = rs_g; b = rs_b;      // (the rows have gone into aux_map inside the scalar epilogue's block)
                } else {
                // the ray state is requested here and used behind the aux rows' composite below (the loads land under it)
                if (rs.rays_t && leader) {
                    t = rs.rays_t[index]; d = rs.depth[index]; r = rs.image[index * 3]; g = rs.image[index * 3 + 1]; b = rs.image[index * 3 + 2];
                }
                const uint32_t rays_in_tile = rays_pt;
                for (uint32_t i = (uint32_t)lane; i < rays_in_tile * nq; i += 64) {
                    const uint32_t ray = i / nq, q = i - ray * nq, base = ray * fstep;
                    const int cnt = __float_as_int(ex[base * 3 + 2]);
                    if (cnt == 0) continue;
                    const int index = __float_as_int(ex[base * 3 + 1]);
                    float4* dst = reinterpret_cast<float4*>(aux_map + (size_t)index * pp.aux_stride) + q;
                    float4 acc = *dst;
                    for (int k = 0; k < cnt; k++) {
                        const float* r = slab + (base + k) * stage_stride;
                        const float wgt = ex[(base + k) * 3];
                        const float4 v = *reinterpret_cast<const float4*>(r + q * 4);
                        acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
                    }
                    *dst = acc;
                }
                }
                PAL_T(PT_AUXMAP);
                if (rs.rays_t) {
                    // k_frame_composite's second phase (raymarching.cu:1025-1111) for the rays of this wave tile: the weights are the ones just formed
                    // (same alpha, same T recurrence), the rows' colours come from the lanes that hold them
                    for (uint32_t k = 0; k < fstep; k++) {   // wave-uniform
                        const int src = lane + (int)k;
                        const float r_k = __shfl(rgb_out[0], src), g_k = __shfl(rgb_out[1], src), b_k = __shfl(rgb_out[2], src);
                        const float dl1_k = k == 0 ? dl1_0 : __shfl(dl.y, src);     // deltas[2 (n0 + lane + k) + 1], held by the row's own lane
                        if (leader && (int)k < cnt) {
                            const float wgt = ex[(lane + k) * 3];
                            t += dl1_k;
                            d = fmaf(wgt, t, d);
                            r = fmaf(wgt, r_k, r); g = fmaf(wgt, g_k, g); b = fmaf(wgt, b_k, b);
                        }
                    }
                    int keep = 0;
                    if (leader) {
                        if (cnt == (int)fstep && !stopped) { rs.rays_t[index] = t; keep = 1; } else rs.rays_alive[slot] = -1;
                        rs.weights_sum[index] = ws; rs.depth[index] = d;
                        rs.image[index * 3] = r; rs.image[index * 3 + 1] = g; rs.image[index * 3 + 2] = b;
                    }
                    const unsigned long long km = __ballot(keep);
                    if (km != 0ull) {   // consecutive slots: at most two chunks of the alive list
                        const uint32_t c0 = (wt * rays_pt) >> 8;
                        const unsigned long long k0 = __ballot(keep && (slot >> 8) == c0);
                        if (lane == 0) {
                            if (k0) atomicAdd(&rs.counts_cur[c0], __popcll(k0));
                            if (km & ~k0) atomicAdd(&rs.counts_cur[c0 + 1], __popcll(km & ~k0));
                        }
                    }
                }
            } else if constexpr (!kWide) {
                for (uint32_t i = (uint32_t)lane; i < 32 * nq; i += 64) {
                    const uint32_t row = i / nq, q = i - row * nq;
                    if ((live >> row) & 1u)
                        *reinterpret_cast<float4*>(aux + (size_t)(n0 + row) * pp.aux_stride + q * 4) = *reinterpret_cast<const float4*>(slab + row * stage_stride + q * 4);
                }
            }
        }
#endif
        PAL_T(PT_RAYSTATE);
    }
#ifdef PNR_PAL_TIMING
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < PT_N; i++) atomicAdd(&g_pal_timing[i], tacc[i]);
        atomicAdd(&g_pal_timing[PT_N], ntiles);
        atomicAdd(&g_pal_timing[PT_N + 1], 1ull);
        atomicAdd(&g_pal_timing[PT_N + 2], wall_clock64() - tstart);
    }
#endif
}

}  // namespace pnr

using namespace pnr;

namespace {
// device copy of the edit parameters of the STAND-ALONE op (the frame loops upload theirs into the frame workspace): a ring of slots per
// device, so that launches in flight keep their own.  Several host threads may call the op: the slot cursor and the lazy allocation sit behind
// a mutex, the host source of the copy is this call's own stack (pageable: staged by the runtime before hipMemcpyAsync returns), and a slot
// is handed out again only behind the event recorded after the launch that read it last (it may have been on another stream).
constexpr int kEditSlots = 64;
struct EditRing { EditParams* dev = nullptr; hipEvent_t used[kEditSlots] = {}; int next = 0; };
EditRing g_edit_ring[kMaxDevices];
std::mutex g_edit_ring_mutex;

bool shape_ok(uint32_t nb, uint32_t clip_dim) { return nb >= 1 && nb <= PNR_MAX_BASIS && clip_dim <= PNR_MAX_CLIP; }

void fill_edit(EditParams& e, const pnr_palette_edit& src) {
    memset(&e, 0, sizeof(e));
    memcpy(e.delta_hsv, src.delta_hsv, sizeof(e.delta_hsv));
    memcpy(e.mean_xyz, src.mean_xyz, sizeof(e.mean_xyz));
    memcpy(e.mean_clip, src.mean_clip, sizeof(e.mean_clip));
    e.std_xyz = src.std_xyz; e.std_clip = src.std_clip;
    e.has_mean_xyz = src.has_mean_xyz; e.has_mean_clip = src.has_mean_clip; e.weight_mode = src.weight_mode;
    memcpy(e.dI, src.dI, sizeof(e.dI)); memcpy(e.dP, src.dP, sizeof(e.dP)); memcpy(e.ddelta, src.ddelta, sizeof(e.ddelta));
}
}  // namespace

// internal (frame.hip): the device image of a pnr_palette_edit, uploaded once per frame into the frame workspace
uint64_t pnr_internal_edit_device_bytes() { return sizeof(EditParams); }
int pnr_internal_edit_upload(const pnr_palette_edit* edit, void* dst, hipStream_t s) {
    EditParams e;
    fill_edit(e, *edit);
    return hipMemcpyAsync(dst, &e, sizeof(e), hipMemcpyHostToDevice, s) == hipSuccess ? PNR_OK : PNR_ERR_LAUNCH;   // pageable source: staged before the call returns
}

extern int g_opt_palette_waves12;

#ifdef PNR_PAL_TIMING
// diagnostic builds only (not in include/pnr.h): out != NULL: copy the PT_N + 3 accumulators out; reset != 0: zero them
extern "C" int pnr_debug_pal_timing(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return PNR_ERR_LAUNCH;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_pal_timing), sizeof(unsigned long long) * (pnr::PT_N + 3)) != hipSuccess) return PNR_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[pnr::PT_N + 3] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_pal_timing), z, sizeof(z)) != hipSuccess) return PNR_ERR_LAUNCH;
    }
    return PNR_OK;
}
#endif

extern "C" {

uint64_t pnr_palette_field_packed_bytes(uint32_t num_basis, uint32_t clip_dim, int pred_clip) {
    return (uint64_t)pal_blocks((int)num_basis, (int)clip_dim, pred_clip ? 1 : 0) * kF16BlockBytes + kVecBytes + kTablesBytes;   // blocks | two vector heads | tables
}
uint32_t pnr_palette_aux_channels(uint32_t num_basis, uint32_t clip_dim) { return (6 + 7 * num_basis + clip_dim + 3) & ~3u; }

int pnr_palette_field_pack(const pnr_palette_weights* pw, void* packed, pnr_stream_t stream) {
    if (!pw || !packed) return PNR_ERR_INVALID;
    if (!shape_ok(pw->num_basis, pw->clip_dim)) return PNR_ERR_UNSUPPORTED;
    if (pw->precision != PNR_FIELD_FP32 && pw->precision != PNR_FIELD_F16X3 && pw->precision != PNR_FIELD_F16X2) return PNR_ERR_UNSUPPORTED;
    if (!pw->sigma0 || !pw->sigma1 || !pw->diff0 || !pw->diff1 || !pw->diff2 || !pw->color0 || !pw->color1 || !pw->color2 || !pw->basis0 || !pw->basis1 ||
        !pw->offsets_radiance || !pw->omega || !pw->basis_color || !pw->or_bias)
        return PNR_ERR_INVALID;
    if (pw->pred_clip && (!pw->clip0 || !pw->clip1)) return PNR_ERR_INVALID;
    PackTable t;
    for (int i = 0; i < kPalMaxBlocks; i++) t.b[i] = PackBlock{nullptr, 0, 0, 0, 0, 0, 0};
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
    add(pw->color0, 31, 64, 2, 2, COL_SH_GEO, ROW_ID);        // PB_C0
    add(pw->color1, 64, 64, 2, 4, COL_FRAG, ROW_ID);          // PB_C1
    add(pw->basis0, 35, 64, 2, 3, COL_ENC_DIFF, ROW_ID);      // PB_B0
    add(pw->basis1, 64, 15, 1, 4, COL_FRAG, ROW_ID);          // PB_B1
    add(pw->offsets_radiance, 15, 3 * nb + 1, 1, 1, COL_FRAG15, ROW_HALF0);  // PB_OR
    add(pw->omega, 15, nb, 1, 1, COL_FRAG15, ROW_HALF0);      // PB_OM
    add(nb > 5 ? pw->offsets_radiance : nullptr, 15, 3 * nb + 1, 1, 1, COL_FRAG15, ROW_HALF0_B);   // PB_OR2: outputs 16..31 (zeros when unused)
    if (pw->pred_clip) {
        add(pw->clip0, 32, 64, 2, 2, COL_LINEAR, ROW_ID);     // PB_CL0
        add(pw->clip1, 64, (int)pw->clip_dim, 1, 4, COL_FRAG, ROW_HALF0);  // PB_CL1
        if (pw->clip_dim > 16) add(pw->clip1, 64, (int)pw->clip_dim, 1, 4, COL_FRAG, ROW_HALF0_B);  // PB_CL1B
    }
    t.n = pal_blocks(nb, (int)pw->clip_dim, pw->pred_clip ? 1 : 0);
    if (q < t.n) return PNR_ERR_INVALID;
    const dim3 grid(cdiv((uint32_t)t.n * 512, 256));
    if (pw->precision != PNR_FIELD_FP32) hipLaunchKernelGGL(k_pack_blocks<1>, grid, dim3(256), 0, as_stream(stream), t, static_cast<unsigned char*>(packed));
    else hipLaunchKernelGGL(k_pack_blocks<0>, grid, dim3(256), 0, as_stream(stream), t, static_cast<unsigned char*>(packed));
    float* vec = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + (size_t)t.n * kF16BlockBytes);
    hipLaunchKernelGGL(k_pack_vec_head, dim3(1), dim3(192), 0, as_stream(stream), pw->diff2, vec);
    hipLaunchKernelGGL(k_pack_vec_head, dim3(1), dim3(192), 0, as_stream(stream), pw->color2, vec + kVecHeadBytes / 4);
    hipLaunchKernelGGL(k_pack_tables, dim3(1), dim3(64), 0, as_stream(stream), pw->basis_color, pw->or_bias, nb, vec + kVecBytes / 4);
    return check_launch();
}

int pnr_palette_field_stages_aux(uint32_t num_basis, uint32_t clip_dim, int pred_clip) {
    const uint32_t aux_stride = pnr_palette_aux_channels(num_basis, clip_dim);
    const uint32_t packed_bytes = (uint32_t)pnr_palette_field_packed_bytes(num_basis, clip_dim, pred_clip);
    return packed_bytes + (kPalThreads / 64) * 32 * (aux_stride + 3) * 4 <= 160 * 1024;
}

int pnr_palette_field_forward(const pnr_palette_field_args* a, pnr_stream_t stream) {
    if (!a) return PNR_ERR_INVALID;
    if (!shape_ok(a->num_basis, a->clip_dim)) return PNR_ERR_UNSUPPORTED;
    if (a->precision != PNR_FIELD_FP32 && a->precision != PNR_FIELD_F16X3 && a->precision != PNR_FIELD_F16X2) return PNR_ERR_UNSUPPORTED;
    if (a->aux_stride < 6 + 7 * a->num_basis + a->clip_dim || a->aux_stride > PNR_CHANNEL_MAXIMUM || (a->aux_stride & 3u)) return PNR_ERR_INVALID;
    if (a->B == 0 && !a->ctl) return PNR_OK;
    if (!a->enc || !a->enc_palette || !a->dirs || !a->packed || !a->sigmas || !a->rgbs || !a->aux) return PNR_ERR_INVALID;
    if (a->pred_clip && !a->enc_clip) return PNR_ERR_INVALID;
    const int edit_mode = a->edit ? a->edit->mode : 0;
    if (edit_mode < 0 || edit_mode > 3) return PNR_ERR_UNSUPPORTED;
    if (edit_mode == 3 && (a->ctl || a->overflow_flag)) return PNR_ERR_UNSUPPORTED;   // the network-heads row is a stand-alone op (no frame loop, no watch)
    if (edit_mode == 1 && a->edit->has_mean_xyz && !a->xyzs) return PNR_ERR_INVALID;
    PaletteParams pp;
    pp.density_scale = a->density_scale; pp.offsets_weight = a->offsets_weight; pp.view_dep_weight = a->view_dep_weight;
    pp.nb = (int)a->num_basis; pp.clip_dim = (int)a->clip_dim; pp.pred_clip = a->pred_clip ? 1 : 0; pp.aux_stride = (int)a->aux_stride;
    for (int k = 0; k < 3; k++) { pp.enc_scale[k] = a->enc_scale[k] > 0.0f ? a->enc_scale[k] : 1.0f; pp.enc_scale_inv[k] = 1.0f / pp.enc_scale[k]; }
    const uint32_t packed_bytes = (uint32_t)pnr_palette_field_packed_bytes(a->num_basis, a->clip_dim, pp.pred_clip);
    const uint32_t rows_ub = a->B;
    // one persistent workgroup per CU (100-126 KiB of LDS): 8 waves, or 12 for the specialised 4-basis kernel when its staging fits (three waves per SIMD
    // at <= 168 registers: the dependent layer chain of a wave leaves the SIMD idle too often with two)
    // PNR_FIELD_F16X2 (opt-in: activations rounded once to fp16) exists for the specialised 4-basis kernel without an edit head; everything else runs it as F16X3
    const bool fp16 = a->precision != PNR_FIELD_FP32;
    const bool nb4 = fp16 && a->num_basis == 4 && !a->overflow_flag;
    const bool x2 = a->precision == PNR_FIELD_F16X2 && nb4 && edit_mode == 0;
    // "wide": the specialised kernels for the shipped shape -- 4 bases, no clip head, rows of 36 floats -- whose tail stages nothing in LDS (see the top of
    // the file): PNR_PAL_WIDE_WAVES waves per workgroup (16: four per SIMD at <= 128 registers) next to 100 KiB of weights + 384 bytes per wave
    const bool wide = nb4 && g_opt_palette_waves12 && !pp.pred_clip && a->aux_stride == 36u && a->aux_stride == pnr_palette_aux_channels(a->num_basis, a->clip_dim);
    const uint32_t waves = wide ? (uint32_t)PNR_PAL_WIDE_WAVES : 8u;
    const uint32_t ntiles = cdiv(rows_ub ? rows_ub : 1, 28);   // wave tiles (a wave tile holds 28 ... 32 rows when it holds whole rays): dealt wave-major, see the kernel
    const uint32_t grid = ntiles < 256u ? ntiles : 256u;
    constexpr uint32_t kLdsLimit = 160 * 1024;
    // staging slab for coalesced aux rows (8-wave kernels): 8 waves x 32 rows x (aux_stride + 3) floats, when it fits next to the weights.  The wide kernels
    // keep their rows in registers: `stage_stride` only tells them that the fused tail is available, their LDS holds 96 + 576 floats per wave behind the weights (per-row weights; the rows of up to 16 rays with 2 .. 8 samples each)
    const bool stages = wide || (a->aux_stride == pnr_palette_aux_channels(a->num_basis, a->clip_dim) && pnr_palette_field_stages_aux(a->num_basis, a->clip_dim, pp.pred_clip));
    const uint32_t stage_stride = stages ? a->aux_stride : 0;
    const uint32_t lds = packed_bytes + (wide ? waves * (96 + 16 * 36) * 4 : (stages ? waves * 32 * (stage_stride + 3) * 4 : 0));
    const bool fuse = a->ctl && a->rays_alive && a->weights_sum && a->aux_map;
    RayState rs = {};
    if (fuse && stages && a->rays_t && a->depth && a->image && a->rays_alive_rw && a->counts_cur) {
        rs.rays_t = a->rays_t; rs.weights_sum = a->weights_sum_rw; rs.depth = a->depth; rs.image = a->image; rs.rays_alive = a->rays_alive_rw; rs.counts_cur = a->counts_cur;
        if (!rs.weights_sum) return PNR_ERR_INVALID;
    }
    hipStream_t s = as_stream(stream);
    const EditParams* ep_dev = nullptr;
    int edit_slot = -1;
    if (edit_mode && a->edit_device) ep_dev = static_cast<const EditParams*>(a->edit_device);   // frame loop: uploaded once per frame
    else if (edit_mode == 1 || edit_mode == 2) {   // this call's parameters go into the next slot of the device's ring (async copy on the launch stream)
        EditRing& ring = g_edit_ring[current_device()];
        {
            std::lock_guard<std::mutex> lock(g_edit_ring_mutex);
            if (!ring.dev && hipMalloc(reinterpret_cast<void**>(&ring.dev), sizeof(EditParams) * kEditSlots) != hipSuccess) return PNR_ERR_LAUNCH;
            edit_slot = ring.next;
            ring.next = (ring.next + 1) % kEditSlots;
            if (!ring.used[edit_slot]) {
                if (hipEventCreateWithFlags(&ring.used[edit_slot], hipEventDisableTiming) != hipSuccess) return PNR_ERR_LAUNCH;
            } else if (hipStreamWaitEvent(s, ring.used[edit_slot], 0) != hipSuccess) return PNR_ERR_LAUNCH;   // the launch that read this slot 64 calls ago
        }
        EditParams e;
        fill_edit(e, *a->edit);
        if (hipMemcpyAsync(ring.dev + edit_slot, &e, sizeof(EditParams), hipMemcpyHostToDevice, s) != hipSuccess) return PNR_ERR_LAUNCH;
        ep_dev = ring.dev + edit_slot;
    }
    PalArgs ka;
    ka.ctl = static_cast<const FrameCtlView*>(a->ctl); ka.B = a->B; ka.enc = a->enc; ka.enc_pal = a->enc_palette; ka.enc_clip = a->enc_clip;
    ka.level_stride = a->level_stride; ka.dirs = a->dirs; ka.deltas = a->deltas; ka.packed = static_cast<const unsigned char*>(a->packed);
    ka.packed_bytes = packed_bytes; ka.pp = pp; ka.sigmas = a->sigmas; ka.rgbs = a->rgbs; ka.aux = a->aux; ka.stage_stride = stage_stride;
    ka.rays_alive = fuse ? a->rays_alive : nullptr; ka.weights_sum = fuse ? a->weights_sum : nullptr; ka.aux_map = fuse ? a->aux_map : nullptr;
    ka.T_thresh = a->T_thresh; ka.xyzs = a->xyzs; ka.ep = ep_dev; ka.overflow_flag = a->overflow_flag; ka.tile_counter = static_cast<uint32_t*>(a->tile_counter);
    ka.rs = rs;
    static bool attr_set[4][4][kMaxDevices] = {};   // [PREC + 2 * CHECK][EDIT]
#define PNR_LAUNCH_PAL(PREC, EDIT, CHECK) PNR_LAUNCH_PAL_NB(PREC, EDIT, CHECK, 0, 8, attr_set[PREC + (CHECK ? 2 : 0)][EDIT])
#define PNR_LAUNCH_PAL_NB(PREC, EDIT, CHECK, NB, WAVES, FLAGS)                                                                                 \
    do {                                                                                                                                       \
        if (!ensure_dynamic_lds(k_palette_field_fwd<PREC, EDIT, CHECK, NB, WAVES>, kLdsLimit, FLAGS)) return PNR_ERR_LAUNCH;                   \
        hipLaunchKernelGGL((k_palette_field_fwd<PREC, EDIT, CHECK, NB, WAVES>), dim3(grid), dim3(WAVES * 64), lds, s, ka);                     \
    } while (0)
    if (fp16 && a->overflow_flag) {   // the instantiation that watches its split operands
        if (edit_mode == 0) PNR_LAUNCH_PAL(1, 0, true); else if (edit_mode == 1) PNR_LAUNCH_PAL(1, 1, true); else PNR_LAUNCH_PAL(1, 2, true);
    } else if (fp16) {
        static bool attr_nb4[8][kMaxDevices] = {};
        // the shipped default of 4 bases (main_palette.py:76): specialised epilogue, 12-wave workgroups when the staging fits
        if (x2 && wide) PNR_LAUNCH_PAL_NB(2, 0, false, 4, PNR_PAL_WIDE_WAVES, attr_nb4[6]);
        else if (x2) PNR_LAUNCH_PAL_NB(2, 0, false, 4, 8, attr_nb4[7]);
        else if (edit_mode == 3 && nb4 && wide) { static bool attr_heads[kMaxDevices] = {}; PNR_LAUNCH_PAL_NB(1, 3, false, 4, PNR_PAL_WIDE_WAVES, attr_heads); }
        else if (edit_mode == 3) PNR_LAUNCH_PAL(1, 3, false);
        else if (nb4 && wide && edit_mode == 0) PNR_LAUNCH_PAL_NB(1, 0, false, 4, PNR_PAL_WIDE_WAVES, attr_nb4[0]);
        else if (nb4 && wide && edit_mode == 1) PNR_LAUNCH_PAL_NB(1, 1, false, 4, PNR_PAL_WIDE_WAVES, attr_nb4[1]);
        else if (nb4 && wide) PNR_LAUNCH_PAL_NB(1, 2, false, 4, PNR_PAL_WIDE_WAVES, attr_nb4[2]);
        else if (nb4 && edit_mode == 0) PNR_LAUNCH_PAL_NB(1, 0, false, 4, 8, attr_nb4[3]);
        else if (nb4 && edit_mode == 1) PNR_LAUNCH_PAL_NB(1, 1, false, 4, 8, attr_nb4[4]);
        else if (nb4) PNR_LAUNCH_PAL_NB(1, 2, false, 4, 8, attr_nb4[5]);
        else if (edit_mode == 0) PNR_LAUNCH_PAL(1, 0, false); else if (edit_mode == 1) PNR_LAUNCH_PAL(1, 1, false); else PNR_LAUNCH_PAL(1, 2, false);
    } else {
        if (edit_mode == 0) PNR_LAUNCH_PAL(0, 0, false); else if (edit_mode == 1) PNR_LAUNCH_PAL(0, 1, false); else if (edit_mode == 2) PNR_LAUNCH_PAL(0, 2, false);
        else PNR_LAUNCH_PAL(0, 3, false);
    }
#undef PNR_LAUNCH_PAL
#undef PNR_LAUNCH_PAL_NB
    if (edit_slot >= 0 && hipEventRecord(g_edit_ring[current_device()].used[edit_slot], s) != hipSuccess) return PNR_ERR_LAUNCH;
    return check_launch();
}

}  // extern "C"
