// march_core.hpp -- the occupancy-grid march state machine shared by raymarch.hip (drop-in entry points)
// and frame.hip (device-driven frame loop).
#pragma once
#include "pnr_common.hpp"
#include "lattice.hpp"
#include <float.h>
#include <string.h>
#include <stdlib.h>

namespace pnr {

// near / far of one ray against an axis-aligned box (reference raymarching.cu:95-148): pnr_near_far_from_aabb's kernel and the frame loops' first
// kernel (pnr_nerf_frame_args::aabb) both call this, so the two ways of getting a frame's nears / fars agree bit for bit
__device__ __forceinline__ void near_far_of(float ox, float oy, float oz, float dx, float dy, float dz, const float* __restrict__ aabb, float min_near,
                                            float& near_out, float& far_out) {
    const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
    if (near > far) { tmp = near; near = far; far = tmp; }
    float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
    if (ny > fy) { tmp = ny; ny = fy; fy = tmp; }
    if (near > fy || ny > far) { near_out = far_out = FLT_MAX; return; }
    if (ny > near) near = ny;
    if (fy < far) far = fy;
    float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
    if (nz > fz) { tmp = nz; nz = fz; fz = tmp; }
    if (near > fz || nz > far) { near_out = far_out = FLT_MAX; return; }
    if (nz > near) near = nz;
    if (fz < far) far = fz;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

// ------------------------------------------------------------------------------------------
// per-ray constants + the march state machine (reference raymarching.cu:336-349, 362-403)
//
// Two latency levers that leave every result bit unchanged:
//  * occupancy mip in LDS.  The bitfield is in Morton order, so a 4x4x4 brick of cells is 64
//    CONSECUTIVE bits = one aligned 8-byte word.  pnr_build_occupancy_mip() reduces every brick to
//    two bits (any cell set / all cells set; 2 x 8 KiB for C=2, H=128).  A march workgroup keeps both
//    masks in LDS: a probe that lands in an all-empty or all-full brick is answered from LDS
//    (~64 cycles) instead of a dependent global byte load (~500+ cycles under load) -- that is nearly
//    every probe of the long empty-space walks that set the duration of a march launch.  The per-cell
//    control flow (which t is probed next) is untouched.
//  * occupied bounding box.  The mip builder also reduces the occupied bricks of every cascade to one
//    world-space box B (expanded by two cells of the respective cascade).  No cell outside B is
//    occupied, B is convex, so once a ray has left B no later probe can emit a sample: the march
//    stops at min(far, exit(B)) instead of walking cell by cell to the scene AABB.  Those walks (rays
//    that just left the object, rays that miss it) are what set the duration of a march launch;
//    cutting them changes no output (a ray that would have found nothing still finds nothing).
//  * when H and bound are powers of two (every shipped config) the reference's double-precision
//    cell coordinate, its frexpf()/scalbnf() and the 1/mip_bound division are exact scalings by
//    powers of two; the POW2 variants do them with one fp32 multiply / exponent-field arithmetic.
// ------------------------------------------------------------------------------------------
struct RayCtx {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, fC, fH, half_H;
    uint32_t H, H3;
    int maxlevel;
    const uint8_t* __restrict__ grid;
    const uint32_t* mip_any;  // LDS (or nullptr)
    const uint32_t* mip_all;
    const float* box;         // LDS: occupied box (min xyz, max xyz), or nullptr
    bool block_skip;          // empty 4^3/8^3/16^3 blocks may be jumped (needs H % 64 == 0 so that blocks nest in the cascades)
    // the 64 cells of the last MIXED brick this ray looked at (one aligned 8-byte word of the Morton-ordered bitfield): a ray that walks a
    // partly filled brick cell by cell pays one dependent global load per brick instead of one per cell
    mutable uint32_t cached_brick;
    mutable unsigned long long cached_bits;
    // the empty block whose jump was last REFUSED because a lattice point sits in the exit plane's window (and in the inner planes'): the plane, the lattice and
    // therefore the verdict are the same from every later cell of that block, so the attempt (~1.5 plain probes) is not repeated there -- the slowest rays of a
    // frame's first launch took 33 refused jumps in a row (profiles/march_stats.py, kind 7).  Skipping an attempt is always exact: a jump is an optional shortcut.
    mutable uint32_t nojump_key;
#ifdef PNR_MARCH_TIMING
    mutable uint32_t n_loads;
#endif
};

struct MarchParams {  // ray-independent constants, computed once on the host
    float bound, dt_gamma, dt_min, dt_max;
    uint32_t C, H, max_steps;
    uint32_t mip_words;  // uint32 words per mask; 0 = no mip
    uint32_t block_skip; // empty-block jumps allowed (PNR_NO_BLOCK_SKIP=1 in the environment turns them off for A/B measurements)
    uint32_t coop;       // frame loop: the last rays of a wave are marched by the whole wave (march_coop_tail); same outputs
};

static MarchParams make_march_params(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, bool with_mip) {
    MarchParams p;
    const float two_sqrt3 = 2.0f * 1.7320508075688772f;          // raymarching.cu:22
    p.bound = bound; p.dt_gamma = dt_gamma;
    p.dt_min = two_sqrt3 / (float)max_steps;                        // :348
    p.dt_max = two_sqrt3 * (float)(1 << (C - 1)) / (float)H;        // :349
    p.C = C; p.H = H; p.max_steps = max_steps;
    p.mip_words = with_mip ? (uint32_t)(((uint64_t)C * H * H * H / 64 + 31) / 32) : 0;
    p.block_skip = g_opt_block_skip ? 1u : 0u;
    p.coop = g_opt_coop_march ? 1u : 0u;
    return p;
}
static inline bool is_pow2f(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    return v > 0.0f && (u & 0x7fffffu) == 0 && ((u >> 23) & 0xff) != 0 && ((u >> 23) & 0xff) != 0xff;
}

__device__ __forceinline__ void ctx_init(RayCtx& c, const float* __restrict__ o, const float* __restrict__ d, const MarchParams& p,
                                         const uint8_t* __restrict__ grid, const uint32_t* mip_lds) {
    c.ox = o[0]; c.oy = o[1]; c.oz = o[2];
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    c.rdx = 1.0f / c.dx; c.rdy = 1.0f / c.dy; c.rdz = 1.0f / c.dz;
    c.bound = p.bound; c.dt_gamma = p.dt_gamma; c.dt_min = p.dt_min; c.dt_max = p.dt_max;
    c.rH = 1.0f / (float)p.H; c.fC = (float)p.C; c.fH = (float)p.H; c.half_H = 0.5f * (float)p.H;
    c.H = p.H; c.H3 = p.H * p.H * p.H; c.maxlevel = (int)p.C - 1; c.grid = grid;
    c.mip_any = mip_lds;
    c.mip_all = mip_lds ? mip_lds + p.mip_words : nullptr;
    c.box = mip_lds ? reinterpret_cast<const float*>(mip_lds + 2 * p.mip_words) : nullptr;
    c.block_skip = mip_lds != nullptr && (p.H % 64u) == 0 && p.block_skip != 0;
    c.cached_brick = 0xffffffffu; c.cached_bits = 0ull;
    c.nojump_key = 0xffffffffu;
#ifdef PNR_MARCH_TIMING
    c.n_loads = 0;
#endif
}

// Slab test of the ray against the occupied box.  `far`: parameter beyond which the ray is outside the box for good
// (never larger than the far passed in).  `t_in` and the entry plane (coordinate `face` on axis `axis`, the LAST slab
// the ray enters) feed skip_to_box().  Any NaN in the slab arithmetic (0 * inf) disables both for that ray.
struct BoxHit { float far, t_in, face, o, rd; bool entry_valid; };

__device__ __forceinline__ BoxHit clip_to_box(const RayCtx& c, float far) {
    BoxHit h; h.far = far; h.t_in = -FLT_MAX; h.face = 0.0f; h.o = 0.0f; h.rd = 0.0f; h.entry_valid = false;
    if (!c.box) return h;
    const float ax = (c.box[0] - c.ox) * c.rdx, bx = (c.box[3] - c.ox) * c.rdx;
    const float ay = (c.box[1] - c.oy) * c.rdy, by = (c.box[4] - c.oy) * c.rdy;
    const float az = (c.box[2] - c.oz) * c.rdz, bz = (c.box[5] - c.oz) * c.rdz;
    if (ax != ax || bx != bx || ay != ay || by != by || az != az || bz != bz) return h;
    const float ex = fminf(ax, bx), ey = fminf(ay, by), ez = fminf(az, bz);
    const float t_in = fmaxf(fmaxf(ex, ey), ez);
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (t_in > t_out) { h.far = -FLT_MAX; return h; }   // the ray never touches the occupied box
    h.far = fminf(far, fmaf(fabsf(t_out), 1e-5f, t_out));  // B already carries a two-cell margin; this only absorbs slab-test rounding
    h.t_in = t_in; h.entry_valid = true;
    if (ex >= ey && ex >= ez) { h.face = ax <= bx ? c.box[0] : c.box[3]; h.o = c.ox; h.rd = c.rdx; }
    else if (ey >= ez)        { h.face = ay <= by ? c.box[1] : c.box[4]; h.o = c.oy; h.rd = c.rdy; }
    else                      { h.face = az <= bz ? c.box[2] : c.box[5]; h.o = c.oz; h.rd = c.rdz; }
    return h;
}
__device__ __forceinline__ float clip_far_to_box(const RayCtx& c, float far) { return clip_to_box(c, far).far; }

// Exact jump over the empty space in front of the occupied box (POW2 configurations with an aligned box only).
//
// Let P be the entry plane of the last slab the ray enters (axis a, coordinate `face`), b its crossing parameter.  B's
// faces lie on the cell grid of the coarsest cascade, which every finer cascade's grid refines, so every cell on the
// outer side of P is entirely outside B and therefore empty, and the ray is on the outer side of P for all t < b.
// Claim: if no lattice point of the ray lies in [b - 2 eps, b + 2 eps], the first lattice point q after b is one the
// reference probes, and the reference emits nothing before it.  (Every lattice point s < b - 2 eps has a computed
// position at least 2 eps |d_a| outside P, more than the position and cell-index rounding, so whatever the reference
// probes there is an empty cell.  Let p be the last point it probes before b: its cell ends at or before P, so the
// reference's tt(p) = p + min(tx,ty,tz) <= p + t_a <= b + eps, and its do/while stops at the first lattice point
// >= tt(p); that point is > p, hence > b - 2 eps, hence >= q, and <= q because tt(p) < q.)
// eps bounds the rounding of t_a and of our own b: 2^-24 * bound * |rd_a| from the sample position, a few ulps of t
// from the rest; the constants below carry a 4x margin.  When the window is not free the next coarse-cell plane
// further out is tried (same argument); if none qualifies the ray walks cell by cell as before.
// The lattice point is obtained with lattice_advance() when the step is constant over the jump, else by walking the
// lattice itself (no probes), which is the reference loop verbatim.
template <bool POW2>
__device__ __forceinline__ float skip_to_box(const RayCtx& c, const BoxHit& h, float t) {
    if constexpr (!POW2) return t;
    if (!h.entry_valid || !c.box || __float_as_uint(c.box[6]) != 1u) return t;
    if (!(h.t_in > t)) return t;
    const float cell = c.box[7];
    const float ard = fabsf(h.rd);
    if (!(ard < 1e6f)) return t;
    const float out = h.rd > 0.0f ? -cell : cell;  // along the axis, away from the box: the ray enters through the low face iff d > 0
    const bool const_min = c.dt_gamma == 0.0f || (h.t_in + c.dt_max) * c.dt_gamma <= c.dt_min;
    const bool const_max = !const_min && t * c.dt_gamma >= c.dt_max;
    for (int k = 0; k < 8; k++) {
        const float plane = fmaf((float)k, out, h.face);                 // exact: multiples of a power of two
        const float b = (plane - h.o) * h.rd;
        const float eps = fmaf(c.bound * 2.3841858e-7f, ard, (fabsf(b) + 1.0f) * 9.5367432e-7f);  // 2^-22 bound |rd| + 2^-20 (|b| + 1)
        if (!(b - t > 2.0f * eps)) return t;
        float q, prev;
        if (const_min) lattice_advance(t, fminf(c.dt_min, c.dt_max), b, q, prev);   // clamp(x <= dt_min) = min(dt_max, dt_min): max_steps so small that dt_min > dt_max steps by dt_max
        else if (const_max) lattice_advance(t, c.dt_max, b, q, prev);
        else lattice_walk(t, c.dt_gamma, c.dt_min, c.dt_max, b, q, prev);
        if (q - b > 2.0f * eps && b - prev > 2.0f * eps) return q;
    }
    return t;
}

// frexpf exponent of a finite non-negative float, clamped to [0, maxlevel] (== reference mip_from_*)
__device__ __forceinline__ int level_of(float mx, int maxlevel) {
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 126;  // zero / denormals give e <= -126 -> clamped to 0, as frexpf does
    return min(maxlevel, max(0, e));
}

template <bool MIP>
__device__ __forceinline__ bool cell_occupied(const RayCtx& c, uint32_t index) {
    if constexpr (MIP) {
        const uint32_t brick = index >> 6, word = brick >> 5, bit = 1u << (brick & 31u);
        if (!(c.mip_any[word] & bit)) return false;
        if (c.mip_all[word] & bit) return true;
        if (brick != c.cached_brick) {   // (the mip builder reads the bitfield as 8-byte words too: the alignment is part of its contract)
            c.cached_bits = *reinterpret_cast<const unsigned long long*>(c.grid + (size_t)brick * 8);
            c.cached_brick = brick;
#ifdef PNR_MARCH_TIMING
            c.n_loads++;
#endif
        }
        return (c.cached_bits >> (index & 63u)) & 1ull;
    }
    return c.grid[index >> 3] & (1u << (index & 7u));
}

// Largest all-empty aligned block around cell `index`, as log2 of its edge in cells: 4 (16^3 cells = 64 consecutive
// bricks = one aligned 64-bit word of the 'any' mask, Morton order), 3 (8^3 = one byte), 2 (4^3 = one bit); 0 = the
// brick holds something.
__device__ __forceinline__ int empty_block_log2(const RayCtx& c, uint32_t index) {
    const uint32_t brick = index >> 6;
    const uint32_t lo = c.mip_any[(brick >> 6) * 2], hi = c.mip_any[(brick >> 6) * 2 + 1];
    if ((lo | hi) == 0) return 4;
    const uint32_t w = (brick & 32u) ? hi : lo;
    if (((w >> (brick & 24u)) & 0xffu) == 0) return 3;
    return ((w >> (brick & 31u)) & 1u) ? 0 : 2;
}

// Probe the cell containing the point at parameter t.  Occupied: returns true and the sample
// (x,y,z,dt), t untouched.  Empty: advances t past the cell (do..while of the reference) and
// returns false.
//
// MIP && POW2: when the cell lies in an all-empty aligned block R of 4^3 / 8^3 / 16^3 cells, t is advanced past R in
// one go -- exactly.  With b the parameter at which the ray leaves R (through the far face of axis a) the argument of
// skip_to_box() applies unchanged provided that, in addition, (i) every point of R is probed at this same cascade
// (blocks never straddle a cascade boundary because their size divides H/4, and the dt-derived cascade is checked at
// both ends of the jump), and (ii) the ray keeps a margin m (4x the position + cell-index rounding) from R's other
// faces at both ends of the jump, hence throughout: then every lattice point before b - 2 eps is computed into a cell
// of R (empty), the reference's last probe p in R has tt(p) <= b + eps (its cell's a-face is at or before R's), and
// with a lattice-free window around b it lands on the same next point q.  Any check failing = the reference's own
// one-cell step.
// JUMPS = false compiles the block jumps out: every empty probe is the reference's own one-cell step (same results, since a jump is an
// exact shortcut; fewer registers and a shorter chain for the hosted march tail, whose rays walk partly filled bricks where jumps rarely apply).
#ifndef PNR_MARCH_NOJUMP_MEMO
#define PNR_MARCH_NOJUMP_MEMO 1
#endif
template <bool MIP, bool POW2, bool JUMPS = true>
__device__ __forceinline__ bool march_probe(const RayCtx& c, float& t, float& x, float& y, float& z, float& dt, int* kind = nullptr) {
    const float t0 = t;
    x = clampf(fmaf(t0, c.dx, c.ox), -c.bound, c.bound);
    y = clampf(fmaf(t0, c.dy, c.oy), -c.bound, c.bound);
    z = clampf(fmaf(t0, c.dz, c.oz), -c.bound, c.bound);
    dt = clampf(t0 * c.dt_gamma, c.dt_min, c.dt_max);
    int level, nx, ny, nz;
    float mip_bound;
    const float hi = (float)(c.H - 1);
    if constexpr (POW2) {
        const int lp = level_of(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), c.maxlevel);
        const int ld = level_of(dt * c.half_H, c.maxlevel);
        level = lp > ld ? lp : ld;
        mip_bound = fminf(__uint_as_float((uint32_t)(127 + level) << 23), c.bound);
        const float mip_rbound = __uint_as_float((254u << 23) - (__float_as_uint(mip_bound) & 0x7f800000u));  // exact 1/2^k
        nx = (int)clampf(fmaf(x, mip_rbound, 1.0f) * c.half_H, 0.0f, hi);
        ny = (int)clampf(fmaf(y, mip_rbound, 1.0f) * c.half_H, 0.0f, hi);
        nz = (int)clampf(fmaf(z, mip_rbound, 1.0f) * c.half_H, 0.0f, hi);
    } else {
        const int lp = mip_from_pos(x, y, z, c.fC), ld = mip_from_dt(dt, c.fH, c.fC);
        level = lp > ld ? lp : ld;
        mip_bound = fminf(scalbnf(1.0f, level), c.bound);
        const float mip_rbound = 1.0f / mip_bound;
        // double intermediate exactly as the reference's `0.5 * (x * mip_rbound + 1) * H`
        nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
        ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
        nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
    }
    const uint32_t index = (uint32_t)level * c.H3 + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    if constexpr (MIP && POW2 && JUMPS) {
        int sh = c.block_skip ? empty_block_log2(c, index) : 0;
        if (kind) *kind = sh == 0 ? 1 : 5;   // instrumented builds only: 0 emit, 1 cell step in a non-empty brick, 2/3/4 block jump, 5 block checks failed
        [[maybe_unused]] const uint32_t jump_key = ((index >> (3 * sh)) << 3) | (uint32_t)sh;     // (level, aligned block of 2^sh cells a side in Morton order, sh)
        if (sh == 0) { if (cell_occupied<MIP>(c, index)) { if (kind) *kind = 0; return true; } }
        else if (PNR_MARCH_NOJUMP_MEMO && jump_key == c.nojump_key) { if (kind) *kind = 7; }
        else {
            const float s = (float)(1 << sh);
            const float lox = (float)((nx >> sh) << sh), loy = (float)((ny >> sh) << sh), loz = (float)((nz >> sh) << sh);
            // world coordinates of R's faces (exact: powers of two), far face per axis in the direction of travel
            const float wlx = fmaf(lox * c.rH, 2.0f, -1.0f) * mip_bound, whx = fmaf((lox + s) * c.rH, 2.0f, -1.0f) * mip_bound;
            const float wly = fmaf(loy * c.rH, 2.0f, -1.0f) * mip_bound, why = fmaf((loy + s) * c.rH, 2.0f, -1.0f) * mip_bound;
            const float wlz = fmaf(loz * c.rH, 2.0f, -1.0f) * mip_bound, whz = fmaf((loz + s) * c.rH, 2.0f, -1.0f) * mip_bound;
            const float bx = signf(c.dx) < 0.0f ? wlx : whx;   // same convention as the one-cell step below
            const float by = signf(c.dy) < 0.0f ? wly : why;   // same convention as the one-cell step below
            const float bz = signf(c.dz) < 0.0f ? wlz : whz;   // same convention as the one-cell step below
            const float ux = (bx - x) * c.rdx, uy = (by - y) * c.rdy, uz = (bz - z) * c.rdz;
            const float tmin = fminf(ux, fminf(uy, uz));
            const float m = c.bound * 4.7683716e-7f;  // 2^-21 bound
            bool ok = tmin == tmin && tmin > 0.0f && tmin < 1e30f;
            // margins at the start ...
            ok = ok && (x - wlx > m) && (whx - x > m) && (y - wly > m) && (why - y > m) && (z - wlz > m) && (whz - z > m);
            // ... and where the ray leaves R: the two axes that do not exit stay inside by m
            const float ex = fmaf(tmin, c.dx, x), ey = fmaf(tmin, c.dy, y), ez = fmaf(tmin, c.dz, z);
            float ard, face, xc, rdc;
            if (ux <= uy && ux <= uz) { ard = fabsf(c.rdx); face = bx; xc = x; rdc = c.rdx; ok = ok && (ey - wly > m) && (why - ey > m) && (ez - wlz > m) && (whz - ez > m); }
            else if (uy <= uz)        { ard = fabsf(c.rdy); face = by; xc = y; rdc = c.rdy; ok = ok && (ex - wlx > m) && (whx - ex > m) && (ez - wlz > m) && (whz - ez > m); }
            else                      { ard = fabsf(c.rdz); face = bz; xc = z; rdc = c.rdz; ok = ok && (ex - wlx > m) && (whx - ex > m) && (ey - wly > m) && (why - ey > m); }
            ok = ok && ard < 1e6f;
            // block jumps need a constant step over the jump (then the dt-derived cascade is constant too); rays whose step
            // grows with t (dt_gamma > 0 between the clamps) walk cell by cell as the reference does
            const float b0 = t0 + tmin;
            const bool const_min = c.dt_gamma == 0.0f || (b0 + c.dt_max) * c.dt_gamma <= c.dt_min;
            const bool const_max = !const_min && t0 * c.dt_gamma >= c.dt_max;
            ok = ok && (const_min || const_max);
            if (ok) {
                // Exit plane first; when a lattice point sits in its window, the cell planes just inside R serve equally (R minus
                // its last cell layers is still an aligned box of empty cells): the ray then needs one or two ordinary steps more.
                const float d = const_min ? fminf(c.dt_min, c.dt_max) : c.dt_max;   // the value clamp() takes below dt_min (raymarching.cu:37-39: min(hi, max(lo, x)))
                const float back = (rdc > 0.0f ? -2.0f : 2.0f) * mip_bound * c.rH;   // one cell, against the direction of travel
#pragma unroll 1
                for (int k = 0; k < 3; k++) {
                    const float b = t0 + (fmaf((float)k, back, face) - xc) * rdc;
                    const float eps = fmaf(c.bound * 2.3841858e-7f, ard, (fabsf(b) + 1.0f) * 9.5367432e-7f);  // as in skip_to_box()
                    if (kind) *kind = k == 0 ? 6 : 7;   // instrumented builds: 6 = the exit plane is too close / its window is taken, 7 = an inner plane as well
                    if (!(b - t0 > 2.0f * eps)) break;
                    float q, prev;
                    lattice_advance(t0, d, b, q, prev);
                    if (q - b > 2.0f * eps && b - prev > 2.0f * eps) { t = q; if (kind) *kind = 2; return false; }
                }
                if (PNR_MARCH_NOJUMP_MEMO) c.nojump_key = jump_key;     // the windows are taken (or the exit is a cell away): the same from the next cell of this block
            }
        }
    } else {
        if (cell_occupied<MIP>(c, index)) return true;
    }
    const float tx = fmaf(fmaf(fmaf(0.5f, signf(c.dx), (float)nx + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -x) * c.rdx;
    const float ty = fmaf(fmaf(fmaf(0.5f, signf(c.dy), (float)ny + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -y) * c.rdy;
    const float tz = fmaf(fmaf(fmaf(0.5f, signf(c.dz), (float)nz + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -z) * c.rdz;
    const float tt = t0 + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    float tc = t0;
    do { tc += clampf(tc * c.dt_gamma, c.dt_min, c.dt_max); } while (tc < tt);
    t = tc;
    return false;
}

// stage both mip masks into LDS (no-op when the launch has no mip)
__device__ __forceinline__ const uint32_t* stage_mip(const uint32_t* __restrict__ mip, uint32_t words_per_mask) {
    extern __shared__ uint32_t mip_lds[];
    if (words_per_mask == 0) return nullptr;
    const uint32_t n = 2 * words_per_mask + 8;  // 'any' mask, 'all' mask, occupied box (6 floats + 2 pad)
    for (uint32_t i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
        if (i + 3 < n) *reinterpret_cast<uint4*>(&mip_lds[i]) = *reinterpret_cast<const uint4*>(&mip[i]);
        else for (uint32_t j = i; j < n; j++) mip_lds[j] = mip[j];
    }
    __syncthreads();
    return mip_lds;
}

// ------------------------------------------------------------------------------------------
// Wave-cooperative march tail.  A march launch used to last as long as its slowest ray: ~95 % of the waves are done after 8.5 us, the
// launch waits for the few rays that walk 7-13 cells through partly filled bricks at ~1.3 us per probe -- one lane busy, 63 idle
// (profiles/march_timing.py).  Once at most kCoopRays rays of a wave are still marching, the whole wave works for them: the rays are
// parked in LDS and each gets a group of 64 / 32 / 16 lanes; lane k of a group runs the ordinary march_probe() -- the very same code, so
// every probe is bit for bit the reference's -- at the k-th LATTICE POINT after the ray's current t (lattice.hpp: the values a ray can
// visit form a fixed sequence; lattice_steps() gives the k-th one in closed form for a constant step, a k-step loop otherwise).  Which of
// those points the reference actually visits is then a chain: from a visited empty point it goes to the first lattice point at or beyond
// that probe's exit parameter (the probe returns it; its index among the group's points comes from a binary search), from a visited
// occupied point -- a sample -- to the next point.  One scalar walker per group follows the chain from point 0 (readlane), hands the
// sample rows to the lanes that hold them, and stops at n_step samples, at `far`, or at the end of the window, where the next batch
// continues.  Probes at points the reference never visits are discarded.  13 dependent probes become one or two batches.
// ------------------------------------------------------------------------------------------
#ifndef PNR_COOP_RAYS
#define PNR_COOP_RAYS 2       // measured 1 / 2 / 4: lego 4.07 / 4.07 / 4.13 ms -- with 4 rays a group is 16 lattice points (3.5 cells): one batch then buys what 3 probes buy
#endif
constexpr int kCoopRays = PNR_COOP_RAYS;   // 1, 2 or 4
template <int NR>
struct CoopSharedT {                // per wave
    float f[NR][12];                // ox oy oz dx dy dz t far last_t
    int32_t i[NR][4];               // first row of the ray's slots (n * n_step), samples so far, still marching, (unused)
    float tl[PNR_WAVE];             // the lattice point every lane probed in this batch
};
using CoopShared = CoopSharedT<kCoopRays>;
__device__ __forceinline__ void wave_lds_sync() {   // LDS operations of one wave execute in order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float readlane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// TS = true (the training march's counting pass): a sample is recorded as its ray parameter, t_store[row] = t (null: counted only), instead of
// the three output rows; a window whose points are all samples is taken in one go (the dense interior of an object: 16-64 samples per batch).
template <bool MIP, bool POW2, bool TS = false, int NR = kCoopRays>
__device__ __forceinline__ uint32_t march_coop_tail(CoopSharedT<NR>& sh, const MarchParams& p, const uint8_t* __restrict__ grid, const uint32_t* mip_lds,
                                                    uint32_t n_step, bool active, const RayCtx& c, float t, float far, float last_t, uint32_t n,
                                                    uint32_t step, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                                    float* __restrict__ t_store = nullptr) {
    const int lane = threadIdx.x & (PNR_WAVE - 1);
    const unsigned long long am = __ballot(active);
    const int nrays = __popcll(am);
    const int my_slot = __popcll(am & ((1ull << lane) - 1ull));
    if (active) {
        float* f = sh.f[my_slot];
        f[0] = c.ox; f[1] = c.oy; f[2] = c.oz; f[3] = c.dx; f[4] = c.dy; f[5] = c.dz; f[6] = t; f[7] = far; f[8] = last_t;
        int32_t* q = sh.i[my_slot];
        q[0] = (int32_t)(n * n_step); q[1] = (int32_t)step; q[2] = 1;
    }
    const bool const_step = p.dt_gamma == 0.0f;
    const float d_const = clampf(0.0f, p.dt_min, p.dt_max);
    for (;;) {
        wave_lds_sync();
        uint32_t mask = 0;
        for (int r = 0; r < nrays; r++) mask |= (sh.i[r][2] != 0 ? 1u : 0u) << r;
        mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)mask);
        if (mask == 0) break;
        const int ns = __popc(mask);
        const int glog = ns > 2 ? 2 : ((ns == 2 || !const_step) ? 1 : 0);   // 4 / 2 / 1 groups; a growing step is walked point by point: windows of at most 32
        const int wlog = 6 - glog, W = 1 << wlog;
        const int g = lane >> wlog, k = lane & (W - 1);
        int slot = -1;
        { uint32_t m = mask; for (int i = 0; i < g; i++) m &= m - 1u; if (m) slot = __ffs((int)m) - 1; }
        const bool has = slot >= 0;
        const float* f = sh.f[has ? slot : 0];
        RayCtx hc;
        {
            const float o[3] = {f[0], f[1], f[2]}, dd[3] = {f[3], f[4], f[5]};
            ctx_init(hc, o, dd, p, grid, mip_lds);
        }
        const float t0 = f[6], far_s = f[7];
        float tk;
        if (const_step) tk = lattice_steps(t0, d_const, (uint32_t)k);
        else {
            tk = t0;
            for (int j = 0; j < W - 1; j++) { const float nt = tk + clampf(tk * p.dt_gamma, p.dt_min, p.dt_max); tk = j < k ? nt : tk; }
        }
        float q = tk, x = 0.0f, y = 0.0f, z = 0.0f, dt = 0.0f;
        int hit = 0;
        if (has && tk < far_s) hit = march_probe<MIP, POW2>(hc, q, x, y, z, dt) ? 1 : 0;
        // index of q (a lattice point: it was reached by lattice steps from tk) among the group's points; the group's end = beyond the window
        sh.tl[lane] = tk;
        wave_lds_sync();
        int lo = k + 1, hi = W;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            if (lo < hi) { const int mid = (lo + hi) >> 1; if (sh.tl[(g << wlog) + mid] < q) lo = mid + 1; else hi = mid; }
        }
        const int nxt = (g << wlog) + lo;
        int emit_row = -1;
        float emit_last = 0.0f;
        [[maybe_unused]] const unsigned long long hits_in_range = TS ? __ballot(hit != 0 && tk < far_s) : 0ull;
        for (int gg = 0; gg < (1 << glog); gg++) {
            uint32_t m = mask;
            for (int i = 0; i < gg; i++) m &= m - 1u;
            if (!m) break;
            const int s = __ffs((int)m) - 1;
            int cur = gg << wlog;
            const int end = cur + W;
            const int row0 = __builtin_amdgcn_readfirstlane(sh.i[s][0]);
            const uint32_t done0 = (uint32_t)__builtin_amdgcn_readfirstlane(sh.i[s][1]);
            float lt = uniform_f(sh.f[s][8]);
            const float far_l = uniform_f(sh.f[s][7]);
            uint32_t ne = 0;
            float tnext = far_l;
            bool fin = false;
            bool walk = true;
            if constexpr (TS) {
                const unsigned long long gmask = (W == 64 ? ~0ull : ((1ull << W) - 1ull)) << cur;
                if ((hits_in_range & gmask) == gmask && done0 + (uint32_t)W <= n_step) {   // every point of the window is a sample
                    if (lane >= cur && lane < end) emit_row = row0 + (int)done0 + (lane - cur);
                    ne = (uint32_t)W;
                    lt = tnext = readlane_f(tk, end - 1) + readlane_f(dt, end - 1);
                    fin = done0 + ne >= n_step;
                    walk = false;
                }
            }
            if (walk) for (;;) {
                const float tc = readlane_f(tk, cur);
                if (!(tc < far_l)) { fin = true; break; }
                if (__builtin_amdgcn_readlane(hit, cur)) {
                    if (lane == cur) { emit_row = row0 + (int)(done0 + ne); emit_last = lt; }
                    const float tn = tc + readlane_f(dt, cur);       // `t += dt` (raymarching.cu:389)
                    lt = tn; tnext = tn; ne++;
                    if (done0 + ne >= n_step) { fin = true; break; }
                    if (++cur == end) break;
                } else {
                    tnext = readlane_f(q, cur);
                    const int j = __builtin_amdgcn_readlane(nxt, cur);
                    if (j >= end) break;
                    cur = j;
                }
            }
            if (lane == (gg << wlog)) {
                sh.i[s][1] = (int32_t)(done0 + ne);
                sh.f[s][8] = lt; sh.f[s][6] = tnext;
                sh.i[s][2] = (!fin && tnext < far_l) ? 1 : 0;
            }
        }
        if constexpr (TS) {
            if (emit_row >= 0 && t_store) t_store[(size_t)emit_row] = tk;
        } else if (emit_row >= 0) {
            float* px = xyzs + (size_t)emit_row * 3;
            float* pd = dirs + (size_t)emit_row * 3;
            float* pl = deltas + (size_t)emit_row * 2;
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = hc.dx; pd[1] = hc.dy; pd[2] = hc.dz;
            const float tn = tk + dt;
            pl[0] = dt; pl[1] = tn - emit_last;
        }
    }
    wave_lds_sync();
    return active ? (uint32_t)sh.i[my_slot][1] : step;
}

}  // namespace pnr
