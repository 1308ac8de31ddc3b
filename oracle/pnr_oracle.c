/*
 * pnr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
 *
 * A plain-C, one-loop-per-"thread" restatement (single-threaded in its canonical build; see the build variants below) of the reference's hot-path
 * algorithms (zfkuang/PaletteNeRF: raymarching/, gridencoder/, shencoder/, palette/src).
 * Every function cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - SH encoder: pinned against the reference's own pure-PyTorch SHEncoder_torch
 *     (testing/test_shencoder.py:8-89) through tests/golden/sh_torch_deg{1..5}.npz.
 *   - renderer control flow (n_step schedule, compaction, composite call order, bg mix, depth
 *     normalisation): pinned by importing nerf/renderer.py and palette/renderer.py in the build
 *     container with these oracle ops injected (tests/golden/gen_golden.py).
 *   - operator wrappers (level offsets, per_level_scale, the [L,B,C] permute, autocast casts, the autograd Functions): the reference's own
 *     gridencoder/grid.py, shencoder/sphere_harmonics.py and raymarching/raymarching.py imported over this library through
 *     oracle/native_facade.py (tests/golden/gen_golden.py); regenerating every fixture through them changed no bit.
 *   - kernel-level arithmetic of march / composite / Morton / packbits / near-far / SH / HSV: the reference ships no golden vectors for
 *     them, but raymarching.cu, shencoder.cu and palette.cu compile unmodified for gfx950 through torch's hipify + hipcc
 *     (oracle/ref_build.py: build_hip -> oracle/_ref/ref_*.so), and tests/test_gpu_reference_kernels.py runs them on the MI355X next to
 *     this repository's kernels: bit-identical for march / composite / Morton / packbits / near-far, <= 2e-6 for the rest.  This library
 *     agrees with the kernels of this repository bit for bit on the same functions, hence with the reference's.
 *   - hash-grid encoder kernels: gridencoder.cu does not compile for gfx950 (atomicAdd(__half2*, __half2) is missing in ROCm 7.2 HIP),
 *     so its arithmetic is pinned through the reference's GridEncoder wrapper over this restatement and by independent int64 / float64
 *     formulations in tests/ -- "parity unpinned by execution" for that one extension.
 *
 * Canonical scalar spec (shared *by description*, not by header, with the HIP kernels):
 *   - IEEE fp32, no implicit contraction (-ffp-contract=off); the places where the reference's
 *     nvcc build contracts a*b+c are written as explicit fmaf() here and in the kernels:
 *       sample position      fmaf(t, d, o)                (raymarching.cu:364-366)
 *       grid-cell coordinate fmaf(x, mip_rbound, 1)       (raymarching.cu:377-379)
 *       perturbation         fmaf(dt0, noise, t0)         (raymarching.cu:354)
 *       hash-grid position   fmaf(x, scale, 0.5)          (gridencoder.cu:134)
 *       interpolation / compositing accumulators  fmaf(w, v, acc)
 *   - per-level scale of the hash grid is computed once on the host with exp2f() and handed to
 *     the loops.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* Build variants of this one source (oracle/Makefile):
 *   liborc.so        the canonical oracle: single-threaded, explicit fmaf() where nvcc contracts (above).
 *   liborc_omp.so    -fopenmp -DORC_OMP: the same arithmetic with the independent per-ray / per-sample loops spread over the host's
 *                    cores (bench.py's all-cores CPU baseline leg; results are identical, every iteration writes its own outputs).
 *   liborc_nofma.so  -DORC_NO_FMA: every ORC_FMA(a,b,c) is a rounded product followed by a rounded sum, i.e. the reference's kernel
 *                    bodies compiled WITHOUT contraction (g++ -ffp-contract=off) -- the build SURVEY.md Appendix B measured the
 *                    reference's own kernel_march_rays_train with (63 001 827 / 15 750 694 samples on scene S0 at 800^2 / 400^2).
 *                    tests/test_oracle.py pins this variant to those two reference-measured counts. */
#ifdef ORC_NO_FMA
static inline float orc_mul_add(float a, float b, float c) { volatile float p = a * b; return p + c; }
#define ORC_FMA(a, b, c) orc_mul_add((a), (b), (c))
#else
#define ORC_FMA(a, b, c) fmaf((a), (b), (c))
#endif
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(dynamic, 64)")
ORC_API void orc_set_threads(int n) { omp_set_num_threads(n); }
ORC_API int orc_max_threads(void) { return omp_get_max_threads(); }
#else
#define ORC_PAR_FOR
ORC_API void orc_set_threads(int n) { (void)n; }
ORC_API int orc_max_threads(void) { return 1; }
#endif

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static inline float orc_clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } /* raymarching.cu:37-39 */
static inline float orc_signf(float x) { return copysignf(1.0f, x); }                           /* raymarching.cu:33-35 */

/* raymarching.cu:59-66 : spread the low 10 bits of v so that there are two zero bits between each */
static inline uint32_t orc_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:68-74 */
static inline uint32_t orc_morton(uint32_t x, uint32_t y, uint32_t z) {
    return orc_expand_bits(x) | (orc_expand_bits(y) << 1) | (orc_expand_bits(z) << 2);
}
/* raymarching.cu:76-84 */
static inline uint32_t orc_compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:45-50 */
static inline int orc_mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
/* raymarching.cu:52-57 ; dt*H*0.5 : the 0.5 is a double literal, the product by 0.5 is exact */
static inline int orc_mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}

/* IEEE binary16 <-> binary32 (round-to-nearest-even), for the fp16-table mode of the grid encoder */
static inline float orc_h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; sh++; } m &= 0x3ffu; u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
static inline uint16_t orc_f2h(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u; int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15; uint32_t m = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(s | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000u; uint32_t shift = (uint32_t)(14 - e); uint32_t hm = m >> shift; uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fffu; uint16_t h = (uint16_t)(s | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* raymarching: utils                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* raymarching.cu:95-148  kernel_near_far_from_aabb */
ORC_API void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                    uint32_t N, float min_near, float* nears, float* fars) {
    ORC_PAR_FOR
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
        if (near > far) { tmp = near; near = far; far = tmp; }
        float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
        if (ny > fy) { tmp = ny; ny = fy; fy = tmp; }
        if (near > fy || ny > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
        if (nz > fz) { tmp = nz; nz = fz; fz = tmp; }
        if (near > fz || nz > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (nz > near) near = nz;
        if (fz < far) far = fz;
        if (near < min_near) near = min_near;
        nears[n] = near; fars[n] = far;
    }
}

/* raymarching.cu:166-201  kernel_sph_from_ray (bg_radius > 0 only; fp tolerance, uses atan2f/sqrtf) */
ORC_API void orc_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float RPI = 0.3183098861837907f;
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = 2 * theta * RPI - 1;
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* raymarching.cu:217-229 */
ORC_API void orc_morton3d(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)orc_morton((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
/* raymarching.cu:240-257 ; note: the shift is an arithmetic shift of a signed int in the reference */
ORC_API void orc_morton3d_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[n * 3] = (int32_t)orc_compact_bits((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)orc_compact_bits((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)orc_compact_bits((uint32_t)(ind >> 2));
    }
}
/* raymarching.cu:271-292 ; N = number of output bytes, strict '>' */
ORC_API void orc_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* the march state machine shared by march_rays_train (both passes) and march_rays             */
/* raymarching.cu:362-403 / 430-482 / 956-1010                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, fC, fH;
    uint32_t C, H, H3;
    const uint8_t* grid;
} orc_ray_ctx;

static inline void orc_ctx_init(orc_ray_ctx* c, const float* o, const float* d, float bound, float dt_gamma,
                                uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1.0f / c->dx; c->rdy = 1.0f / c->dy; c->rdz = 1.0f / c->dz;
    c->bound = bound; c->dt_gamma = dt_gamma;
    const float two_sqrt3 = 2.0f * 1.7320508075688772f;                 /* raymarching.cu:22,348 */
    c->dt_min = two_sqrt3 / (float)max_steps;                            /* :348 */
    c->dt_max = two_sqrt3 * (float)(1 << (C - 1)) / (float)H;            /* :349 */
    c->rH = 1.0f / (float)H; c->fC = (float)C; c->fH = (float)H;
    c->C = C; c->H = H; c->H3 = H * H * H; c->grid = grid;
}

/* One iteration of the loop body.  Returns 1 and fills (x,y,z,dt) when the cell is occupied
 * (caller then does t += dt); returns 0 after having advanced *t past the empty cell. */
static inline int orc_march_probe(const orc_ray_ctx* c, float* t, float* px, float* py, float* pz, float* pdt) {
    const float tt0 = *t;
    const float x = orc_clampf(ORC_FMA(tt0, c->dx, c->ox), -c->bound, c->bound);
    const float y = orc_clampf(ORC_FMA(tt0, c->dy, c->oy), -c->bound, c->bound);
    const float z = orc_clampf(ORC_FMA(tt0, c->dz, c->oz), -c->bound, c->bound);
    const float dt = orc_clampf(tt0 * c->dt_gamma, c->dt_min, c->dt_max);
    const int lp = orc_mip_from_pos(x, y, z, c->fC), ld = orc_mip_from_dt(dt, c->fH, c->fC);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1.0f / mip_bound;
    const float hi = (float)(c->H - 1);
    /* :377-379  double intermediate, narrowed to float by clamp(float,...), truncated to int */
    const int nx = (int)orc_clampf((float)(0.5 * (double)ORC_FMA(x, mip_rbound, 1.0f) * (double)c->H), 0.0f, hi);
    const int ny = (int)orc_clampf((float)(0.5 * (double)ORC_FMA(y, mip_rbound, 1.0f) * (double)c->H), 0.0f, hi);
    const int nz = (int)orc_clampf((float)(0.5 * (double)ORC_FMA(z, mip_rbound, 1.0f) * (double)c->H), 0.0f, hi);
    const uint32_t index = (uint32_t)level * c->H3 + orc_morton((uint32_t)nx, (uint32_t)ny, (uint32_t)nz); /* :381 (integer form, A3) */
    const int occ = c->grid[index / 8] & (1 << (index % 8));
    if (occ) { *px = x; *py = y; *pz = z; *pdt = dt; return 1; }
    /* :393-401 distance to the next voxel boundary */
    const float tx = ORC_FMA(ORC_FMA(ORC_FMA(0.5f, orc_signf(c->dx), (float)nx + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -x) * c->rdx;
    const float ty = ORC_FMA(ORC_FMA(ORC_FMA(0.5f, orc_signf(c->dy), (float)ny + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -y) * c->rdy;
    const float tz = ORC_FMA(ORC_FMA(ORC_FMA(0.5f, orc_signf(c->dz), (float)nz + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -z) * c->rdz;
    const float tt = tt0 + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    float tc = tt0;
    do { tc += orc_clampf(tc * c->dt_gamma, c->dt_min, c->dt_max); } while (tc < tt);
    *t = tc;
    return 0;
}

/* raymarching.cu:315-483  kernel_march_rays_train.
 * The reference reserves output space with two atomicAdd()s (:408-409), so its row order is
 * scheduling dependent.  The oracle visits rays in index order, i.e. the order the atomics would
 * produce on a machine that executes threads sequentially: rays[n] = (n, exclusive-prefix, count). */
ORC_API void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                  float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                  const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                  int32_t* rays, int32_t* counter, const float* noises) {
    for (uint32_t n = 0; n < N; n++) {
        orc_ray_ctx c; orc_ctx_init(&c, rays_o + n * 3, rays_d + n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t0 = nears[n];
        t0 = ORC_FMA(orc_clampf(t0 * dt_gamma, c.dt_min, c.dt_max), noises[n], t0);   /* :354 */
        float t = t0, x, y, z, dt; uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {                                   /* :362 */
            if (orc_march_probe(&c, &t, &x, &y, &z, &dt)) { num_steps++; t += dt; }
        }
        const uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int32_t)num_steps;  /* :408 */
        const uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;                      /* :409 */
        rays[ray_index * 3] = (int32_t)n; rays[ray_index * 3 + 1] = (int32_t)point_index; rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;                                   /* :419 */
        float* px = xyzs + (size_t)point_index * 3; float* pd = dirs + (size_t)point_index * 3; float* pl = deltas + (size_t)point_index * 2;
        t = t0; uint32_t step = 0; float last_t = t;
        while (t < far && step < num_steps) {                                        /* :430 */
            if (orc_march_probe(&c, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt; pl[0] = dt; pl[1] = t - last_t; last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            }
        }
    }
}

/* raymarching.cu:907-1011  kernel_march_rays (inference) */
ORC_API void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                            const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                            uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                            float* xyzs, float* dirs, float* deltas, const float* noises) {
    (void)nears;
    ORC_PAR_FOR
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        orc_ray_ctx c; orc_ctx_init(&c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
        float* px = xyzs + (size_t)n * n_step * 3; float* pd = dirs + (size_t)n * n_step * 3; float* pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index]; const float far = fars[index];
        t = ORC_FMA(orc_clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t);      /* :952, noise indexed by slot (quirk 5) */
        float last_t = t, x, y, z, dt; uint32_t step = 0;
        while (t < far && step < n_step) {
            if (orc_march_probe(&c, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt; pl[0] = dt; pl[1] = t - last_t; last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* compositing                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* raymarching.cu:504-580 */
ORC_API void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                              uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth, float* image) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {                              /* :524 ('>' quirk 1) */
            weights_sum[index] = 0; depth[index] = 0; image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0; continue;
        }
        const float* s = sigmas + offset; const float* c = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float w = alpha * T;
            r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b);
            t += dl[1]; d = fmaf(w, t, d); ws += w;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;                                                  /* :560 tested after the update */
            s++; c += 3; dl += 2;
        }
        weights_sum[index] = ws; depth[index] = d; image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:583-645 */
ORC_API void orc_composite_rays_flex_train_forward(const float* sigmas, const float* input, const float* deltas, const int32_t* rays,
                                                   uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh, float* output) {
    float temp[128];
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        float* out = output + (size_t)index * n_channel;
        if (num_steps == 0 || offset + num_steps >= M) {                             /* :601 ('>=' quirk 1) */
            for (uint32_t i = 0; i < n_channel; i++) out[i] = 0; continue;
        }
        const float* s = sigmas + offset; const float* in = input + (size_t)offset * n_channel; const float* dl = deltas + (size_t)offset * 2;
        float T = 1.0f;
        for (uint32_t i = 0; i < n_channel; i++) temp[i] = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float w = alpha * T;
            for (uint32_t i = 0; i < n_channel; i++) temp[i] = fmaf(w, in[i], temp[i]);
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; in += n_channel; dl += 2;
        }
        for (uint32_t i = 0; i < n_channel; i++) out[i] = temp[i];
    }
}

/* raymarching.cu:681-761 ; grad_sigmas/grad_rgbs are caller-zeroed (raymarching.py:283-284) */
ORC_API void orc_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                               const float* rgbs, const float* deltas, const int32_t* rays, const float* weights_sum,
                                               const float* image, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum[index]; const float* gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2], ws_final = weights_sum[index];
        const float* s = sigmas + offset; const float* c = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        float* gs = grad_sigmas + offset; float* gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float w = alpha * T;
            r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b); ws += w;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * w; gc[1] = gi[1] * w; gc[2] = gi[2] * w;
            /* :741-746 */
            float acc = gi[0] * fmaf(T, c[0], -(r_final - r));
            acc = fmaf(gi[1], fmaf(T, c[1], -(g_final - g)), acc);
            acc = fmaf(gi[2], fmaf(T, c[2], -(b_final - b)), acc);
            acc = fmaf(gws, 1.0f - ws_final, acc);
            gs[0] = dl[0] * acc;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; gs++; gc += 3;
        }
    }
}

/* raymarching.cu:764-819 ; quirk 2: break happens BEFORE the gradient of the breaking sample is written */
ORC_API void orc_composite_rays_flex_train_backward(const float* grad_output, const float* sigmas, const float* deltas, const int32_t* rays,
                                                    uint32_t M, uint32_t N, uint32_t n_channel, float T_thresh, float* grad_input) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;
        const float* go = grad_output + (size_t)index * n_channel;
        const float* s = sigmas + offset; const float* dl = deltas + (size_t)offset * 2; float* gin = grad_input + (size_t)offset * n_channel;
        float T = 1.0f;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float w = alpha * T;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            for (uint32_t i = 0; i < n_channel; i++) gin[i] = go[i] * w;
            s++; dl += 2; gin += n_channel;
        }
    }
}

/* raymarching.cu:848-882 */
ORC_API void orc_spread_ray_to_sample(const float* input, const int32_t* rays, uint32_t M, uint32_t N, uint32_t n_channel, float* output) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0) continue;
        for (uint32_t step = 0; step < num_steps && offset + step < M; step++)
            for (uint32_t i = 0; i < n_channel; i++) output[(size_t)(offset + step) * n_channel + i] = input[(size_t)index * n_channel + i];
    }
}

/* raymarching.cu:1025-1111 */
ORC_API void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image) {
    ORC_PAR_FOR
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float* s = sigmas + (size_t)n * n_step; const float* c = rgbs + (size_t)n * n_step * 3; const float* dl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index], ws = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;                                                    /* :1064 */
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float T = 1.0f - ws;                                                /* :1074 */
            const float w = alpha * T;
            ws += w;
            t += dl[1]; d = fmaf(w, t, d);
            r = fmaf(w, c[0], r); g = fmaf(w, c[1], g); b = fmaf(w, c[2], b);
            if (T < T_thresh) break;                                                  /* :1088 T of BEFORE this sample */
            s++; c += 3; dl += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;               /* :1100-1104 */
        weights_sum[index] = ws; depth[index] = d; image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:1114-1185 ; reads weights_sum/rays_alive/rays_t, writes only output */
ORC_API void orc_composite_rays_flex(uint32_t n_alive, uint32_t n_step, uint32_t n_channel, float T_thresh, const int32_t* rays_alive,
                                     const float* rays_t, const float* sigmas, const float* input, const float* deltas,
                                     const float* weights_sum, float* output) {
    (void)rays_t;
    ORC_PAR_FOR
    for (uint32_t n = 0; n < n_alive; n++) {
        float temp[128];
        const int index = rays_alive[n];
        const float* s = sigmas + (size_t)n * n_step; const float* in = input + (size_t)n * n_step * n_channel; const float* dl = deltas + (size_t)n * n_step * 2;
        float* out = output + (size_t)index * n_channel;
        float ws = weights_sum[index];
        for (uint32_t i = 0; i < n_channel; i++) temp[i] = out[i];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
            for (uint32_t i = 0; i < n_channel; i++) temp[i] = fmaf(w, in[i], temp[i]);
            if (T < T_thresh) break;
            s++; in += n_channel; dl += 2; step++;
        }
        for (uint32_t i = 0; i < n_channel; i++) out[i] = temp[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* multiresolution hash grid                                                                    */
/* ------------------------------------------------------------------------------------------ */

#define ORC_MAX_D 5
static const uint32_t orc_primes[7] = { 1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u }; /* gridencoder.cu:42 */

/* gridencoder.cu:54-72 (ch = 0 form; returns the row index times C) */
static inline uint32_t orc_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t hashmap_size,
                                      uint32_t resolution, const uint32_t* pg) {
    uint32_t stride = 1, index = 0, d;
    for (d = 0; d < D && stride <= hashmap_size; d++) { index += pg[d] * stride; stride *= align_corners ? resolution : (resolution + 1); }
    if (gridtype == 0 && stride > hashmap_size) { index = 0; for (d = 0; d < D; d++) index ^= pg[d] * orc_primes[d]; }
    return (index % hashmap_size) * C;
}

/* gridencoder.cu:125-126 ; computed once per level on the host */
ORC_API void orc_grid_level_params(uint32_t L, float S, uint32_t H, float* scale, uint32_t* resolution) {
    for (uint32_t l = 0; l < L; l++) {
        scale[l] = exp2f((float)l * S) * (float)H - 1.0f;
        resolution[l] = (uint32_t)ceil((double)scale[l]) + 1;
    }
}

/* gridencoder.cu:75-223  kernel_grid, fp32 table.  outputs: [L,B,C]; dy_dx (optional): [B, L*D*C] */
ORC_API void orc_grid_encode_forward(const float* inputs, const float* grid, const int32_t* offsets, float* outputs,
                                     uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx,
                                     uint32_t gridtype, int align_corners) {
    float scale[64]; uint32_t res[64];
    orc_grid_level_params(L, S, H, scale, res);
    for (uint32_t level = 0; level < L; level++) {
        const float* g = grid + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        ORC_PAR_FOR
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            float* out = outputs + ((size_t)level * B + b) * C;
            float* dd = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : 0;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;    /* :98-104 */
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
                if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
                continue;
            }
            float pos[ORC_MAX_D]; uint32_t pg[ORC_MAX_D], pl[ORC_MAX_D]; float acc[8];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale[level], align_corners ? 0.0f : 0.5f);     /* :134 */
                const float fl = floorf(pos[d]); pg[d] = (uint32_t)fl; pos[d] -= (float)pg[d];
            }
            for (uint32_t ch = 0; ch < C; ch++) acc[ch] = 0;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {                          /* :145-169 */
                float w = 1;
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, hashmap_size, res[level], pl);
                for (uint32_t ch = 0; ch < C; ch++) acc[ch] = fmaf(w, g[index + ch], acc[ch]);
            }
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = acc[ch];
            if (dd) {                                                                 /* :179-222 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float ga[8]; for (uint32_t ch = 0; ch < C; ch++) ga[ch] = 0;
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale[level];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; }
                        }
                        pl[gd] = pg[gd];
                        const uint32_t il = orc_grid_index(D, C, gridtype, align_corners, hashmap_size, res[level], pl);
                        pl[gd] = pg[gd] + 1;
                        const uint32_t ir = orc_grid_index(D, C, gridtype, align_corners, hashmap_size, res[level], pl);
                        for (uint32_t ch = 0; ch < C; ch++) ga[ch] = fmaf(w, g[ir + ch] - g[il + ch], ga[ch]);
                    }
                    for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = ga[ch];
                }
            }
        }
    }
}

/* gridencoder.cu:75-175 with scalar_t = half (table and outputs are IEEE binary16 bit patterns).
 * The accumulator is a half (gridencoder.cu:142,165): acc = half(float(acc) + float(half(w * float(g)))). */
ORC_API void orc_grid_encode_forward_half(const float* inputs, const uint16_t* grid, const int32_t* offsets, uint16_t* outputs,
                                          uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                          uint32_t gridtype, int align_corners) {
    float scale[64]; uint32_t res[64];
    orc_grid_level_params(L, S, H, scale, res);
    for (uint32_t level = 0; level < L; level++) {
        const uint16_t* g = grid + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            uint16_t* out = outputs + ((size_t)level * B + b) * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) { for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0; continue; }
            float pos[ORC_MAX_D]; uint32_t pg[ORC_MAX_D], pl[ORC_MAX_D]; uint16_t acc[8];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale[level], align_corners ? 0.0f : 0.5f);
                const float fl = floorf(pos[d]); pg[d] = (uint32_t)fl; pos[d] -= (float)pg[d];
            }
            for (uint32_t ch = 0; ch < C; ch++) acc[ch] = 0;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, hashmap_size, res[level], pl);
                for (uint32_t ch = 0; ch < C; ch++)
                    acc[ch] = orc_f2h(orc_h2f(acc[ch]) + orc_h2f(orc_f2h(w * orc_h2f(g[index + ch]))));
            }
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = acc[ch];
        }
    }
}

/* gridencoder.cu:226-313  kernel_grid_backward, fp32.  grad: [L,B,C]; grad_grid caller-zeroed (grid.py:72).
 * The reference scatters with atomicAdd in scheduling order; the oracle adds in (level, b) order. */
ORC_API void orc_grid_encode_backward(const float* grad, const float* inputs, const int32_t* offsets, float* grad_grid,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                      uint32_t gridtype, int align_corners) {
    float scale[64]; uint32_t res[64];
    orc_grid_level_params(L, S, H, scale, res);
    for (uint32_t level = 0; level < L; level++) {
        float* gg = grad_grid + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            const float* gr = grad + ((size_t)level * B + b) * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;    /* :253-258 */
            if (oob) continue;
            float pos[ORC_MAX_D]; uint32_t pg[ORC_MAX_D], pl[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale[level], align_corners ? 0.0f : 0.5f);
                const float fl = floorf(pos[d]); pg[d] = (uint32_t)fl; pos[d] -= (float)pg[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, hashmap_size, res[level], pl);
                for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += w * gr[ch];     /* :309 */
            }
        }
    }
}

/* gridencoder.cu:316-342  kernel_input_backward */
ORC_API void orc_grid_input_backward(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float* dd = dy_dx + (size_t)b * L * D * C;
        float r = 0;
        for (uint32_t l = 0; l < L; l++)
            for (uint32_t ch = 0; ch < C; ch++) r = fmaf(grad[((size_t)l * B + b) * C + ch], dd[l * D * C + d * C + ch], r);
        grad_inputs[t] = r;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* spherical harmonics (shencoder.cu:27-355).                                                   */
/* The reference hard-codes 64 polynomials; this restatement evaluates the same real-SH basis  */
/* from its closed form  Y_l^m = K_l^|m| * A_|m|(x,y) * Q_l^|m|(z)  with                        */
/*   A_m = Re/Im (x+iy)^m,  Q_l^m = d^m P_l / dz^m,  Condon-Shortley phase (-1)^m,              */
/* in double precision, then rounds to float.  It is the same polynomial in (x,y,z) as the     */
/* reference's table (not merely equal on the unit sphere) -- pinned by SHEncoder_torch goldens */
/* for degree <= 5 and by the orthonormality test for degree <= 8.                             */
/* ------------------------------------------------------------------------------------------ */
static void orc_legendre_coeffs(int l, double* c /* l+1 coeffs, ascending powers */) {
    /* Bonnet recurrence on coefficient vectors */
    double p0[16] = {1}, p1[16] = {0, 1}, p2[16];
    if (l == 0) { memcpy(c, p0, sizeof(double) * 1); return; }
    for (int n = 1; n < l; n++) {
        memset(p2, 0, sizeof(p2));
        for (int i = 0; i <= n; i++) p2[i + 1] += (2.0 * n + 1) / (n + 1) * p1[i];
        for (int i = 0; i <= n - 1; i++) p2[i] -= (double)n / (n + 1) * p0[i];
        memcpy(p0, p1, sizeof(p0)); memcpy(p1, p2, sizeof(p1));
    }
    memcpy(c, p1, sizeof(double) * (l + 1));
}
static double orc_fact(int n) { double r = 1; for (int i = 2; i <= n; i++) r *= i; return r; }

ORC_API void orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx) {
    const uint32_t C2 = degree * degree;
    const double PI = 3.14159265358979323846;
    ORC_PAR_FOR
    for (uint32_t b = 0; b < B; b++) {
        const double x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
        double re[9], im[9]; re[0] = 1; im[0] = 0;
        for (int m = 1; m < 9; m++) { re[m] = x * re[m - 1] - y * im[m - 1]; im[m] = x * im[m - 1] + y * re[m - 1]; }
        for (int l = 0; l < (int)degree; l++) {
            double pc[16]; orc_legendre_coeffs(l, pc);
            for (int m = 0; m <= l; m++) {
                /* Q = d^m P_l/dz^m and its derivative, Horner-free direct sums (double) */
                double Q = 0, dQ = 0;
                for (int k = m; k <= l; k++) {
                    double f = pc[k]; for (int j = 0; j < m; j++) f *= (k - j);
                    Q += f * pow(z, k - m);
                    if (k - m >= 1) dQ += f * (k - m) * pow(z, k - m - 1);
                }
                const double K = sqrt((2.0 * l + 1) / (4 * PI) * orc_fact(l - m) / orc_fact(l + m)) * (m ? sqrt(2.0) : 1.0) * ((m & 1) ? -1.0 : 1.0);
                const uint32_t ip = (uint32_t)(l * l + l + m), in_ = (uint32_t)(l * l + l - m);
                outputs[(size_t)b * C2 + ip] = (float)(K * re[m] * Q);
                if (m) outputs[(size_t)b * C2 + in_] = (float)(K * im[m] * Q);
                if (dy_dx) {
                    float* dxp = dy_dx + (size_t)b * D * C2; float* dyp = dxp + C2; float* dzp = dyp + C2;
                    const double dre_dx = m ? m * re[m - 1] : 0, dre_dy = m ? -m * im[m - 1] : 0;
                    const double dim_dx = m ? m * im[m - 1] : 0, dim_dy = m ? m * re[m - 1] : 0;
                    dxp[ip] = (float)(K * dre_dx * Q); dyp[ip] = (float)(K * dre_dy * Q); dzp[ip] = (float)(K * re[m] * dQ);
                    if (m) { dxp[in_] = (float)(K * dim_dx * Q); dyp[in_] = (float)(K * dim_dy * Q); dzp[in_] = (float)(K * im[m] * dQ); }
                }
            }
        }
    }
}

/* shencoder.cu:358-382 ; grad_inputs is caller-zeroed and accumulated into ('+=', :378) */
ORC_API void orc_sh_encode_backward(const float* grad, uint32_t B, uint32_t D, uint32_t degree, const float* dy_dx, float* grad_inputs) {
    const uint32_t C2 = degree * degree;
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float* g = grad + (size_t)b * C2; const float* dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
        float acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], dd[ch], acc);
        grad_inputs[t] = acc;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* palette: RGB <-> HSV (palette/src/palette.cu:45-133), histogram (palette/src/bindings.cpp)   */
/* ------------------------------------------------------------------------------------------ */

/* palette.cu:45-86 ; H in [0,360), S,V in percent; IS_EQUAL = |x-y| < 1e-9 (palette.cu:19) */
ORC_API void orc_rgb_to_hsv(uint32_t n, const float* input, float* output) {
    for (uint32_t i = 0; i < n; i++) {
        const float r = input[i * 3], g = input[i * 3 + 1], b = input[i * 3 + 2];
        const float c_max = fmaxf(fmaxf(r, g), b), c_min = fminf(fminf(r, g), b), diff = c_max - c_min;
        float h, s;
        if ((double)fabsf(diff - 0.0f) < 1e-9) h = 0;
        else if ((double)fabsf(c_max - r) < 1e-9) h = (float)fmod((double)(60.0f * ((g - b) / diff) + 360.0f), 360.0);
        else if ((double)fabsf(c_max - g) < 1e-9) h = (float)fmod((double)(60.0f * ((b - r) / diff) + 120.0f), 360.0);
        else h = (float)fmod((double)(60.0f * ((r - g) / diff) + 240.0f), 360.0);
        if ((double)fabsf(c_max - 0.0f) < 1e-9) s = 0; else s = (diff / c_max) * 100.0f;
        output[i * 3] = h; output[i * 3 + 1] = s; output[i * 3 + 2] = c_max * 100.0f;
    }
}
/* palette.cu:88-133 */
ORC_API void orc_hsv_to_rgb(uint32_t n, const float* input, float* output) {
    for (uint32_t i = 0; i < n; i++) {
        const float h = input[i * 3], s = input[i * 3 + 1], v = input[i * 3 + 2];
        const float c = s / 100.0f * v / 100.0f;
        const float x = c * (1.0f - fabsf((float)fmod((double)(h / 60.0f), 2.0) - 1.0f));
        const float m = v / 100.0f - c;
        float r = 0, g = 0, b = 0;
        if (h >= 0 && h < 60) { r = c; g = x; }
        else if (h >= 60 && h < 120) { r = x; g = c; }
        else if (h >= 120 && h < 180) { g = c; b = x; }
        else if (h >= 180 && h < 240) { g = x; b = c; }
        else if (h >= 240 && h < 300) { r = x; b = c; }
        else { r = c; b = x; }
        output[i * 3] = r + m; output[i * 3 + 1] = g + m; output[i * 3 + 2] = b + m;
    }
}

/* nerf/utils.py:53-149 get_rays, deterministic core (the reference evaluates it with torch ops: linspace + 0.5, (i - cx) / fx,
 * torch.norm, directions @ R^T; reduction orders inside norm / matmul are not specified there, the fma order below is ours) */
ORC_API void orc_get_rays(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t* inds,
                          uint32_t N, float* rays_o, float* rays_d) {
    (void)H;
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t n = 0; n < N; n++) {
            const uint32_t p = inds ? (uint32_t)inds[(size_t)b * N + n] : n;
            const float xs = (((float)(p % W) + 0.5f) - cx) / fx;
            const float ys = (((float)(p / W) + 0.5f) - cy) / fy;
            const float nrm = sqrtf(fmaf(xs, xs, fmaf(ys, ys, 1.0f)));
            const float d0 = xs / nrm, d1 = ys / nrm, d2 = 1.0f / nrm;
            const float* P = poses + (size_t)b * 16;
            for (int k = 0; k < 3; k++) {
                rays_d[((size_t)b * N + n) * 3 + k] = fmaf(d2, P[k * 4 + 2], fmaf(d1, P[k * 4 + 1], d0 * P[k * 4 + 0]));
                rays_o[((size_t)b * N + n) * 3 + k] = P[k * 4 + 3];
            }
        }
}

/* palette/src/bindings.cpp:40-91 compute_RGB_histogram */
ORC_API void orc_rgb_histogram(const float* rgb, const float* weights, uint32_t n, int bpc, double* bin_weights, float* bin_centers) {
    const int num_bins = 1 << (bpc * 3);
    for (int i = 0; i < num_bins; i++) bin_weights[i] = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t index = 0;
        for (int k = 0; k < 3; k++) {
            float c = fmaxf(0.0f, fminf(0.999f, rgb[i * 3 + k]));
            index <<= bpc; index += (uint32_t)(c * (float)(1 << bpc));
        }
        bin_weights[index] += (double)weights[i];
    }
    for (int ibin = 0; ibin < num_bins; ibin++) {
        uint32_t code = (uint32_t)ibin;
        for (int k = 0; k < 3; k++) {
            const float c = (float)(code & ((1u << bpc) - 1));
            bin_centers[ibin * 3 + (2 - k)] = (c + 0.5f) / (float)(1 << bpc);
            code >>= bpc;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* tiny MLPs of the field (nerf/network.py:95-124 ; palette/network.py:156-280).                */
/* Dense layers are nn.Linear (y = x W^T [+ b]); the oracle evaluates them as a k-ordered fmaf */
/* chain in fp32, which is what the f32-input MFMA computes bit for bit.                        */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_linear(const float* x, const float* W, const float* bias, float* y, uint32_t B, uint32_t in_dim, uint32_t out_dim) {
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t o = 0; o < out_dim; o++) {
            float acc = bias ? bias[o] : 0.0f;
            for (uint32_t k = 0; k < in_dim; k++) acc = fmaf(x[(size_t)b * in_dim + k], W[(size_t)o * in_dim + k], acc);
            y[(size_t)b * out_dim + o] = acc;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* occupancy maintenance: NeRFRenderer.update_extra_state / mark_untrained_grid                 */
/* (nerf/renderer.py:467-561, :395-465).  The reference's methods are Python over torch ops; the */
/* arithmetic those ops perform ON A GPU is restated here -- separate elementwise kernels, so no */
/* contraction, and a division by a host scalar is a multiplication by its fp32 reciprocal      */
/* (ATen's div kernel for a CPU-scalar operand).  The random numbers are inputs.                */
/* ------------------------------------------------------------------------------------------ */
/* per-cascade constants as Python forms them: doubles, narrowed to fp32 where they meet the fp32 tensor (renderer.py:494-499) */
static void orc_occ_cascade(uint32_t c, float bound, uint32_t H, float* span, float* half_cell) {
    double b = ldexp(1.0, (int)c);
    if ((double)bound < b) b = (double)bound;
    const double half = b / (double)H;
    *span = (float)(b - half);
    *half_cell = (float)half;
}
/* points [count,4] = world x, y, z, global cell id (c * H^3 + morton) as int32 bits, -1 = no sample.
 * mode 0: sample s = c * H^3 + morton cell; mode 1: s = c * 2n + j, j < n: coords[c][j] (renderer.py:516-517), j >= n: the
 * (occ_rand[c][j-n] % nnz)-th cell with density_grid > 0 in ascending order (:520-523).  noise [samples,3]. */
ORC_API void orc_occupancy_points(uint32_t C, uint32_t H, float bound, int mode, uint32_t n, const float* noise, const int32_t* coords,
                                  const int32_t* occ_rand, const float* density_grid, uint32_t first, uint32_t count, float* points) {
    const uint32_t cells = H * H * H;
    const float inv_hm1 = 1.0f / (float)(H - 1);
    int32_t* occ = NULL;
    uint32_t* nnz = NULL;
    if (mode == 1) {
        occ = (int32_t*)malloc((size_t)C * cells * sizeof(int32_t));
        nnz = (uint32_t*)calloc(C, sizeof(uint32_t));
        for (uint32_t c = 0; c < C; c++)
            for (uint32_t m = 0; m < cells; m++)
                if (density_grid[(size_t)c * cells + m] > 0.0f) occ[(size_t)c * cells + nnz[c]++] = (int32_t)m;
    }
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t s = first + i;
        uint32_t c, cell;
        int live = 1;
        if (mode == 0) { c = s / cells; cell = s % cells; }
        else {
            c = s / (2 * n);
            const uint32_t j = s % (2 * n);
            if (j < n) {
                const int32_t* q = coords + ((size_t)c * n + j) * 3;
                cell = orc_morton((uint32_t)q[0], (uint32_t)q[1], (uint32_t)q[2]);
            } else if (nnz[c] > 0) cell = (uint32_t)occ[(size_t)c * cells + (uint32_t)occ_rand[(size_t)c * n + (j - n)] % nnz[c]];
            else { cell = 0; live = 0; }
        }
        float span, half_cell;
        orc_occ_cascade(c, bound, H, &span, &half_cell);
        const uint32_t q[3] = {orc_compact_bits(cell), orc_compact_bits(cell >> 1), orc_compact_bits(cell >> 2)};
        for (int d = 0; d < 3; d++) {
            const float x = (2.0f * (float)q[d]) * inv_hm1 - 1.0f;                        /* renderer.py:491 */
            const float r = (noise[(size_t)s * 3 + d] * 2.0f - 1.0f) * half_cell;         /* :499 */
            points[(size_t)i * 4 + d] = x * span + r;                                    /* :497, :499 */
        }
        const int32_t id = live ? (int32_t)(c * cells + cell) : -1;
        memcpy(&points[(size_t)i * 4 + 3], &id, 4);
    }
    free(occ); free(nnz);
}
/* candidates (sigma * density_scale per sample, cells from orc_occupancy_points) -> tmp grid (largest candidate of a cell; the reference
 * keeps an unspecified one, renderer.py:505/537) -> EMA-max (:541-542) -> mean (:543) -> min(mean, density_thresh) (:549) -> packbits (:550).
 * state[0] = mean, state[1] = threshold. */
ORC_API void orc_occupancy_commit(uint32_t C, uint32_t H, float* density_grid, const float* points, const float* candidates, uint32_t count,
                                  float decay, float density_thresh, uint8_t* bitfield, float* state) {
    const size_t total = (size_t)C * H * H * H;
    float* tmp = (float*)malloc(total * sizeof(float));
    for (size_t i = 0; i < total; i++) tmp[i] = -1.0f;
    for (uint32_t i = 0; i < count; i++) {
        int32_t id, a, b;
        memcpy(&id, &points[(size_t)i * 4 + 3], 4);
        if (id < 0) continue;
        memcpy(&a, &tmp[id], 4); memcpy(&b, &candidates[i], 4);
        if (b > a) tmp[id] = candidates[i];        /* order of the bit patterns = order of non-negative floats; -1 loses to all of them */
    }
    double sum = 0.0;
    for (size_t i = 0; i < total; i++) {
        if (density_grid[i] >= 0.0f && tmp[i] >= 0.0f) density_grid[i] = fmaxf(density_grid[i] * decay, tmp[i]);
        sum += (double)(density_grid[i] > 0.0f ? density_grid[i] : 0.0f);
    }
    const float mean = (float)(sum * (1.0 / (double)total));
    const float thresh = density_thresh < mean ? density_thresh : mean;
    if (state) { state[0] = mean; state[1] = thresh; }
    orc_packbits(density_grid, (uint32_t)(total / 8), thresh, bitfield);
    free(tmp);
}
/* nerf/renderer.py:395-465: cam = (world - t) @ R per pose (an i-ordered fma chain here), frustum test with a cell of slack, count / too_close */
ORC_API void orc_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H, float bound,
                                     float min_near, int filter_close_point, float* density_grid, int32_t* n_marked) {
    const uint32_t cells = H * H * H;
    const float inv_hm1 = 1.0f / (float)(H - 1), kx = cx / fx, ky = cy / fy;
    int32_t marked = 0;
    for (uint32_t c = 0; c < C; c++) {
        float span, half_cell;
        orc_occ_cascade(c, bound, H, &span, &half_cell);
        const float slack = half_cell * 2.0f;
        for (uint32_t cell = 0; cell < cells; cell++) {
            const uint32_t q[3] = {orc_compact_bits(cell), orc_compact_bits(cell >> 1), orc_compact_bits(cell >> 2)};
            float w[3];
            for (int d = 0; d < 3; d++) w[d] = ((2.0f * (float)q[d]) * inv_hm1 - 1.0f) * span;
            uint32_t seen = 0, close = 0;
            for (uint32_t b = 0; b < B; b++) {
                const float* P = poses + (size_t)b * 16;
                const float d0 = w[0] - P[3], d1 = w[1] - P[7], d2 = w[2] - P[11];
                const float X = fmaf(d2, P[8], fmaf(d1, P[4], d0 * P[0]));
                const float Y = fmaf(d2, P[9], fmaf(d1, P[5], d0 * P[1]));
                const float Z = fmaf(d2, P[10], fmaf(d1, P[6], d0 * P[2]));
                const int in = Z > 0.0f && fabsf(X) < kx * Z + slack && fabsf(Y) < ky * Z + slack;
                seen += (uint32_t)in;
                close += (uint32_t)(in && Z < min_near);
                if (filter_close_point) close += (uint32_t)(sqrtf(X * X + Y * Y + Z * Z) < min_near);
            }
            if (seen == 0 || close != 0) { density_grid[(size_t)c * cells + cell] = -1.0f; marked++; }
        }
    }
    if (n_marked) *n_marked = marked;
}
