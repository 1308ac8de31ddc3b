// pnr_common.hpp -- device-side scalar helpers and launch plumbing shared by the gfx950 kernels.
//
// Canonical scalar spec (DESIGN.md "Scalar semantics"): IEEE fp32, translation units are built
// with -ffp-contract=off, and every contraction the reference's nvcc build performs is written
// as an explicit fmaf().  Wavefront width is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/pnr.h"

#define PNR_WAVE 64

namespace pnr {

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PNR_OK : PNR_ERR_LAUNCH;
}
inline hipStream_t as_stream(pnr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// One process may drive several GPUs: function attributes and events belong to a device, so one-time set-up is tracked per device id.
constexpr int kMaxDevices = 64;
inline int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0; return d; }
// raise a kernel's dynamic-LDS limit once per device; false = the runtime refused
template <typename K>
inline bool ensure_dynamic_lds(K kernel, uint32_t bytes, bool* done /* [kMaxDevices] */) {
    const int d = current_device();
    if (done[d]) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done[d] = true;
    return true;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf(float x) { return copysignf(1.0f, x); }

// 10-bit-per-axis bit interleave (reference raymarching.cu:59-84)
__host__ __device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t gather3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// cascade level selectors (reference raymarching.cu:45-57)
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (dt * H) * 0.5f;  // the reference multiplies by a double 0.5: exact either way
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}

// Eight independent global loads issued back to back, each returning into registers of its own.  On gfx950 a load whose destination
// overlaps its own address registers (`global_load_dwordx2 v[6:7], v[6:7], off` -- what the compiler writes once registers are scarce) is
// slow: the lookup launch of the frame loop took 75.6 instead of 65.6 us with otherwise identical code ("March: the hosted
// tail").  All eight addresses are formed first (one empty asm statement with all of them as operands pins that point; it also hides where
// a pointer came from, hence the address-space cast: without it the loads become flat_load), nothing is scheduled into the group, and the
// addresses stay alive past the loads.
typedef float f32x2 __attribute__((ext_vector_type(2)));   // (built-in vector types: a load through an address-space pointer needs no operator=)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
struct GlobalPtr { typedef const T __attribute__((address_space(1))) * type; };
template <typename T>
__device__ __forceinline__ void load8_fresh(const T* (&p)[8], T (&v)[8]) {
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = *(typename GlobalPtr<T>::type)(uintptr_t)p[i];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no staging registers, no ds_write pass).  The LDS
// destination of a wave is its first lane's address + lane x 16, so `lds` must be lane-linear -- a plain copy loop over float4 indices is.
// Completion: s_waitcnt vmcnt(0) (lds_copy_wait) and then the workgroup barrier; a wave must not end before its copies have landed.
__device__ __forceinline__ void lds_copy16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void lds_copy_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }   // LDS aperture: the low 32 bits are the LDS offset

// wave64 inclusive prefix sum of an int (DPP-free, shuffle based)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & (PNR_WAVE - 1);
#pragma unroll
    for (int off = 1; off < PNR_WAVE; off <<= 1) {
        int n = __shfl_up(v, off, PNR_WAVE);
        if (lane >= off) v += n;
    }
    return v;
}

}  // namespace pnr

// Run-time switches for A/B measurements and tests (pnr_set_option; initialised from PNR_NO_BLOCK_SKIP / PNR_NO_AUX_FUSION).
// Neither changes any result.
extern int g_opt_block_skip;   // exact jumps over empty 4^3 / 8^3 / 16^3 blocks in the march
extern int g_opt_coop_march;   // frame loops: wave-cooperative march tail (frame.hip: march_coop_tail)
extern int g_opt_hosted_tail;  // frame loops: rays a march launch has not finished within its probe budget are marched by the first workgroups of the lookup launch (frame.hip: hosted_march_tail)
extern int g_opt_march_budget, g_opt_march_budget0;   // that budget in probe rounds: later launches / a frame's first launch (0 = the first launch keeps the in-wave cooperative tail)
extern int g_opt_march_blocks;   // frame loops: workgroup cap of a budgeted march launch
extern int g_opt_aux_fusion;   // PaletteNeRF frame loop: aux composite inside the field kernel
extern int g_opt_composite_fusion;   // NeRF frame loop: 1 = n_step == 1 iterations composited inside the field kernel, 2 = every iteration (no composite launch)
extern int g_opt_dynamic_tiles;      // frame loops: field kernels hand wave tiles out through a device counter instead of a static schedule
extern int g_opt_grid_fast;          // stand-alone lookup op: k_grid_fwd_d3c2 for D = 3, C = 2 (gridencoder.hip)
extern int g_opt_train_coop;  // training march: cooperative counting pass
extern int g_opt_mlp_f16x3;  // training MLPs: split-fp16 matrix products (default) or exact fp32
extern int g_opt_coarse_image;  // binned table gradient: LDS images for the coarsest levels
extern int g_opt_scatter_staged;  // binned table gradient: LDS-ordered, coalesced record writes
extern int g_opt_cell_merge;  // binned table gradient: cell-run merging on mid levels
extern int g_opt_grid_nt;
extern int g_opt_flex_coop;   // drop-in composite_rays_flex: workgroup-cooperative kernel (composite.hip)
extern int g_opt_iteration_margin;   // frame loops: spare iterations enqueued beyond the previous frame's count before the first host look
