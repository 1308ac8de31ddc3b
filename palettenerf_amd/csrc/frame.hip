// frame.hip -- device-driven inference frame of the NeRF path (MI355X-first; no reference counterpart as
// one call).  It runs the loop of nerf/renderer.py:344-380
//     while step < max_steps: n_step = max(min(N // n_alive, 8), 1); march; field; composite; compact
// with the SAME schedule and the same per-ray arithmetic, but
//   * n_alive / n_step / step live in a device control block that every kernel reads; the host
//     enqueues iterations back to back and looks at the block only every few iterations (one 64-byte
//     read-back) instead of synchronising on a boolean-mask compaction every iteration;
//   * the alive list is compacted on the device (wave64 ballot + prefix sum, stable);
//   * no staging buffers are zero-filled: the march writes the delta == 0 sentinel itself, padding
//     rows do not exist, dead slots are skipped by the grid and field kernels;
//   * world -> [0,1] normalisation, density_scale and the sample counter are folded into kernels.
// Per iteration: march (+ compaction of the previous alive list + this iteration's schedule; the rays it does not finish within its probe
// budget go to a queue), grid (level-major, L2-resident tables; its first workgroups finish the queued rays: hosted march tail), field
// (MFMA + the iteration's whole compositing step + the survivor counts of the compaction) = 3 launches, no host round trip.
// (k_frame_composite remains for pnr_set_option("composite_fusion", 0 / 1) and for PaletteNeRF shapes whose aux rows do not fit the LDS.)
#include "pnr_common.hpp"
#include "march_core.hpp"
#include "grid_core.hpp"
#include "field_core.hpp"
#include <hip/hip_ext.h>
#include <vector>
#include <stdlib.h>

// palette_field.hip (internal): device image of the edit parameters
uint64_t pnr_internal_edit_device_bytes();
int pnr_internal_edit_upload(const pnr_palette_edit* edit, void* dst, hipStream_t s);

#ifndef PNR_FRAME_LEVEL_PAIRS
#define PNR_FRAME_LEVEL_PAIRS 1
#endif
#ifndef PNR_FRAME_LEVEL_PAIRS_PAL
#define PNR_FRAME_LEVEL_PAIRS_PAL 0
#endif
namespace pnr {

struct FrameCtl {  // 64 bytes, two copies ping-ponged by iteration parity
    int32_t n_alive, n_step, step, done;
    int32_t iterations, pad0;
    unsigned long long rendered;  // march-emitted samples (delta > 0)
    unsigned long long rows;      // evaluated rows, sum of n_alive * n_step
    int32_t pad1[6];
};
static_assert(sizeof(FrameCtl) == 64, "FrameCtl layout");

constexpr uint32_t kRayBlock = 256;
constexpr uint32_t kHdr = 4;  // scratch header ints before the per-chunk counts

// ------------------------------------------------------------------------------------------
// Hosted march tail (round 3).  A march launch lasts as long as its slowest rays -- 0.5 % of a later lego launch need more than two
// probes (profiles/march_stats.py), and the launch waited 20-30 us for them at ~1.3 us per dependent probe.  Kernels of one stream
// cannot overlap (hipExtAnyOrderLaunch is not honoured on gfx9: profiles/micro/any_order.hip), so the overlap happens INSIDE the next
// launch: k_frame_march<.., 2> gives every ray `budget` probe rounds, writes the rays that are still marching to a queue in the frame
// workspace (their exact t / last_t / samples so far) and ends; the first kHostedBlocks workgroups of the lookup launch (they are
// dispatched first) take those rays 64 to a wave, run the very same march_probe() loop from the recorded state -- the same rows, bit
// for bit -- and look the new rows up themselves on all 16 levels, while the other workgroups do the lookup of everybody else's rows
// and skip the handed-over ones (a per-row flag the march wrote: stable for the whole launch, no race).  Field, composite and
// compaction see a complete iteration.  Schedule, sample rows and per-ray arithmetic are unchanged.
// ------------------------------------------------------------------------------------------
struct StragglerRec { int32_t n, index, step; float t, last_t, far; int32_t pad0, pad1; };   // 32 bytes
static_assert(sizeof(StragglerRec) == 32, "StragglerRec layout");
constexpr int kQueueCtrs = 4;             // counters per set (two sets ping-ponged by iteration parity): [0] rays queued
constexpr uint32_t kHostedBlocks = 256;   // workgroups of a lookup launch that serve the queue first (one wave per SIMD of the chip)
#ifndef PNR_MAX_MARCH_BLOCKS
#define PNR_MAX_MARCH_BLOCKS 4096   // (2048: a frame of 2 500 chunks gave 452 workgroups a second chunk behind their block barriers; first launch 88.1 -> 86.4 us on average)
#endif
constexpr uint32_t kMaxMarchBlocks = PNR_MAX_MARCH_BLOCKS;
// What the hosted march tail needs on top of the lookup's own arguments.  The frame-constant part lives in the workspace (k_frame_begin
// writes it there from its own arguments): the lookup kernels' argument block stays the size it had.
struct HostedConst {
    const int32_t* qctr_all;      // [2][kQueueCtrs]: per iteration parity, [0] rays queued
    const StragglerRec* qrecs;
    const float* rays_o; const float* rays_d;
    const uint8_t* bitfield; const uint32_t* mip;
    MarchParams p;
    float* xyzs; float* dirs; float* deltas;
    int32_t* partials[2];         // per iteration parity: the march's per-workgroup sample counts; the hosted workgroups' follow at [partial_base]
};
struct HostedArgs {
    const HostedConst* hc;
    const uint8_t* rowflag;       // [rows] 1 = the row belongs to a queued ray (the ordinary workgroups skip it); nullptr = no flags this iteration
    uint32_t blocks;              // workgroups at the front of the launch that serve the straggler queue (0: none)
    uint32_t partial_base;        // first slot of the hosted workgroups in partials[parity] (= workgroups of this iteration's march launch)
    uint32_t gx, n_tab;           // with hosted workgroups the launch is one-dimensional: blocks + gx x 16 levels x n_tab tables (no empty workgroups next to the hosted ones)
};


__device__ __forceinline__ int schedule_n_step(int N, int n_alive) {  // nerf/renderer.py:364
    return n_alive > 0 ? max(min(N / n_alive, 8), 1) : 1;
}

// A frame's first launch.  Per ray (thread i = processing slot i): the optional gather into processing order (pnr_nerf_frame_args::ray_order -- the frame
// runs on copies of the per-ray inputs in that order and on outputs kept in that order, scattered back to ray ids at the end; per-ray results do not
// depend on the slot a ray occupies (no perturbation on this path), and listing rays tile by tile (8x8 pixels per wave) keeps the 64 rays of a wave
// spatially compact, which the hash-grid gathers and the march like), the optional near / far against the box (pnr_nerf_frame_args::aabb:
// pnr_near_far_from_aabb's arithmetic, written by ray id as that op writes them), and the state "in front of iteration 0" as k_frame_march expects it
// from a previous iteration: a full alive list whose chunks all survive (identity compaction), nothing marched yet.  (Three launches until round 5:
// near/far from the caller, sort, init -- at an eighth of a frame each costs more in launch than in work.)
struct FrameBegin {
    const int32_t* order;                       // NULL: slot = ray id
    const float *rays_o, *rays_d;
    const float *nears_in, *fars_in;            // read when aabb == NULL
    const float* aabb; float min_near;          // else: computed here ...
    float *nears_out, *fars_out;                // ... and written by ray id
    float *so, *sd, *sf;                        // order != NULL: the inputs in processing order (the near only seeds rays_t)
    float* aux_zero; uint32_t aux_stride;       // PaletteNeRF: the frame's aux map (processing order), zeroed here; stride a multiple of 4 floats
};
__global__ void __launch_bounds__(kRayBlock) k_frame_begin(uint32_t N, FrameBegin fb, int32_t* __restrict__ alive,
                                                           float* __restrict__ rays_t, float* __restrict__ weights_sum, float* __restrict__ depth,
                                                           float* __restrict__ image, FrameCtl* __restrict__ ctl, int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ scratch_hdr, int32_t* __restrict__ qctr_all, HostedConst hc,
                                                           HostedConst* __restrict__ hc_out) {
    const uint32_t i = blockIdx.x * kRayBlock + threadIdx.x;
    if (i < N) {
        const uint32_t r = fb.order ? (uint32_t)fb.order[i] : i;
        float near, far;
        if (fb.order || fb.aabb) {
            float o[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { o[k] = fb.rays_o[(size_t)r * 3 + k]; d[k] = fb.rays_d[(size_t)r * 3 + k]; }
            if (fb.aabb) {
                near_far_of(o[0], o[1], o[2], d[0], d[1], d[2], fb.aabb, fb.min_near, near, far);
                fb.nears_out[r] = near; fb.fars_out[r] = far;
            } else {
                near = fb.nears_in[r]; far = fb.fars_in[r];
            }
            if (fb.order) {
#pragma unroll
                for (int k = 0; k < 3; k++) { fb.so[i * 3 + k] = o[k]; fb.sd[i * 3 + k] = d[k]; }
                fb.sf[i] = far;
            }
        } else {
            near = fb.nears_in[r];
        }
        alive[i] = (int32_t)i;   // the reference's arange (nerf/renderer.py:352)
        rays_t[i] = near;
        weights_sum[i] = 0.0f; depth[i] = 0.0f;
        image[i * 3] = 0.0f; image[i * 3 + 1] = 0.0f; image[i * 3 + 2] = 0.0f;
    }
    if (fb.aux_zero && blockIdx.x * kRayBlock < N) {    // the workgroup's rows are one contiguous run: 16 bytes per lane and store, coalesced
        const uint32_t row0 = blockIdx.x * kRayBlock, rows = N - row0 < kRayBlock ? N - row0 : kRayBlock;
        float4* base = reinterpret_cast<float4*>(fb.aux_zero + (size_t)row0 * fb.aux_stride);
        const uint32_t n4 = rows * (fb.aux_stride / 4);
        for (uint32_t k = threadIdx.x; k < n4; k += kRayBlock) base[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (threadIdx.x == 0 && blockIdx.x * kRayBlock < N) counts[blockIdx.x] = (int32_t)(N - blockIdx.x * kRayBlock < kRayBlock ? N - blockIdx.x * kRayBlock : kRayBlock);
    if (i == 0) {
        FrameCtl c = {};
        c.n_alive = (int32_t)N; c.n_step = 0; c.iterations = -1; c.done = N == 0;
        ctl[0] = c; ctl[1] = c;
        scratch_hdr[1] = 0;   // the fp16-range overflow flag of this frame
        for (int k = 0; k < 2 * kQueueCtrs; k++) qctr_all[k] = 0;   // both counter sets of the straggler queue
        *hc_out = hc;
    }
}
// The epilogue of run_cuda (nerf/renderer.py:382-384) on request: the same fp32 operations as the torch expressions
//   image + (1 - weights_sum).unsqueeze(-1) * bg_color ;  torch.clamp(depth - nears, min=0) / (fars - nears)
struct FrameFinish {
    int on; float bg[3]; const float* bg_map; const float* nears; const float* fars;
    float* depth_raw;        // optional: the un-normalised depth by ray id (palette/renderer.py:522 depth_origin)
    const FrameCtl* ctl;     // optional: the launch is a no-op unless this control block says the frame is done (the speculative launch behind a chunk)
};
__device__ __forceinline__ float finish_channel(const FrameFinish& f, float v, float ws, uint32_t ray, uint32_t ch) {
    const float t = 1.0f - ws;
    const float bg = f.bg_map ? f.bg_map[(size_t)ray * 3 + ch] : f.bg[ch];
    return v + t * bg;
}
__device__ __forceinline__ float finish_depth(const FrameFinish& f, float d, uint32_t ray) {
    return fmaxf(d - f.nears[ray], 0.0f) / (f.fars[ray] - f.nears[ray]);
}

// The frame's last launch: outputs from processing order back to ray ids (order == NULL: in place, sws == ws, ...) and run_cuda's epilogue on request
// (fin.on bit 0: image, bit 1: depth, bit 2: the aux row's first three columns = PaletteNeRF's direct_rgb, palette/renderer.py:529-540)
template <uint32_t LANES>   // lanes per ray: 16 (32 for rows of more than 64 floats) with an aux row to move (float4 each), 4 without
__global__ void __launch_bounds__(kRayBlock) k_frame_unsort_outputs(uint32_t N, const int32_t* __restrict__ order, const float* __restrict__ sws,
                                                                    const float* __restrict__ sdepth, const float* __restrict__ simage,
                                                                    const float* __restrict__ saux, uint32_t aux_stride, float* __restrict__ ws,
                                                                    float* __restrict__ depth, float* __restrict__ image, float* __restrict__ aux,
                                                                    FrameFinish fin) {
    if (fin.ctl && !fin.ctl->done) return;
    const uint32_t i = (blockIdx.x * kRayBlock + threadIdx.x) / LANES, q = threadIdx.x % LANES;
    if (i >= N) return;
    const uint32_t r = order ? (uint32_t)order[i] : i;
    if (q == 0) {
        const float d = sdepth[i];
        if (order) ws[r] = sws[i];
        if (fin.depth_raw) fin.depth_raw[r] = d;
        depth[r] = (fin.on & 2) ? finish_depth(fin, d, r) : d;
    }
    if (q < 3) { const float v = simage[(size_t)i * 3 + q]; image[(size_t)r * 3 + q] = (fin.on & 1) ? finish_channel(fin, v, sws[i], r, q) : v; }
    if (saux) {
        if (q * 4 < aux_stride) {
            float4 v = *reinterpret_cast<const float4*>(saux + (size_t)i * aux_stride + q * 4);
            if (q == 0 && (fin.on & 4)) {
                const float w = sws[i];
                v.x = finish_channel(fin, v.x, w, r, 0); v.y = finish_channel(fin, v.y, w, r, 1); v.z = finish_channel(fin, v.z, w, r, 2);
            }
            *reinterpret_cast<float4*>(aux + (size_t)r * aux_stride + q * 4) = v;
        }
    } else if (aux && (fin.on & 4) && q < 3) {   // in place
        aux[(size_t)r * aux_stride + q] = finish_channel(fin, aux[(size_t)r * aux_stride + q], sws[i], r, q);
    }
}

// reference raymarching.cu:907-1011, n_alive / n_step from the control block; also writes the delta == 0
// sentinel of unfilled slots (the reference relies on zero-initialised buffers) and counts emitted samples.
#ifdef PNR_MARCH_STATS
__device__ unsigned long long g_march_stats[8];
__device__ unsigned int g_march_max[64];   // per iteration: max probes of any ray
__device__ unsigned int g_march_hist[2][32];   // probes per ray and launch: [0] first launch of a frame, [1] later launches (last bin: 31 or more)
__device__ unsigned int g_march_kinds[64][8];   // per iteration: probe kinds of (one of) the slowest rays  // probes, empty probes, (unused), ray-launches
#endif
// xor-shuffle whose lane index is formed on the spot: left to the compiler, the six `(lane ^ off) * 4` addresses of a wave reduction are
// computed once, kept for every later reduction of the kernel -- and spilled to scratch when registers are scarce (k_frame_march at six waves
// per SIMD: 7 of its 23 spill slots, reloaded on the prologue's critical path)
__device__ __forceinline__ uint32_t shfl_xor_fresh(uint32_t v, int off) {
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ off) << 2, (int)v);
}
__device__ __forceinline__ unsigned long long shfl_xor_fresh(unsigned long long v, int off) {
    return ((unsigned long long)shfl_xor_fresh((uint32_t)(v >> 32), off) << 32) | shfl_xor_fresh((uint32_t)v, off);
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* sh /* [kRayBlock / 64] */) {
    for (int off = PNR_WAVE / 2; off > 0; off >>= 1) v += shfl_xor_fresh(v, off);
    __syncthreads();
    if ((threadIdx.x & (PNR_WAVE - 1)) == 0) sh[threadIdx.x / PNR_WAVE] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int wv = 0; wv < (int)(kRayBlock / PNR_WAVE); wv++) t += sh[wv];
    return t;
}


// One launch per iteration does the stable compaction of the previous iteration's alive list, the schedule of this iteration and the
// march itself.  Every workgroup totals the per-chunk survivor counts (k_frame_composite / k_frame_field wrote them; a few coalesced
// loads per thread out of L2) -- that gives n_alive and n_step of this iteration without a separate launch -- and derives the output
// offset of its chunk by summing the counts in front of it; a surviving ray is written to its compacted slot and marched from there.
// Workgroup 0 also adds the previous march's sample partials and writes this iteration's control block for the kernels that follow.
#ifdef PNR_MARCH_TIMING
// instrumented builds only: per-wave time stamps (wall_clock64: 100 MHz) of ONE iteration's march launch (g_march_timing_iter)
constexpr int kTimingWaves = 8192;
__device__ unsigned long long g_march_timing[kTimingWaves * 8];
__device__ int g_march_timing_iter = 3;
#define PNR_STAMP(k) do { if (timing && (threadIdx.x & 63) == 0) g_march_timing[(size_t)twave * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PNR_STAMP(k) do { } while (0)
#endif
// (the wave-cooperative march tail -- march_coop_tail, CoopShared, kCoopRays -- lives in march_core.hpp: the drop-in march kernels use it too)
#ifndef PNR_MARCH_WAVES
#define PNR_MARCH_WAVES 4     // waves per SIMD the march kernel WITH the in-wave cooperative tail is compiled for (register budget 512 / PNR_MARCH_WAVES): it needs 114 VGPRs;
                              // squeezed into 80 (6 waves, all chunks of a later iteration resident at once) it spills and the lego frame is 0.3 ms slower (4.38 vs 4.07 ms)
#endif
#ifndef PNR_MARCH2_JUMPS
#define PNR_MARCH2_JUMPS true   // the budgeted march attempts the exact block jumps (false: plain cell steps only -- fewer registers; the rays that would have jumped end up in the queue)
#endif
#ifndef PNR_MARCH_WAVES_Q
#define PNR_MARCH_WAVES_Q 5   // waves per SIMD of the budgeted march (MODE 2: no in-wave tail; 80 VGPRs spill 24 registers: 23 us per launch against 17.8 at 102)
#endif
// MODE 1: every ray is finished here, the last rays of a wave cooperatively (march_coop_tail; p.coop == 0: plain SIMT loop).
// MODE 2: `budget` probe rounds per ray, then the rays still marching go to the straggler queue (hosted_march_tail finishes them).
template <bool MIP, bool POW2, int MODE>
__global__ void __launch_bounds__(kRayBlock) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? PNR_MARCH_WAVES_Q : PNR_MARCH_WAVES))) k_frame_march(const FrameCtl* __restrict__ prev, FrameCtl* __restrict__ cur, const int32_t* __restrict__ alive_prev,
                                                           int32_t* __restrict__ rays_alive, const int32_t* __restrict__ counts /* the previous iteration's survivors per chunk */,
                                                           int32_t* __restrict__ counts_cur /* this iteration's: cleared here, filled by its field / composite launch */,
                                                           int32_t* __restrict__ scratch_rw, uint32_t N, uint32_t max_steps,
                                                           const int32_t* __restrict__ partials_prev, uint32_t n_partials_prev,
                                                           const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, MarchParams p, const uint8_t* __restrict__ grid,
                                                           const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                           float* __restrict__ deltas, const uint32_t* __restrict__ mip,
                                                           int32_t* __restrict__ emitted_partials /* [gridDim.x] */, uint32_t budget,
                                                           int32_t* __restrict__ qctr_all /* [2][kQueueCtrs] */, StragglerRec* __restrict__ qrecs,
                                                           uint8_t* __restrict__ rowflag /* [rows]: 1 = the row belongs to a queued ray */) {
    __shared__ int csum[kRayBlock / PNR_WAVE];
    __shared__ unsigned long long red[kRayBlock / PNR_WAVE];
    __shared__ CoopShared coop[MODE == 1 ? kRayBlock / PNR_WAVE : 1];
#ifdef PNR_MARCH_TIMING
    const bool timing = prev->iterations + 1 == g_march_timing_iter && blockIdx.x * (kRayBlock / PNR_WAVE) < (uint32_t)kTimingWaves;
    const uint32_t twave = blockIdx.x * (kRayBlock / PNR_WAVE) + threadIdx.x / PNR_WAVE;
#endif
    PNR_STAMP(0);
    // Every independent load of the prologue is issued before the first wait: the chunk counts (their total is n_alive, the part in front of
    // this workgroup's chunk its output offset), this thread's slot of the previous alive list and the occupancy mip on its way to LDS --
    // one trip to L2 instead of three dependent ones (profiles/march_timing.py on a PNR_MARCH_TIMING build: the median wave has its compacted
    // slot 6.2 -> 3.9 us after the launch; the launch itself is bounded by its few longest rays, ~1.3 us per probe of the slowest lane).
    constexpr int kCountLoads = 8, kMipLoads = 5;
    const uint32_t max_chunks = (N + kRayBlock - 1) / kRayBlock;   // the counts array and the alive lists are sized for N rays: loads inside that
    int32_t cv[kCountLoads];                                       // range need not wait for the control block (masked once it has arrived)
#pragma unroll
    for (int u = 0; u < kCountLoads; u++) {
        const uint32_t j = threadIdx.x + (uint32_t)u * kRayBlock;
        cv[u] = j < max_chunks ? counts[j] : 0;
    }
    const uint32_t slot0 = blockIdx.x * kRayBlock + threadIdx.x;
    int index0 = slot0 < N ? alive_prev[slot0] : -1;
    extern __shared__ uint32_t mip_smem[];
    const uint32_t mip_n = MIP ? 2 * p.mip_words + 8 : 0;   // 'any' mask, 'all' mask, occupied box (stage_mip's layout)
    const bool mip_fast = MIP && mip_n <= (uint32_t)kMipLoads * kRayBlock * 4 && (mip_n & 3u) == 0;
    // (global_load_lds: 16 bytes per lane straight into LDS, destination = wave-uniform base + lane x 16 -- exactly this layout; no staging
    // registers (20 per lane before) and no ds_write pass.  A wave must not end while such a load is in flight: the early exits below wait.)
    if (mip_fast) {
#pragma unroll
        for (int u = 0; u < kMipLoads; u++) {
            const uint32_t i = (threadIdx.x + (uint32_t)u * kRayBlock) * 4;
            if (i < mip_n) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(&mip[i]),
                                                            (__attribute__((address_space(3))) void*)(&mip_smem[i]), 16, 0, 0);
        }
    }
    if (prev->done) { if (blockIdx.x == 0 && threadIdx.x == 0) *cur = *prev; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    const uint32_t n_prev = (uint32_t)prev->n_alive, nchunks = (n_prev + kRayBlock - 1) / kRayBlock;
    // The host sizes the launch for its upper bound of n_alive (N until it has looked at the control block): the workgroups beyond the last chunk
    // leave at once -- a third of a typical launch, which otherwise held wave slots through the whole prologue.
    if (blockIdx.x >= nchunks && blockIdx.x != 0) { if (threadIdx.x == 0) emitted_partials[blockIdx.x] = 0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (slot0 >= n_prev) index0 = -1;
    unsigned long long part = 0;   // high word: all chunks, low word: the chunks in front of this workgroup's first one
#pragma unroll
    for (int u = 0; u < kCountLoads; u++) {
        const uint32_t j = threadIdx.x + (uint32_t)u * kRayBlock;
        if (j >= nchunks) cv[u] = 0;
        part += ((unsigned long long)(uint32_t)cv[u] << 32) | (j < blockIdx.x ? (unsigned long long)(uint32_t)cv[u] : 0ull);
    }
    for (uint32_t j = threadIdx.x + kCountLoads * kRayBlock; j < nchunks; j += kRayBlock) {
        const unsigned long long v = (unsigned long long)(uint32_t)counts[j];
        part += (v << 32) | (j < blockIdx.x ? v : 0ull);
    }
    const unsigned long long sums = block_sum_u64(part, red);
    const uint32_t n_alive = (uint32_t)(sums >> 32);
    const uint32_t n_step = (uint32_t)schedule_n_step((int)N, (int)n_alive);
    const int step_now = prev->step + prev->n_step;
    const bool done = n_alive == 0 || (uint32_t)step_now >= max_steps;
    if (blockIdx.x == 0) {
        unsigned long long e = 0;
        for (uint32_t j = threadIdx.x; j < n_partials_prev; j += kRayBlock) e += (unsigned long long)partials_prev[j];
        const unsigned long long emitted_prev = block_sum_u64(e, red);
        if (threadIdx.x == 0) {
            FrameCtl c = *prev;
            c.rendered += emitted_prev;
            c.rows += (unsigned long long)n_prev * (unsigned long long)prev->n_step;
            c.step = step_now;
            c.iterations += 1;
            c.n_alive = (int32_t)n_alive;
            c.n_step = (int32_t)n_step;
            c.done = done;
            c.pad0 = scratch_rw[1];   // fp16-range overflow flag raised by a field launch of an earlier iteration (0 = none)
            *cur = c;
            scratch_rw[2] = 0;     // the wave-tile counter of this iteration's field launch
            qctr_all[((c.iterations + 1) & 1) * kQueueCtrs] = 0;   // the NEXT iteration's straggler count (this iteration's set is in use; the other one was last read by the previous lookup launch)
        }
    }
    if (done || blockIdx.x >= nchunks) { if (threadIdx.x == 0) emitted_partials[blockIdx.x] = 0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    PNR_STAMP(1);
    const uint32_t* mip_lds = nullptr;
    if (mip_fast) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the mip has landed in LDS
        __syncthreads();
        mip_lds = mip_smem;
    } else {
        mip_lds = stage_mip(mip, MIP ? p.mip_words : 0);
    }
    PNR_STAMP(2);
    uint32_t emitted = 0;
    uint32_t base = (uint32_t)(sums & 0xffffffffull), summed_to = blockIdx.x;   // base = sum of counts[0 .. summed_to)
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x == 0) counts_cur[chunk] = 0;   // (the two count arrays alternate: nobody reads this one before the iteration's field launch adds to it)
        int index = index0;
        if (chunk != blockIdx.x) {   // a second chunk for this workgroup (frames of more than gridDim.x * 256 rays): the plain sequence
            unsigned long long pre = 0;
            for (uint32_t j = summed_to + threadIdx.x; j < chunk; j += kRayBlock) pre += (unsigned long long)counts[j];
            base += (uint32_t)block_sum_u64(pre, red);
            summed_to = chunk;
            const uint32_t slot = chunk * kRayBlock + threadIdx.x;
            index = slot < n_prev ? alive_prev[slot] : -1;
        }
        const int keep = index >= 0 ? 1 : 0;
        const unsigned long long alive_mask = __ballot(keep);
        if (lane == 0) csum[wave] = __popcll(alive_mask);
        __syncthreads();
        int woff = 0;
        for (int wv = 0; wv < wave; wv++) woff += csum[wv];
        __syncthreads();
        PNR_STAMP(3);
        // (no early exit for the lanes without a ray: the tail of the wave is marched cooperatively, idle lanes included)
        uint32_t n = 0, step = 0;
        float t = 0.0f, far = -FLT_MAX, last_t = 0.0f;
        RayCtx c = {};
        if (keep) {
            n = base + (uint32_t)woff + (uint32_t)__popcll(alive_mask & ((1ull << lane) - 1ull));
            rays_alive[n] = index;
            ctx_init(c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, p, grid, mip_lds);
            t = rays_t[index];
            const BoxHit bh = clip_to_box(c, fars[index]);
            far = bh.far;
            t = fmaf(clampf(t * c.dt_gamma, c.dt_min, c.dt_max), 0.0f, t);  // perturb == False on the inference path
            last_t = t;
            t = skip_to_box<MIP && POW2>(c, bh, t);
        }
        PNR_STAMP(4);
#ifdef PNR_MARCH_TIMING
        uint32_t my_probes = 0;
#endif
#ifdef PNR_MARCH_STATS
        unsigned long long probes = 0, empties = 0;
        unsigned int kinds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        bool active = keep && t < far;     // n_step >= 1
        [[maybe_unused]] bool queued = false;       // MODE 2: this lane's ray went to the straggler queue
        [[maybe_unused]] uint32_t step_queued = 0;  // ... with this many samples already written here
        for (uint32_t round = 0;; round++) {
            const unsigned long long am = __ballot(active);
            if (am == 0ull) break;
            if constexpr (MODE == 1) { if (p.coop && __popcll(am) <= kCoopRays) break; }
            if constexpr (MODE == 2) {
                // a ray that has spent `budget` probes WITHOUT finding a sample is handed over (a ray that emits a sample per probe -- n_step of them in the
                // last iterations -- is not a straggler): one counter atomic per wave and round in which any lane gives up
                const bool give_up = active && round - step >= budget;
                const unsigned long long gm = __ballot(give_up);
                if (gm != 0ull) {
                    int32_t* qctr = qctr_all + ((prev->iterations + 1) & 1) * kQueueCtrs;
                    const int leader = __ffsll((long long)gm) - 1;
                    int slot0 = 0;
                    if (lane == leader) slot0 = atomicAdd(&qctr[0], __popcll(gm));
                    slot0 = __builtin_amdgcn_readlane(slot0, leader);
                    if (give_up) {
                        StragglerRec r;
                        r.n = (int32_t)n; r.index = index; r.step = (int32_t)step; r.t = t; r.last_t = last_t; r.far = far; r.pad0 = 0; r.pad1 = 0;
                        qrecs[(uint32_t)slot0 + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull))] = r;
                        queued = true; step_queued = step; active = false;
                    }
                    if (gm == am) break;
                }
            }
            if (active) {
                float x, y, z, dt;
#ifdef PNR_MARCH_TIMING
                my_probes++;
#endif
#ifdef PNR_MARCH_STATS
                probes++;
                int kind = 7;
                const bool hit = march_probe<MIP, POW2>(c, t, x, y, z, dt, &kind);
                kinds[kind & 7]++;
                if (!hit) empties++;
#else
                const bool hit = march_probe<MIP, POW2, (MODE != 2 || PNR_MARCH2_JUMPS)>(c, t, x, y, z, dt);
#endif
                if (hit) {
                    const size_t row = (size_t)n * n_step + step;
                    float* px = xyzs + row * 3;
                    float* pd = dirs + row * 3;
                    float* pl = deltas + row * 2;
                    px[0] = x; px[1] = y; px[2] = z;
                    pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                    t += dt;
                    pl[0] = dt; pl[1] = t - last_t;
                    last_t = t;
                    step++;
                }
                active = t < far && step < n_step;
            }
        }
        if constexpr (MODE == 1) {
            if (p.coop && __ballot(active) != 0ull)
                step = march_coop_tail<MIP, POW2>(coop[wave], p, grid, mip_lds, n_step, active, c, t, far, last_t, n, step, xyzs, dirs, deltas);
        }
#ifdef PNR_MARCH_STATS
        if (keep) {
            const int so = prev->iterations + 1 == 0 ? 0 : 4;
            if (atomicMax(&g_march_max[(prev->iterations + 1) & 63], (unsigned int)probes) < (unsigned int)probes)
                for (int kk = 0; kk < 8; kk++) g_march_kinds[(prev->iterations + 1) & 63][kk] = kinds[kk];
            atomicAdd(&g_march_stats[so + 0], probes); atomicAdd(&g_march_stats[so + 1], empties); atomicAdd(&g_march_stats[so + 3], 1ull);
            atomicAdd(&g_march_hist[so ? 1 : 0][probes > 31 ? 31 : (int)probes], 1u);
            for (int kk = 0; kk < 8; kk++) if (kinds[kk]) atomicAdd(&g_march_kinds[so ? 63 : 62][kk], kinds[kk]);   // rows 62 / 63: kind totals of the first / later launches
        }
        {   // wave-level: max probes over the wave (what the wave actually executes)
            unsigned long long mx = probes;
            for (int off = 32; off > 0; off >>= 1) { unsigned long long o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_march_stats[prev->iterations + 1 == 0 ? 2 : 6], mx);
        }
#endif
        if (keep) {
            emitted += step;
            if constexpr (MODE == 2) {   // rows from step_queued on belong to the queue's consumer (samples, then the sentinel): the lookup's other workgroups skip them
                uint8_t* pf = rowflag + (size_t)n * n_step;
                for (uint32_t k = 0; k < n_step; k++) pf[k] = (queued && k >= step_queued) ? 1 : 0;
                if (queued) step = n_step;
            }
            float* pl = deltas + ((size_t)n * n_step + step) * 2;
            for (; step < n_step; step++) { pl[0] = 0.0f; pl[1] = 0.0f; pl += 2; }
        }
        PNR_STAMP(5);
#ifdef PNR_MARCH_TIMING
        if (timing) {   // wave maxima of the probe count and of the dependent global brick loads
            uint32_t mp = my_probes, ml = c.n_loads;
            for (int off = 32; off > 0; off >>= 1) { mp = max(mp, (uint32_t)__shfl_xor((int)mp, off, 64)); ml = max(ml, (uint32_t)__shfl_xor((int)ml, off, 64)); }
            const uint32_t crowd = (uint32_t)__popcll(__ballot(my_probes >= 5u));   // lanes of this wave with a long walk: a crowd marches at full SIMT efficiency, a loner leaves 63 lanes idle
            if ((threadIdx.x & 63) == 0) { g_march_timing[(size_t)twave * 8 + 6] = mp; g_march_timing[(size_t)twave * 8 + 7] = ml | ((unsigned long long)crowd << 16); }
        }
#endif
    }
    // one partial per workgroup, summed by workgroup 0 of the next iteration's march: thousands of same-address atomics would serialise in L2
    __shared__ uint32_t wsum[kRayBlock / PNR_WAVE];
    for (int off = PNR_WAVE / 2; off > 0; off >>= 1) emitted += shfl_xor_fresh(emitted, off);
    if ((threadIdx.x & (PNR_WAVE - 1)) == 0) wsum[threadIdx.x / PNR_WAVE] = emitted;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int wv = 0; wv < (int)(kRayBlock / PNR_WAVE); wv++) tot += wsum[wv];
        emitted_partials[blockIdx.x] = (int32_t)tot;
    }
}

// ------------------------------------------------------------------------------------------
// Hash-grid lookup of the frame loop: gridencoder.cu:75-175 for D = 3, C = 2, fused with GridEncoder.forward's (x + bound) / (2 bound)
// (gridencoder/grid.py:142); rows from the control block; dead slots (delta == 0) are skipped.  One row body per table layout
// (grid_row<KIND>), used by the launch's ordinary workgroups -- level-major: blockIdx.y is the level, its constants are scalars --
// and by the hosted march tail, which looks its own rows up on all levels.
//   GK_SINGLE  up to three separate fp32 tables (encoder, encoder_palette, encoder_clip): blockIdx.z
//   GK_PAIR    PaletteNeRF's `encoder` and `encoder_palette` (looked up at the same positions) interleaved row by row -- (a.x, a.y, b.x, b.y) =
//              16 bytes per index: one gather fetches both tables' rows, the lookup of the pair costs the lane requests and L2->L1 line fills of
//              ONE table.  Same corner order and fmaf chains as two separate lookups: the encoder outputs are bit-identical.
//   GK_TRIPLE  --pred_clip: the three tables in one copy, 32 bytes per row (two 16-byte loads from the same sector per corner)
//   GK_HALF1/2 fp16 tables (the reference's -O / --fp16 mode: `embeddings.to(torch.half)`, gridencoder/grid.py:38): 1 or 2 tables interleaved
//              row by row, a row = NT x half2.  Interpolation with the reference's half accumulator -- every addend and every partial sum
//              rounded to fp16 (gridencoder.cu:142,165 with scalar_t = at::Half) -- so the encoder output equals k_grid_fwd<__half>'s bit for
//              bit; it is handed to the field kernel as fp32 (exact).
// ------------------------------------------------------------------------------------------
enum GridKind { GK_SINGLE = 0, GK_PAIR = 1, GK_TRIPLE = 2, GK_HALF1 = 3, GK_HALF2 = 4 };
struct GridArgs {
    const float* xyzs; const float* deltas;
    const void* table[3]; float* enc[3];
    const int32_t* offsets; LevelParams lp;
    uint32_t level_stride; float bound, two_bound, inv_two_bound /* exact reciprocal when 2 * bound is a power of two, else 0 */; uint32_t gridtype;
};
// kind: how a row index is formed on this level (gridencoder.cu:49-72 evaluated once per level instead of once per corner) --
//   0 the reference's general form (stride test per dimension, hash or tiled, `%`); 1 dense: side^3 <= size, the index is below the size and the
//   `%` is the identity; 2 hashed with a power-of-two size: a mask.  All three give the same index (tests: every table layout against k_grid_fwd).
struct LevelCtx { uint32_t off0, hashmap_size, resolution; float scale; uint32_t kind; };
__device__ __forceinline__ LevelCtx level_ctx(const GridArgs& g, uint32_t level) {
    LevelCtx lc;
    lc.off0 = (uint32_t)g.offsets[level];
    lc.hashmap_size = (uint32_t)g.offsets[level + 1] - lc.off0;
    lc.scale = g.lp.scale[level];
    lc.resolution = g.lp.resolution[level];
    lc.kind = level_kind(g.gridtype, lc.hashmap_size, lc.resolution);
    return lc;
}
// the eight corners of row b's cell on one level: row indices (x CMUL) and the fractional position the weights come from; false = the point is outside [0, 1]^3
template <uint32_t CMUL>
__device__ __forceinline__ bool grid_corner_rows(const GridArgs& g, const LevelCtx& lc, uint32_t b, uint32_t idxs[8], float pos[3]) {
    float in[3];
    bool oob = false;
    // the row's three coordinates requested together (one 12-byte load): left alone the compiler sinks each 4-byte load to its use and waits for it there -- three
    // memory round trips in a row in front of the eight gathers.  One empty asm statement that names all three pins the point where they must have arrived.
    // (Frame times did not move -- the launch has seven waves per SIMD to cover the trips -- but the hosted tail's spills did: k_frame_grid 44 -> 0 B of scratch
    // per lane, the pair kernel 84 -> 56.)
    float xyz[3];
#pragma unroll
    for (int d = 0; d < 3; d++) xyz[d] = g.xyzs[(size_t)b * 3 + d];
    asm volatile("" : "+v"(xyz[0]), "+v"(xyz[1]), "+v"(xyz[2]));
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // GridEncoder.forward's (x + bound) / (2 bound) (gridencoder/grid.py:142).  With 2 * bound a power of two -- every shipped scene -- the division
        // is an exact scaling and so is the multiplication by the exact reciprocal: same bits, one v_mul instead of the ~10-instruction IEEE
        // division (three per (sample, level): an eighth of this kernel's vector instructions).  Any other bound divides.
        const float sft = xyz[d] + g.bound;
        in[d] = g.inv_two_bound != 0.0f ? sft * g.inv_two_bound : sft / g.two_bound;
        oob |= (in[d] < 0.0f) | (in[d] > 1.0f);
    }
    if (oob) return false;
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = fmaf(in[d], lc.scale, 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    corner_rows_by_kind<CMUL>(lc.kind, g.gridtype, lc.hashmap_size, lc.resolution, pg, idxs);   // (kind is wave-uniform in the level-major workgroups; 0 in the hosted tail)
    return true;
}
// trilinear weights in the reference's order of multiplications (gridencoder.cu:150-163)
__device__ __forceinline__ void grid_corner_weights(const float pos[3], float ws[8]) {
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) w *= (idx & (1u << d)) ? pos[d] : 1.0f - pos[d];
        ws[idx] = w;
    }
}
// The eight gathers of a row go out first, back to back, and the weights are formed while they are in flight: every address exists before
// the first load (one empty asm statement with all of them as operands pins that point) and nothing is scheduled across the end of the
// group.  Left to itself the compiler does this too -- until it is asked to keep the kernel within a register budget
// (amdgpu_waves_per_eu, which the hosted tail needs): then it forms the weights first and threads the loads between the address
// arithmetic, and the lego launch takes 76 instead of 66 us with the very same instructions (profiles/scratch/prof_ref.sh).
// ORDERED = false (the hosted tail's own few rows; GK_TRIPLE, whose 16 addresses + 48 values do not fit the budget): the compiler's order.
template <typename T, uint32_t MUL, bool ORDERED>
__device__ __forceinline__ void gather8(const T* tab, const uint32_t (&idxs)[8], T (&v)[8]) {
    const T* p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = tab + (size_t)idxs[i] * MUL;
    if constexpr (!ORDERED) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = *p[i];
        return;
    }
    load8_fresh(p, v);
}
// (table / enc0: the table and first output of this call -- GK_SINGLE picks them per blockIdx.z; resolved by the caller, outside its row loop)
template <int KIND, bool ORDERED = true>
__device__ __forceinline__ void grid_row(const GridArgs& g, const LevelCtx& lc, uint32_t level, const void* __restrict__ table, float* __restrict__ enc0, uint32_t b) {
    uint32_t idxs[8];
    float pos[3], ws[8];
    const size_t o = ((size_t)level * g.level_stride + b) * 2;
    if constexpr (KIND == GK_SINGLE) {
        float acc[2] = {0.0f, 0.0f};
        if (grid_corner_rows<1>(g, lc, b, idxs, pos)) {
            f32x2 v[8];
            gather8<f32x2, 1, ORDERED>(static_cast<const f32x2*>(table) + lc.off0, idxs, v);
            grid_corner_weights(pos, ws);
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) { acc[0] = fmaf(ws[idx], v[idx].x, acc[0]); acc[1] = fmaf(ws[idx], v[idx].y, acc[1]); }
        }
        *reinterpret_cast<float2*>(enc0 + o) = make_float2(acc[0], acc[1]);
    } else if constexpr (KIND == GK_PAIR) {
        float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (grid_corner_rows<1>(g, lc, b, idxs, pos)) {
            f32x4 v[8];
            gather8<f32x4, 1, ORDERED>(static_cast<const f32x4*>(table) + lc.off0, idxs, v);
            grid_corner_weights(pos, ws);
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                out.x = fmaf(ws[idx], v[idx].x, out.x); out.y = fmaf(ws[idx], v[idx].y, out.y);
                out.z = fmaf(ws[idx], v[idx].z, out.z); out.w = fmaf(ws[idx], v[idx].w, out.w);
            }
        }
        *reinterpret_cast<float2*>(enc0 + o) = make_float2(out.x, out.y);
        *reinterpret_cast<float2*>(g.enc[1] + o) = make_float2(out.z, out.w);
    } else if constexpr (KIND == GK_TRIPLE) {
        float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float2 outc = make_float2(0.0f, 0.0f);
        if (grid_corner_rows<1>(g, lc, b, idxs, pos)) {
            f32x4 v[8];
            f32x2 vc[8];
            const f32x4* tab = static_cast<const f32x4*>(table) + (size_t)lc.off0 * 2;
            gather8<f32x4, 2, false>(tab, idxs, v);
            gather8<f32x2, 4, false>(reinterpret_cast<const f32x2*>(tab + 1), idxs, vc);
            grid_corner_weights(pos, ws);
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                out.x = fmaf(ws[idx], v[idx].x, out.x); out.y = fmaf(ws[idx], v[idx].y, out.y);
                out.z = fmaf(ws[idx], v[idx].z, out.z); out.w = fmaf(ws[idx], v[idx].w, out.w);
                outc.x = fmaf(ws[idx], vc[idx].x, outc.x); outc.y = fmaf(ws[idx], vc[idx].y, outc.y);
            }
        }
        *reinterpret_cast<float2*>(enc0 + o) = make_float2(out.x, out.y);
        *reinterpret_cast<float2*>(g.enc[1] + o) = make_float2(out.z, out.w);
        *reinterpret_cast<float2*>(g.enc[2] + o) = outc;
    } else {
        constexpr int NT = KIND == GK_HALF2 ? 2 : 1;
        typedef uint32_t RowT __attribute__((ext_vector_type(NT)));   // NT x half2
        __half acc[2 * NT];
#pragma unroll
        for (int ch = 0; ch < 2 * NT; ch++) acc[ch] = __float2half(0.0f);
        if (grid_corner_rows<1>(g, lc, b, idxs, pos)) {
            RowT v[8];
            gather8<RowT, 1, ORDERED>(static_cast<const RowT*>(table) + lc.off0, idxs, v);
            grid_corner_weights(pos, ws);
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                __half hv[2 * NT];
                __builtin_memcpy(hv, &v[idx], sizeof(RowT));
#pragma unroll
                for (int ch = 0; ch < 2 * NT; ch++)   // the reference's half accumulator: addend and sum rounded to fp16 (corner_accumulate<__half>)
                    acc[ch] = __float2half(__half2float(acc[ch]) + __half2float(__float2half(ws[idx] * __half2float(hv[ch]))));
            }
        }
        *reinterpret_cast<float2*>(enc0 + o) = make_float2(__half2float(acc[0]), __half2float(acc[1]));
        if constexpr (NT == 2) *reinterpret_cast<float2*>(g.enc[1] + o) = make_float2(__half2float(acc[2]), __half2float(acc[3]));
    }
}

#ifdef PNR_HOSTED_TIMING
// instrumented builds only: wall-clock stamps (100 MHz) of ONE iteration's lookup launch.  [0..7]: launch-wide (earliest start, latest end of an
// ordinary workgroup, latest end of a hosted one, rays queued); then 8 per hosted workgroup: start, mip staged, march done, lookups done, probes of its slowest lane
__device__ unsigned long long g_hosted_timing[8 + 8 * 256 + 2 * 16 * 128];   // ... then (start, end) of every 16th ordinary workgroup of the first 2048 per level
__device__ int g_hosted_timing_iter = 3;
#endif
#ifndef PNR_HOSTED_MIN_GLOG
#define PNR_HOSTED_MIN_GLOG 3u
#endif
#ifndef PNR_HOSTED_JUMPS
#define PNR_HOSTED_JUMPS false
#endif
// The queue's consumer: rays_per_wave rays to a wave (8 when the queue is short -- the lookups that follow then spread over more waves --
// up to 64), the march loop of k_frame_march from the recorded state, then the new rows' lookups, (row, level) pairs dealt to the lanes.
template <int KIND>
__device__ __forceinline__ void hosted_march_tail(const FrameCtl* ctl, const GridArgs& g, const HostedArgs& ha) {
    struct { uint32_t blocks; const int32_t* qctr; const StragglerRec* qrecs; const float* rays_o; const float* rays_d; const uint8_t* bitfield;
             const uint32_t* mip; MarchParams p; float* xyzs; float* dirs; float* deltas; int32_t* partials; } h;
    {
        const HostedConst* hc = ha.hc;
        const int par = ctl->iterations & 1;
        h.blocks = ha.blocks; h.qctr = hc->qctr_all + par * kQueueCtrs; h.qrecs = hc->qrecs; h.rays_o = hc->rays_o; h.rays_d = hc->rays_d;
        h.bitfield = hc->bitfield; h.mip = hc->mip; h.p = hc->p; h.xyzs = hc->xyzs; h.dirs = hc->dirs; h.deltas = hc->deltas;
        h.partials = hc->partials[par] + ha.partial_base;
    }
    // These waves carry a dependent chain and share their SIMD with up to seven lookup waves that have plenty of independent work: they go first.
    __builtin_amdgcn_s_setprio(3);
    __shared__ float lv_scale[16];
    __shared__ uint32_t lv_res[16], lv_off[16], lv_size[16];
    __shared__ uint32_t wrows[kRayBlock / PNR_WAVE][PNR_WAVE];
    __shared__ uint32_t wemit[kRayBlock / PNR_WAVE];
    const uint32_t count = (uint32_t)h.qctr[0];
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    uint32_t glog = PNR_HOSTED_MIN_GLOG;   // rays per wave: as few as the hosted waves (kHostedBlocks * 4) allow -- a wave marches at the pace of its slowest ray, and its lookups spread over the idle lanes
    while (glog < 6u && (count >> glog) > kHostedBlocks * (kRayBlock / PNR_WAVE)) glog++;
    const uint32_t G = 1u << glog;
    const uint32_t ntasks = (count + G - 1) >> glog;
    // tasks are dealt four to a workgroup (one per wave): a workgroup with a task has no idle wave sitting on a wave slot through the whole march,
    // and the workgroups beyond the queue leave at once (one wave per workgroup, the first form: 576 idle resident waves on a typical later launch)
    constexpr uint32_t kWavesPerBlock = kRayBlock / PNR_WAVE;
    if (blockIdx.x * kWavesPerBlock >= ntasks) { if (threadIdx.x == 0) h.partials[blockIdx.x] = 0; return; }   // block-uniform (also: count == 0)
#ifdef PNR_HOSTED_TIMING
    const bool timing = ctl->iterations == g_hosted_timing_iter;
    unsigned long long* tm = g_hosted_timing + 8 + 8 * blockIdx.x;
    uint32_t my_probes = 0;
    if (timing && threadIdx.x == 0) { tm[0] = wall_clock64(); g_hosted_timing[3] = count; }
#endif
    const uint32_t* mip_lds = stage_mip(h.mip, h.p.mip_words);
#ifdef PNR_HOSTED_TIMING
    if (timing && threadIdx.x == 0) tm[1] = wall_clock64();
#endif
    if (threadIdx.x < 16) {
        const LevelCtx lc = level_ctx(g, threadIdx.x);
        lv_scale[threadIdx.x] = lc.scale; lv_res[threadIdx.x] = lc.resolution; lv_off[threadIdx.x] = lc.off0; lv_size[threadIdx.x] = lc.hashmap_size;
    }
    __syncthreads();
    const uint32_t n_step = (uint32_t)ctl->n_step;
    const uint32_t n_tab = KIND == GK_SINGLE ? ha.n_tab : 1u;
    uint32_t emitted = 0;
    for (uint32_t task = blockIdx.x * kWavesPerBlock + (uint32_t)wave; task < ntasks; task += h.blocks * kWavesPerBlock) {
        const uint32_t qi = (task << glog) + (uint32_t)lane;
        const bool have = (uint32_t)lane < G && qi < count;
        StragglerRec r = {};
        if (have) r = h.qrecs[qi];
        const uint32_t row0 = (uint32_t)r.n * n_step;
        uint32_t step = (uint32_t)r.step;
        if (have) {
            RayCtx c;
            ctx_init(c, h.rays_o + (size_t)r.index * 3, h.rays_d + (size_t)r.index * 3, h.p, h.bitfield, mip_lds);
            float t = r.t, last_t = r.last_t;
            bool active = t < r.far && step < n_step;
            while (active) {
                float x, y, z, dt;
#ifdef PNR_HOSTED_TIMING
                my_probes++;
#endif
                if (march_probe<true, true, PNR_HOSTED_JUMPS>(c, t, x, y, z, dt)) {
                    const size_t row = (size_t)row0 + step;
                    float* px = h.xyzs + row * 3;
                    float* pd = h.dirs + row * 3;
                    float* pl = h.deltas + row * 2;
                    px[0] = x; px[1] = y; px[2] = z;
                    pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                    t += dt;
                    pl[0] = dt; pl[1] = t - last_t;
                    last_t = t;
                    step++;
                }
                active = t < r.far && step < n_step;
            }
            emitted += step - (uint32_t)r.step;
            float* pl = h.deltas + ((size_t)row0 + step) * 2;
            for (uint32_t k = step; k < n_step; k++) { pl[0] = 0.0f; pl[1] = 0.0f; pl += 2; }   // the unfilled slots' sentinel
        }
        // lookups of the rows just written (this wave's own stores: program order + the fence make them visible to its loads)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#ifdef PNR_HOSTED_TIMING
        if (timing) {
            uint32_t mp = my_probes;
            for (int off = 32; off > 0; off >>= 1) mp = max(mp, (uint32_t)__shfl_xor((int)mp, off, 64));
            if (lane == 0) { atomicMax(&tm[2], wall_clock64()); atomicMax(&tm[4], (unsigned long long)mp); }
        }
#endif
        const uint32_t nnew = have ? step - (uint32_t)r.step : 0u;
        for (uint32_t k = 0; k < n_step; k++) {   // the k-th new row of every ray that has one (n_step is 1 in most iterations)
            const bool has_row = k < nnew;
            const unsigned long long m = __ballot(has_row);
            if (m == 0ull) break;
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (has_row) wrows[wave][__popcll(m & ((1ull << lane) - 1ull))] = row0 + (uint32_t)r.step + k;
            wave_lds_sync();
            for (uint32_t pi = (uint32_t)lane; pi < cnt * 16u; pi += PNR_WAVE) {
                const uint32_t b = wrows[wave][pi >> 4], level = pi & 15u;
                LevelCtx lc;
                lc.scale = lv_scale[level]; lc.resolution = lv_res[level]; lc.off0 = lv_off[level]; lc.hashmap_size = lv_size[level];
                lc.kind = 0u;   // (levels differ per lane here: the general form)
                for (uint32_t tz = 0; tz < n_tab; tz++) grid_row<KIND, false>(g, lc, level, g.table[tz], g.enc[tz], b);
            }
            wave_lds_sync();
        }
    }
    for (int off = PNR_WAVE / 2; off > 0; off >>= 1) emitted += __shfl_xor(emitted, off, PNR_WAVE);
    if (lane == 0) wemit[wave] = emitted;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int wv = 0; wv < (int)(kRayBlock / PNR_WAVE); wv++) tot += wemit[wv];
        h.partials[blockIdx.x] = (int32_t)tot;
#ifdef PNR_HOSTED_TIMING
        if (timing) tm[3] = wall_clock64();
#endif
    }
}

template <int KIND>
__device__ __forceinline__ void frame_grid_body(const FrameCtl* __restrict__ ctl, const GridArgs& g, const HostedArgs& h) {
    if (ctl->done) return;
    // GK_SINGLE with PNR_FRAME_LEVEL_PAIRS: a workgroup takes two levels, the y-th finest and the y-th coarsest, for each of its rows (8 instead of 16
    // workgroup rows per table): the row's position is read once, a scattered and a dense level share a thread
    constexpr uint32_t kLv = KIND == GK_SINGLE ? (PNR_FRAME_LEVEL_PAIRS == 2 ? 4u : (PNR_FRAME_LEVEL_PAIRS ? 2u : 1u))
                                               : ((KIND == GK_PAIR && PNR_FRAME_LEVEL_PAIRS_PAL) ? 2u : 1u);   // levels per workgroup row
    constexpr bool kPairs = kLv > 1;
    constexpr uint32_t kNY = 16u / kLv;
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, nbx = gridDim.x;
    if (h.blocks) {   // launch-uniform: a one-dimensional launch, the hosted workgroups first
        if (bx < h.blocks) {
#ifndef PNR_NO_HOSTED_CODE
            hosted_march_tail<KIND>(ctl, g, h);
#endif
            return;
        }
        const uint32_t id = bx - h.blocks, yz = id / h.gx;
        bx = id - yz * h.gx; by = yz % kNY; bz = yz / kNY; nbx = h.gx;
    }
    const uint32_t stride = nbx * 256u;
    const uint32_t rows = (uint32_t)ctl->n_alive * (uint32_t)ctl->n_step;
    // the finest levels -- scattered rows, 5x the time of a dense level -- are dispatched first and the dense ones fill the launch's tail
    // (ascending order: lego 3.80 ms / lookup at 0.66 of the roofline, descending: 3.79 / 0.675; garden 13.45 -> 13.2 ms)
    const uint32_t level = 15u - by;
#ifdef PNR_HOSTED_TIMING
    const bool timing = ctl->iterations == g_hosted_timing_iter && (bx & 15u) == 0 && bx < 2048u && bz == 0;
    unsigned long long* tmm = g_hosted_timing + 8 + 8 * 256 + 2 * (level * 128 + (bx >> 4));
    if (timing && threadIdx.x == 0) tmm[0] = wall_clock64();
#endif
    const LevelCtx lc = level_ctx(g, level);
    [[maybe_unused]] const LevelCtx lc2 = level_ctx(g, kPairs ? by : level);
    [[maybe_unused]] const LevelCtx lc3 = level_ctx(g, kLv == 4 ? 11u - by : level), lc4 = level_ctx(g, kLv == 4 ? 4u + by : level);
    const void* table = g.table[KIND == GK_SINGLE ? bz : 0];
    float* enc0 = g.enc[KIND == GK_SINGLE ? bz : 0];
    for (uint32_t b = bx * 256u + threadIdx.x; b < rows; b += stride) {
        const float d0 = g.deltas[(size_t)b * 2];
        const uint32_t fl = h.rowflag ? h.rowflag[b] : 0u;
        if (d0 == 0.0f || fl) continue;
        grid_row<KIND>(g, lc, level, table, enc0, b);
        if constexpr (kLv == 4) { grid_row<KIND>(g, lc3, 11u - by, table, enc0, b); grid_row<KIND>(g, lc4, 4u + by, table, enc0, b); }
        if constexpr (kPairs) grid_row<KIND>(g, lc2, by, table, enc0, b);
    }
#ifdef PNR_HOSTED_TIMING
    if (timing && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tmm[1] = wall_clock64(); }
#endif
}

#ifndef PNR_GRID_WAVES
#define PNR_GRID_WAVES 8
#endif
#ifndef PNR_GRID_WAVES_SINGLE
#define PNR_GRID_WAVES_SINGLE 7   // the one-table kernel with its hosted tail: 72 registers (fewer spills in the tail) beat the eighth wave -- lego 3.80 -> 3.68 ms; the pair kernel is better off with eight (garden 13.3 against 13.6)
#endif
#define PNR_GRID_KERNEL(NAME, KIND, WAVES)                                                                                                   \
    __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) NAME(const FrameCtl* __restrict__ ctl, GridArgs g, HostedArgs h) { \
        frame_grid_body<KIND>(ctl, g, h);                                                                                                    \
    }
PNR_GRID_KERNEL(k_frame_grid, GK_SINGLE, PNR_GRID_WAVES_SINGLE)
PNR_GRID_KERNEL(k_frame_grid_pair, GK_PAIR, PNR_GRID_WAVES)
PNR_GRID_KERNEL(k_frame_grid_triple, GK_TRIPLE, PNR_GRID_WAVES)
PNR_GRID_KERNEL(k_frame_grid_h1, GK_HALF1, PNR_GRID_WAVES)
PNR_GRID_KERNEL(k_frame_grid_h2, GK_HALF2, PNR_GRID_WAVES)
#undef PNR_GRID_KERNEL

__global__ void __launch_bounds__(256) k_interleave_tables(const float2* __restrict__ a, const float2* __restrict__ b, uint64_t rows,
                                                           float4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) { const float2 u = a[i], v = b[i]; out[i] = make_float4(u.x, u.y, v.x, v.y); }
}
__global__ void __launch_bounds__(256) k_interleave_tables3(const float2* __restrict__ a, const float2* __restrict__ b, const float2* __restrict__ c,
                                                            uint64_t rows, float4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) {
        const float2 u = a[i], v = b[i], t = c[i];
        out[2 * i] = make_float4(u.x, u.y, v.x, v.y);
        out[2 * i + 1] = make_float4(t.x, t.y, 0.0f, 0.0f);
    }
}

// the fused MFMA field of field.hip with rows from the control block, dead-slot skipping and density_scale
constexpr int kFieldThreads = 512;
// The compositing step of the iteration is done right here by the lanes that hold the samples' sigma and rgb: same operations in the same
// order as k_frame_composite's phase 2 on the same values (what it would re-read from sigmas / rgbs), including the per-chunk survivor
// counts of the compaction.  fuse_mode 1: iterations with one sample per ray only (27 of the 29 of the benchmark frame; a 256-sample tile
// IS chunk `tile` of the alive list, one plain store per tile) -- k_frame_composite returns at once on such iterations; fuse_mode 2
// (default): every iteration, and the frame loop has no composite launch.  With n_step samples per ray a wave tile holds
// floor(32 / n_step) whole rays (30 or 28 of its 32 rows for n_step = 3, 5, 6, 7); the lane of a ray's first row walks the ray's rows
// through wave shuffles; survivors are added to their chunk's count with at most two atomics per wave (the count arrays alternate between
// iterations and the march launch clears the one its iteration fills).
// CHECK (split-fp16 only): watch the split operands for magnitudes beyond fp16's range and raise scratch[1] (SplitWatch, field_core.hpp)
template <int PREC, bool CHECK>
__global__ void __launch_bounds__(kFieldThreads) k_frame_field(const FrameCtl* __restrict__ ctl, const float* __restrict__ enc, uint32_t level_stride,
                                                               const float* __restrict__ dirs, const float* __restrict__ deltas,
                                                               const float* __restrict__ packed, float density_scale, float enc_scale, float* __restrict__ sigmas,
                                                               float* __restrict__ rgbs, int fuse_mode, float T_thresh, int32_t* __restrict__ rays_alive,
                                                               float* __restrict__ rays_t, float* __restrict__ weights_sum, float* __restrict__ depth,
                                                               float* __restrict__ image, int32_t* __restrict__ scratch, int32_t* __restrict__ counts_cur) {
    if (ctl->done) return;
    const uint32_t n_step = (uint32_t)ctl->n_step;
    const uint32_t B = (uint32_t)ctl->n_alive * n_step;
    const bool fuse = fuse_mode && n_step == 1;
    const bool fuse_rays = fuse_mode == 2 && n_step > 1;
    const uint32_t rpw = fuse_rays ? (32u / n_step) * n_step : 32u;   // rows of a wave tile: whole rays
    const uint32_t rpt = rpw * (kFieldThreads / PNR_WAVE);
    const uint32_t ntiles = (B + rpt - 1) / rpt;
    if (blockIdx.x >= ntiles) return;
    __shared__ float w[kPackedFloats];
    __shared__ int wsum[kFieldThreads / PNR_WAVE];
    for (int i = threadIdx.x * 4; i < kPackedFloats; i += kFieldThreads * 4) lds_copy16(&packed[i], &w[i]);
    bool weights_ready = false;   // the wait for the weights sits behind the first tile's own loads (every wave passes it exactly once: here or after the loop)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const uint32_t l = (uint32_t)lane & 31u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t n = tile * rpt + wave * rpw + l;
        const bool mine = l < rpw && n < B;
        const float dl0 = mine ? deltas[(size_t)n * 2] : 0.0f;
        const bool valid = mine && dl0 != 0.0f;
        FieldOut o = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t nc = n < B ? n : (B - 1);
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if (valid) { dx = dirs[(size_t)nc * 3]; dy = dirs[(size_t)nc * 3 + 1]; dz = dirs[(size_t)nc * 3 + 2]; }
        if (!weights_ready) { lds_copy_wait(); __syncthreads(); weights_ready = true; }   // (tile loop trip counts are block-uniform: all waves of a block are here together)
        if (__any(valid)) {   // wave-uniform: otherwise all 32 slots of this wave are dead or out of range
            SplitWatch<CHECK> sw;
            o = nerf_field_tile<PREC, CHECK>(w, lane, valid, enc, level_stride, nc, dx, dy, dz, enc_scale, sw);
            if constexpr (CHECK) { if (sw.overflowed()) scratch[1] = 1; }
        }
        float sigma = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (valid && h == 0) {
            sigma = density_scale * __expf(o.sigma_logit);   // nerf/renderer.py:372; hardware exp / rcp: ~1e-7 on these arguments
            cr = __frcp_rn(1.0f + __expf(-o.o0));
            cg = __frcp_rn(1.0f + __expf(-o.o1));
            cb = __frcp_rn(1.0f + __expf(-o.o2));
            if (!fuse && !fuse_rays) {
                sigmas[n] = sigma;
                rgbs[(size_t)n * 3] = cr; rgbs[(size_t)n * 3 + 1] = cg; rgbs[(size_t)n * 3 + 2] = cb;
            }
        }
        if (fuse) {   // block-uniform
            int keep = 0;
            if (h == 0 && n < B) {   // slot n of the alive list: k_frame_composite phase 2 with n_step == 1
                const int index = rays_alive[n];
                float ws = weights_sum[index], t = rays_t[index], d = depth[index];
                float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
                bool stepped = false;
                if (dl0 != 0.0f) {
                    const float alpha = 1.0f - __expf(-sigma * dl0);
                    const float T = 1.0f - ws;
                    const float wgt = alpha * T;
                    ws += wgt;
                    t += deltas[(size_t)n * 2 + 1];
                    d = fmaf(wgt, t, d);
                    r = fmaf(wgt, cr, r); g = fmaf(wgt, cg, g); b = fmaf(wgt, cb, b);
                    stepped = !(T < T_thresh);
                }
                if (!stepped) rays_alive[n] = -1; else { rays_t[index] = t; keep = 1; }
                weights_sum[index] = ws; depth[index] = d;
                image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wsum[wave] = __popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
                for (int wv = 0; wv < kFieldThreads / (int)PNR_WAVE; wv++) tot += wsum[wv];
                counts_cur[tile] = tot;
            }
            __syncthreads();
        } else if (fuse_rays) {   // block-uniform: k_frame_composite phase 2 for rays of n_step rows, all inside this wave's lower half
            const float dl1 = (valid && h == 0) ? deltas[(size_t)n * 2 + 1] : 0.0f;
            const bool leader = h == 0 && mine && (l % n_step) == 0;
            const uint32_t slot = n / n_step;
            int index = 0;
            float ws = 0.0f, t = 0.0f, d = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
            if (leader) {
                index = rays_alive[slot];
                ws = weights_sum[index]; t = rays_t[index]; d = depth[index];
                r = image[index * 3]; g = image[index * 3 + 1]; b = image[index * 3 + 2];
            }
            uint32_t step = 0;
            bool running = leader;
            for (uint32_t k = 0; k < n_step; k++) {   // wave-uniform: row k of every ray in lock step
                const int src = lane + (int)k;
                const float s_k = __shfl(sigma, src), a_k = __shfl(dl0, src), b_k = __shfl(dl1, src);
                const float r_k = __shfl(cr, src), g_k = __shfl(cg, src), c_k = __shfl(cb, src);
                if (running) {
                    if (a_k == 0.0f) running = false;
                    else {
                        const float alpha = 1.0f - __expf(-s_k * a_k);
                        const float T = 1.0f - ws;
                        const float wgt = alpha * T;
                        ws += wgt;
                        t += b_k;
                        d = fmaf(wgt, t, d);
                        r = fmaf(wgt, r_k, r); g = fmaf(wgt, g_k, g); b = fmaf(wgt, c_k, b);
                        if (T < T_thresh) running = false; else step++;
                    }
                }
            }
            int keep = 0;
            if (leader) {
                if (step < n_step) rays_alive[slot] = -1; else { rays_t[index] = t; keep = 1; }
                weights_sum[index] = ws; depth[index] = d;
                image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
            }
            const unsigned long long km = __ballot(keep);
            if (km != 0ull) {   // the wave's rays are consecutive slots: at most two chunks
                const uint32_t c0 = ((tile * rpt + wave * rpw) / n_step) >> 8;
                const unsigned long long k0 = __ballot(keep && (slot >> 8) == c0);
                if (lane == 0) {
                    if (k0) atomicAdd(&counts_cur[c0], __popcll(k0));
                    if (km & ~k0) atomicAdd(&counts_cur[c0 + 1], __popcll(km & ~k0));
                }
            }
        }
    }
}

// reference raymarching.cu:1025-1111 (+ :1114-1185 for the packed aux row of the palette model: every channel
// is composited with the SAME weights, starting from the weights_sum of BEFORE this iteration, exactly as the
// reference's composite_rays_flex calls that precede composite_rays) + the per-chunk survivor count of the compaction
__global__ void __launch_bounds__(kRayBlock) k_frame_composite(const FrameCtl* __restrict__ ctl, float T_thresh, int32_t* __restrict__ rays_alive,
                                                               float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                               const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                               float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image,
                                                               int32_t* __restrict__ counts_cur, const float* __restrict__ aux, float* __restrict__ aux_map,
                                                               uint32_t aux_stride, int aux_done_when_one_step, int all_done_when_one_step) {
    if (ctl->done) return;
    const uint32_t n_alive = (uint32_t)ctl->n_alive, n_step = (uint32_t)ctl->n_step;
    if (all_done_when_one_step && n_step == 1) return;   // k_frame_field has composited this iteration and counted the survivors
    const uint32_t nchunks = (n_alive + kRayBlock - 1) / kRayBlock;
    __shared__ int wsum[kRayBlock / PNR_WAVE];
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (aux && !(aux_done_when_one_step && n_step <= 8 && (32 % n_step) == 0)) {   // with 1, 2, 4 or 8 samples per ray the palette field kernel has composited the rows itself
            // Phase 1 (palette): the packed aux row.  aux_stride / 4 lanes per ray, one float4 of channels per lane: a ray's row is one
            // coalesced 208-byte read per sample and per map instead of a 208-byte-strided walk by a single thread.  Every
            // lane re-derives the (cheap) weights; they start from the weights_sum of BEFORE this iteration because phase 2,
            // which updates it, runs after the barrier.
            // nq lanes per ray (one float4 of channels each), kRayBlock / nq rays per pass
            const uint32_t nq = aux_stride / 4, rays_per_pass = kRayBlock / nq;
            const uint32_t slot = threadIdx.x / nq, q = threadIdx.x - slot * nq;
            for (uint32_t r = slot; r < kRayBlock && slot < rays_per_pass; r += rays_per_pass) {
                const uint32_t n = chunk * kRayBlock + r;
                if (n >= n_alive) continue;
                const int index = rays_alive[n];
                const float* s = sigmas + (size_t)n * n_step;
                const float* in = aux + (size_t)n * n_step * aux_stride + q * 4;
                const float* dl = deltas + (size_t)n * n_step * 2;
                float4* out = reinterpret_cast<float4*>(aux_map + (size_t)index * aux_stride) + q;
                float4 acc = *out;
                float ws = weights_sum[index];
                for (uint32_t step = 0; step < n_step; step++) {
                    if (dl[0] == 0) break;
                    const float alpha = 1.0f - __expf(-s[0] * dl[0]);
                    const float T = 1.0f - ws;
                    const float wgt = alpha * T;
                    ws += wgt;
                    const float4 v = *reinterpret_cast<const float4*>(in);
                    acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
                    if (T < T_thresh) break;
                    s++; in += aux_stride; dl += 2;
                }
                *out = acc;
            }
            __syncthreads();
        }
        const uint32_t n = chunk * kRayBlock + threadIdx.x;
        int keep = 0;
        if (n < n_alive) {
            const int index = rays_alive[n];
            const float ws0 = weights_sum[index];
            const float* s = sigmas + (size_t)n * n_step;
            const float* c = rgbs + (size_t)n * n_step * 3;
            const float* dl = deltas + (size_t)n * n_step * 2;
            float t = rays_t[index], ws = ws0, d = depth[index];
            float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
            uint32_t step = 0;
            while (step < n_step) {
                if (dl[0] == 0) break;
                const float alpha = 1.0f - __expf(-s[0] * dl[0]);
                const float T = 1.0f - ws;
                const float wgt = alpha * T;
                ws += wgt;
                t += dl[1];
                d = fmaf(wgt, t, d);
                r = fmaf(wgt, c[0], r); g = fmaf(wgt, c[1], g); b = fmaf(wgt, c[2], b);
                if (T < T_thresh) break;
                s++; c += 3; dl += 2; step++;
            }
            if (step < n_step) rays_alive[n] = -1; else { rays_t[index] = t; keep = 1; }
            weights_sum[index] = ws; depth[index] = d;
            image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
        }
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & (PNR_WAVE - 1)) == 0) wsum[threadIdx.x / PNR_WAVE] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int wv = 0; wv < (int)(kRayBlock / PNR_WAVE); wv++) tot += wsum[wv];
            counts_cur[chunk] = tot;
        }
        __syncthreads();
    }
}

static inline uint64_t align256(uint64_t v) { return (v + 255) & ~uint64_t(255); }

struct FrameWorkspace {
    FrameCtl* ctl;
    int32_t* alive[2];
    float *rays_t, *xyzs, *dirs, *deltas, *enc, *sigmas, *rgbs;
    float *enc_pal, *enc_clip, *aux;  // palette model only
    void* edit;                       // palette model only: device image of the edit parameters
    float *s_o, *s_d, *s_near, *s_far, *s_ws, *s_depth, *s_image, *s_aux;  // ray_order: inputs / outputs in processing order
    int32_t* scratch;
    int32_t* partials[2];   // the march's per-workgroup sample counts (+ the hosted tail's): written by iteration i, summed by iteration i + 1
    HostedConst* hosted;    // hosted march tail: frame constants (written by k_frame_begin)
    int32_t* qctr;          // straggler queue: two counter sets ...
    StragglerRec* qrecs;    // ... its records (one iteration's worth: a march launch fills it, the lookup launch that follows empties it) ...
    uint8_t* rowflag;       // ... and the per-row "belongs to a queued ray" flags
    uint64_t bytes;
};
static FrameWorkspace carve(void* base, uint32_t N, uint32_t aux_stride = 0, bool with_clip = false) {
    FrameWorkspace w;
    uint64_t off = 0;
    auto take = [&](uint64_t nbytes) { char* p = base ? static_cast<char*>(base) + off : nullptr; off += align256(nbytes); return p; };
    const uint64_t n = N ? N : 1;
    w.ctl = reinterpret_cast<FrameCtl*>(take(2 * sizeof(FrameCtl)));
    w.alive[0] = reinterpret_cast<int32_t*>(take(n * 4));
    w.alive[1] = reinterpret_cast<int32_t*>(take(n * 4));
    w.rays_t = reinterpret_cast<float*>(take(n * 4));
    w.xyzs = reinterpret_cast<float*>(take(n * 12));
    w.dirs = reinterpret_cast<float*>(take(n * 12));
    w.deltas = reinterpret_cast<float*>(take(n * 8));
    w.enc = reinterpret_cast<float*>(take(n * 16 * 2 * 4));
    w.sigmas = reinterpret_cast<float*>(take(n * 4));
    w.rgbs = reinterpret_cast<float*>(take(n * 12));
    w.scratch = reinterpret_cast<int32_t*>(take((kHdr + 2 * (n / kRayBlock + 2)) * 4));   // header, then the two per-chunk survivor count arrays
    w.partials[0] = reinterpret_cast<int32_t*>(take((kMaxMarchBlocks + kHostedBlocks) * 4));
    w.partials[1] = reinterpret_cast<int32_t*>(take((kMaxMarchBlocks + kHostedBlocks) * 4));
    w.enc_pal = w.enc_clip = w.aux = nullptr;
    w.edit = nullptr;
    if (aux_stride) {
        w.edit = take(pnr_internal_edit_device_bytes());
        w.enc_pal = reinterpret_cast<float*>(take(n * 16 * 2 * 4));
        if (with_clip) w.enc_clip = reinterpret_cast<float*>(take(n * 16 * 2 * 4));
        w.aux = reinterpret_cast<float*>(take(n * aux_stride * 4));
    }
    w.s_o = reinterpret_cast<float*>(take(n * 12)); w.s_d = reinterpret_cast<float*>(take(n * 12));
    w.s_near = reinterpret_cast<float*>(take(n * 4)); w.s_far = reinterpret_cast<float*>(take(n * 4));
    w.s_ws = reinterpret_cast<float*>(take(n * 4)); w.s_depth = reinterpret_cast<float*>(take(n * 4));
    w.s_image = reinterpret_cast<float*>(take(n * 12));
    w.s_aux = aux_stride ? reinterpret_cast<float*>(take(n * aux_stride * 4)) : nullptr;
    // (the hosted tail's arrays come last: everything above keeps the offsets it had)
    w.hosted = reinterpret_cast<HostedConst*>(take(sizeof(HostedConst)));
    w.qctr = reinterpret_cast<int32_t*>(take(2 * kQueueCtrs * 4));
    w.qrecs = reinterpret_cast<StragglerRec*>(take(n * sizeof(StragglerRec)));
    w.rowflag = reinterpret_cast<uint8_t*>(take(n));
    w.bytes = off;
    return w;
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_interleave_tables(const float* a, const float* b, uint64_t rows, float* out, pnr_stream_t stream) {
    if (rows == 0) return PNR_OK;
    if (!a || !b || !out) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_interleave_tables, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(a),
                       reinterpret_cast<const float2*>(b), rows, reinterpret_cast<float4*>(out));
    return check_launch();
}

int pnr_interleave_tables3(const float* a, const float* b, const float* c, uint64_t rows, float* out, pnr_stream_t stream) {
    if (rows == 0) return PNR_OK;
    if (!a || !b || !c || !out) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_interleave_tables3, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(a),
                       reinterpret_cast<const float2*>(b), reinterpret_cast<const float2*>(c), rows, reinterpret_cast<float4*>(out));
    return check_launch();
}

uint64_t pnr_nerf_frame_workspace_bytes(uint32_t N) { return carve(nullptr, N).bytes; }
uint64_t pnr_palette_frame_workspace_bytes(uint32_t N, uint32_t num_basis, uint32_t clip_dim, int pred_clip) {
    return carve(nullptr, N, pnr_palette_aux_channels(num_basis, clip_dim), pred_clip != 0).bytes;
}

// phase: kWhole = the frame call; kSubmit = enqueue the frame's first chunk of iterations, its last launch and the control-block read-back, then return WITHOUT
// waiting (the caller prepares its next frame while this one runs); kFinish = wait for that read-back, enqueue further chunks while the frame is not done (the
// iteration count is data), fill stats / kernel_ms.  kFinish must follow kSubmit on the same host thread, device, stream and argument struct, with nothing of the
// frame's buffers touched in between; kWhole = kSubmit + kFinish.
enum FramePhase { kWhole = 0, kSubmit = 1, kFinish = 2 };
static int render_frame_impl(const pnr_nerf_frame_args* a, const pnr_palette_frame_args* pal, pnr_stream_t stream, FramePhase phase);

int pnr_nerf_render_frame(const pnr_nerf_frame_args* a, pnr_stream_t stream) { return render_frame_impl(a, nullptr, stream, kWhole); }
int pnr_nerf_render_frame_submit(const pnr_nerf_frame_args* a, pnr_stream_t stream) { return render_frame_impl(a, nullptr, stream, kSubmit); }
int pnr_nerf_render_frame_finish(const pnr_nerf_frame_args* a, pnr_stream_t stream) { return render_frame_impl(a, nullptr, stream, kFinish); }

static int palette_frame_check(const pnr_palette_frame_args* p) {
    if (!p) return PNR_ERR_INVALID;
    if (p->num_basis < 1 || p->num_basis > PNR_MAX_BASIS || p->clip_dim > PNR_MAX_CLIP) return PNR_ERR_UNSUPPORTED;
    if (p->edit && (p->edit->mode < 0 || p->edit->mode > 2)) return PNR_ERR_UNSUPPORTED;
    if (p->base.N && (!p->embeddings_palette || !p->aux_map || (p->pred_clip && !p->embeddings_clip))) return PNR_ERR_INVALID;
    return PNR_OK;
}
int pnr_palette_render_frame(const pnr_palette_frame_args* p, pnr_stream_t stream) {
    if (int rc = palette_frame_check(p)) return rc;
    return render_frame_impl(&p->base, p, stream, kWhole);
}
int pnr_palette_render_frame_submit(const pnr_palette_frame_args* p, pnr_stream_t stream) {
    if (int rc = palette_frame_check(p)) return rc;
    return render_frame_impl(&p->base, p, stream, kSubmit);
}
int pnr_palette_render_frame_finish(const pnr_palette_frame_args* p, pnr_stream_t stream) {
    if (int rc = palette_frame_check(p)) return rc;
    return render_frame_impl(&p->base, p, stream, kFinish);
}

static int render_frame_impl(const pnr_nerf_frame_args* a, const pnr_palette_frame_args* pal, pnr_stream_t stream, FramePhase phase) {
    if (!a) return PNR_ERR_INVALID;
    if (a->N == 0) return PNR_OK;
    if (!a->rays_o || !a->rays_d || !a->nears || !a->fars || !a->bitfield || !a->embeddings || !a->offsets || !a->packed_weights || !a->weights_sum ||
        !a->depth || !a->image || !a->workspace)
        return PNR_ERR_INVALID;
    if (a->C == 0 || a->C > 16 || a->H == 0 || a->max_steps == 0 || a->num_levels != 16) return PNR_ERR_UNSUPPORTED;
    if (a->field_precision != PNR_FIELD_FP32 && a->field_precision != PNR_FIELD_F16X3 && a->field_precision != PNR_FIELD_F16X2) return PNR_ERR_UNSUPPORTED;
    const uint32_t aux_stride = pal ? pnr_palette_aux_channels(pal->num_basis, pal->clip_dim) : 0;
    const bool with_clip = pal && pal->pred_clip;
    hipStream_t s = as_stream(stream);
    const uint32_t N = a->N;
    FrameWorkspace w = carve(a->workspace, N, aux_stride, with_clip);
    if (a->workspace_bytes < w.bytes) return PNR_ERR_INVALID;
    const bool sorted = a->ray_order != nullptr;
    const float *in_o = a->rays_o, *in_d = a->rays_d, *in_far = a->fars;   // (the nears are only read by the first launch: rays_t)
    float *out_ws = a->weights_sum, *out_depth = a->depth, *out_image = a->image, *out_aux = pal ? pal->aux_map : nullptr;
    if (sorted) {
        in_o = w.s_o; in_d = w.s_d; in_far = w.s_far;
        out_ws = w.s_ws; out_depth = w.s_depth; out_image = w.s_image; out_aux = pal ? w.s_aux : nullptr;
    }
    const float* tables[3] = {a->embeddings, pal ? pal->embeddings_palette : nullptr, with_clip ? pal->embeddings_clip : nullptr};
    const uint32_t n_enc = pal ? (with_clip ? 3u : 2u) : 1u;
    const int aux_fused = (pal && g_opt_aux_fusion && pnr_palette_field_stages_aux(pal->num_basis, pal->clip_dim, pal->pred_clip)) ? 1 : 0;
    const int composite_fused = (!pal && g_opt_composite_fusion) ? g_opt_composite_fusion : 0;   // NeRF: 1 = one-sample-per-ray iterations are composited inside the field kernel, 2 = all of them (no composite launch)
    const bool pal_composite_fused = pal && aux_fused && g_opt_composite_fusion == 2;   // PaletteNeRF: the ray state is composited inside the field kernel as well (needs the staged aux rows)
    const bool half_tables = a->table_dtype == PNR_DTYPE_F16;   // fp16 tables: nerf = `embeddings` as halves; palette = embeddings_pair as interleaved halves
    if (half_tables && pal && (with_clip || !pal->embeddings_pair)) return PNR_ERR_UNSUPPORTED;
    if (a->table_dtype != PNR_DTYPE_F32 && a->table_dtype != PNR_DTYPE_F16) return PNR_ERR_UNSUPPORTED;
    const float4* pair_table = (pal && !half_tables && !with_clip && pal->embeddings_pair) ? reinterpret_cast<const float4*>(pal->embeddings_pair) : nullptr;
    const float4* triple_table = (pal && !half_tables && with_clip && pal->embeddings_triple) ? reinterpret_cast<const float4*>(pal->embeddings_triple) : nullptr;
    pnr_palette_field_args pf = {};
    if (pal) {
        pf.enc = w.enc; pf.enc_palette = w.enc_pal; pf.enc_clip = w.enc_clip; pf.level_stride = N; pf.dirs = w.dirs; pf.deltas = w.deltas;
        pf.packed = a->packed_weights; pf.num_basis = pal->num_basis;
        pf.clip_dim = pal->clip_dim; pf.pred_clip = pal->pred_clip; pf.density_scale = a->density_scale; pf.offsets_weight = pal->offsets_weight;
        pf.view_dep_weight = pal->view_dep_weight; pf.aux_stride = aux_stride; pf.sigmas = w.sigmas; pf.rgbs = w.rgbs; pf.aux = w.aux;
        pf.precision = a->field_precision; pf.xyzs = w.xyzs;
        for (int k = 0; k < 3; k++) pf.enc_scale[k] = a->enc_scale[k];
        pf.overflow_flag = a->watch_overflow ? w.scratch + 1 : nullptr;
        pf.tile_counter = g_opt_dynamic_tiles ? w.scratch + 2 : nullptr;
        if (pal->edit && pal->edit->mode != 0) {   // RegionEdit / Stylizer: parameters uploaded once for the whole frame
            if (phase != kFinish) {
                const int rc = pnr_internal_edit_upload(pal->edit, w.edit, s);
                if (rc != PNR_OK) return rc;
            }
            pf.edit = pal->edit; pf.edit_device = w.edit;
        }
    }

    // per host thread AND per device (a process may drive several GPUs): the pinned read-back slot, the timing events and the iteration
    // prediction of the previous frame rendered there
    // (released when the host thread ends: a pool that replaces its worker threads does not accumulate pinned blocks and events)
    struct PerDevice {
        FrameCtl* host_ctl = nullptr; std::vector<hipEvent_t> ev; uint32_t predicted_iterations = 0; hipEvent_t done_ev = nullptr;
        // a frame submitted and not yet finished (pnr_*_render_frame_submit): what its finish call continues from
        struct Pending { bool on = false; const void* args = nullptr; int iter = 0; uint32_t alive_ub = 0, chunk = 0, looks = 0, prev_partials = 0; size_t ev_used = 0; } pending;
        ~PerDevice() {
            if (host_ctl) (void)hipHostFree(host_ctl);
            if (done_ev) (void)hipEventDestroy(done_ev);
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        }
    };
    static thread_local PerDevice per_device[kMaxDevices];
    PerDevice& dev_state = per_device[current_device()];
    FrameCtl*& host_ctl = dev_state.host_ctl;  // one in-flight frame per host thread and device
    if (!host_ctl && hipHostMalloc(reinterpret_cast<void**>(&host_ctl), sizeof(FrameCtl), hipHostMallocPortable) != hipSuccess) return PNR_ERR_LAUNCH;

    const float enc_scale = a->enc_scale[0] > 0.0f ? a->enc_scale[0] : 1.0f;
    const bool use_mip = a->mip && (a->H % 4) == 0 && pnr_occupancy_mip_bytes(a->C, a->H) <= 64 * 1024;
    const bool pow2 = is_pow2f(a->bound) && (a->H & (a->H - 1)) == 0;
    const MarchParams mp = make_march_params(a->bound, a->dt_gamma, a->max_steps, a->C, a->H, use_mip);
    const uint32_t march_lds = mp.mip_words ? (2 * mp.mip_words + 8) * 4 : 0;
    const LevelParams lp = make_level_params(16, a->S, a->base_resolution);
    const uint32_t* mip = static_cast<const uint32_t*>(a->mip);
    const bool hosted = g_opt_hosted_tail && use_mip && pow2 && mp.mip_words != 0 && (a->H % 64u) == 0;   // (what MODE 2 and hosted_march_tail are compiled for)

    const uint32_t cstride = N / kRayBlock + 2;
    auto counts_of = [&](int parity) { return w.scratch + kHdr + (uint32_t)(parity & 1) * cstride; };   // iteration i fills counts_of(i), its march reads counts_of(i + 1)
    HostedConst hconst = {};
    hconst.qctr_all = w.qctr; hconst.qrecs = w.qrecs; hconst.rays_o = in_o; hconst.rays_d = in_d; hconst.bitfield = a->bitfield; hconst.mip = mip; hconst.p = mp;
    hconst.xyzs = w.xyzs; hconst.dirs = w.dirs; hconst.deltas = w.deltas; hconst.partials[0] = w.partials[0]; hconst.partials[1] = w.partials[1];
    FrameBegin fb = {};
    fb.order = a->ray_order; fb.rays_o = a->rays_o; fb.rays_d = a->rays_d; fb.nears_in = a->nears; fb.fars_in = a->fars;
    fb.aabb = a->aabb; fb.min_near = a->min_near; fb.nears_out = a->nears; fb.fars_out = a->fars;
    fb.so = w.s_o; fb.sd = w.s_d; fb.sf = w.s_far;
    auto& pending = dev_state.pending;
    if (phase == kFinish) {
        if (!pending.on || pending.args != static_cast<const void*>(pal ? static_cast<const void*>(pal) : static_cast<const void*>(a))) return PNR_ERR_INVALID;   // no frame of THIS struct was submitted on this thread and device
    } else {
        // one submitted frame per host thread and device: a second _submit is refused (finish the first one).  A WHOLE-frame call drops a submitted frame that
        // was never finished (the caller gave it up -- an exception between its two halves): its launches are in the stream in front of this frame's, nothing
        // waits for them any more, and this call would otherwise be refused for as long as the thread lives.
        if (pending.on && phase == kSubmit) return PNR_ERR_INVALID;
        pending.on = false;
        if (pal) {   // the aux map starts at zero (palette/renderer.py:436-441): inside the first launch when rows are float4-aligned
            if ((aux_stride & 3u) == 0 && (reinterpret_cast<uintptr_t>(out_aux) & 15u) == 0) { fb.aux_zero = out_aux; fb.aux_stride = aux_stride; }
            else if (hipMemsetAsync(out_aux, 0, (size_t)N * aux_stride * 4, s) != hipSuccess) return PNR_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(k_frame_begin, dim3(cdiv(N, kRayBlock)), dim3(kRayBlock), 0, s, N, fb, w.alive[1], w.rays_t, out_ws, out_depth, out_image,
                           w.ctl, counts_of(1), w.scratch, w.qctr, hconst, w.hosted);
    }
    // optional live timing of the roofline kernel: HIP events on the launch stream around every k_frame_grid launch
    std::vector<hipEvent_t>& ev = dev_state.ev;
    size_t ev_used = 0;
    auto next_event = [&]() -> hipEvent_t {
        if (ev_used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev.push_back(e); }
        return ev[ev_used++];
    };
    const bool timing = a->kernel_ms != nullptr;
    FrameFinish fin = {};
    fin.on = a->finish; fin.bg[0] = a->bg_color[0]; fin.bg[1] = a->bg_color[1]; fin.bg[2] = a->bg_color[2];
    fin.bg_map = a->bg_map; fin.nears = a->nears; fin.fars = a->fars;   // indexed by ray id
    fin.depth_raw = a->depth_raw;
    auto launch_last = [&](const FrameCtl* done_ctl) {
        fin.ctl = done_ctl;
        if (sorted && pal && aux_stride > 64)
            hipLaunchKernelGGL(k_frame_unsort_outputs<32>, dim3(cdiv(N, kRayBlock / 32)), dim3(kRayBlock), 0, s, N, a->ray_order, w.s_ws, w.s_depth, w.s_image,
                               w.s_aux, aux_stride, a->weights_sum, a->depth, a->image, pal->aux_map, fin);
        else if (sorted && pal)
            hipLaunchKernelGGL(k_frame_unsort_outputs<16>, dim3(cdiv(N * 16, kRayBlock)), dim3(kRayBlock), 0, s, N, a->ray_order, w.s_ws, w.s_depth, w.s_image,
                               w.s_aux, aux_stride, a->weights_sum, a->depth, a->image, pal->aux_map, fin);
        else if (sorted)
            hipLaunchKernelGGL(k_frame_unsort_outputs<4>, dim3(cdiv(N * 4, kRayBlock)), dim3(kRayBlock), 0, s, N, a->ray_order, w.s_ws, w.s_depth, w.s_image,
                               (const float*)nullptr, 0u, a->weights_sum, a->depth, a->image, (float*)nullptr, fin);
        else if (fin.on || fin.depth_raw)   // unsorted frame: the same kernel in place
            hipLaunchKernelGGL(k_frame_unsort_outputs<4>, dim3(cdiv(N * 4, kRayBlock)), dim3(kRayBlock), 0, s, N, (const int32_t*)nullptr, a->weights_sum, a->depth,
                               a->image, (const float*)nullptr, aux_stride, a->weights_sum, a->depth, a->image, pal ? pal->aux_map : (float*)nullptr, fin);
    };
    uint32_t alive_ub = N;   // host-side upper bound of n_alive (it only shrinks)
    // Iterations enqueued between two looks at the control block.  Consecutive frames of a camera path need nearly the same
    // number of iterations, so the first chunk is the previous frame's count (one look per frame when the guess holds; launches
    // past the end are no-ops that cost a few microseconds each); after that, short chunks that grow for long, translucent marches.
    uint32_t& predicted_iterations = dev_state.predicted_iterations;
    // (+1: the launch that finds no ray left is the one that reports it; + g_opt_iteration_margin spare iterations.  Along a camera path the
    // count drifts by one or two from frame to frame; a spare iteration is four early-exit launches (~19 us), a wrong guess one host round
    // trip.  Measured on the moving-camera benchmark the round trip is the cheaper of the two: the margin defaults to 0)
    const uint32_t want = predicted_iterations + 1u + (uint32_t)g_opt_iteration_margin;
    uint32_t chunk = predicted_iterations ? (want < 1024u ? want : 1024u) : 8u;
    uint32_t looks = 0;
    uint32_t prev_partials = 0;   // workgroups of the previous march launch (= sample partials to add up)
    int iter = 0;
    if (phase == kFinish) {   // continue where the submit call stopped: its chunk, the frame's last launch and the read-back are in the stream
        iter = pending.iter; alive_ub = pending.alive_ub; chunk = pending.chunk; looks = pending.looks; prev_partials = pending.prev_partials; ev_used = pending.ev_used;
        pending.on = false;
    }
    auto enqueue_chunk = [&]() -> int {
        for (uint32_t k = 0; k < chunk; k++, iter++) {
            FrameCtl* cur = w.ctl + (iter & 1);                 // this iteration's control block, written by its march launch
            const FrameCtl* prev = w.ctl + ((iter + 1) & 1);    // the previous iteration's (k_frame_begin's in front of iteration 0)
            int32_t* alive_in = w.alive[iter & 1];              // this iteration's compacted list (the march writes it, the composite punches holes)
            const int32_t* alive_prev = w.alive[(iter + 1) & 1];
            const uint32_t ray_blocks = cdiv(alive_ub, kRayBlock);
            const uint32_t rows_ub = (uint64_t)alive_ub * 8 < N ? alive_ub * 8 : N;
            // hosted tail (MODE 2): the march gives every ray `budget` sample-less probes and queues the rest for the lookup launch's first workgroups
            const uint32_t budget = hosted ? (uint32_t)(iter == 0 ? g_opt_march_budget0 : g_opt_march_budget) : 0u;
            const int mode = budget ? 2 : 1;
            // MODE 2 runs five workgroups per CU (1 280 resident).  A typical later lego launch has 1 352 chunks: its last 72 workgroups start ~8 us late
            // (launch 16.9 us), and capping the launch at 1 280 is no way out -- a workgroup's second chunk waits at the block barriers for the slowest
            // wave of its first one (18.4 us).  With three chunks and more per resident workgroup (garden: 4 256) the cap does pay: the chunks of a
            // workgroup share its prologue (mip staging, chunk sums): 38.7 -> 33.1 us per launch.  "march_blocks" overrides (0 / 65536 = this rule).
            const uint32_t kResident = 1280;
            uint32_t march_cap = kMaxMarchBlocks;
            if (mode == 2) {
                if (g_opt_march_blocks > 0 && g_opt_march_blocks < 65536) march_cap = (uint32_t)g_opt_march_blocks < kMaxMarchBlocks ? (uint32_t)g_opt_march_blocks : kMaxMarchBlocks;
                else if (ray_blocks >= 2 * kResident) march_cap = kResident;
            }
            const dim3 gm(ray_blocks < march_cap ? ray_blocks : march_cap), bm(kRayBlock);
#define PNR_LAUNCH_MARCH(MIPV, P2V, MODEV)                                                                                                                \
            hipLaunchKernelGGL((k_frame_march<MIPV, P2V, MODEV>), gm, bm, march_lds, s, prev, cur, alive_prev, alive_in, counts_of(iter + 1), counts_of(iter), w.scratch, N, a->max_steps,    \
                               w.partials[(iter + 1) & 1], prev_partials, w.rays_t, in_o, in_d, mp, a->bitfield, in_far, w.xyzs, w.dirs, w.deltas, mip,           \
                               w.partials[iter & 1], budget, w.qctr, w.qrecs, w.rowflag)
            if (mode == 2) PNR_LAUNCH_MARCH(true, true, 2);   // (hosted implies the mip and power-of-two configuration)
            else if (use_mip && pow2) PNR_LAUNCH_MARCH(true, true, 1);
            else if (use_mip) PNR_LAUNCH_MARCH(true, false, 1);
            else if (pow2) PNR_LAUNCH_MARCH(false, true, 1);
            else PNR_LAUNCH_MARCH(false, false, 1);
#undef PNR_LAUNCH_MARCH
            if (k + 1 == chunk) {
                // The look: the march launch is the only writer of the control block (sample and row totals of everything in front of it, the overflow flag,
                // `done`), so the chunk's last one is read back right behind itself -- the host wakes up while that iteration's lookup and field
                // launches (empty when the frame is done, which is what the chunk length bets on) and the frame's last launch are still running.
                // Wait for THIS read-back, not for the stream: another host thread may already have queued the next frame behind it (pipeline.FramesInFlight
                // with a shared stream: frames back to back without the host's gap between them, kernels never overlapping)
                if (hipMemcpyAsync(host_ctl, cur, sizeof(FrameCtl), hipMemcpyDeviceToHost, s) != hipSuccess) return PNR_ERR_LAUNCH;
                if (!dev_state.done_ev && hipEventCreateWithFlags(&dev_state.done_ev, hipEventDisableTiming) != hipSuccess) return PNR_ERR_LAUNCH;
                if (hipEventRecord(dev_state.done_ev, s) != hipSuccess) return PNR_ERR_LAUNCH;
            }
            const uint32_t gx = cdiv(rows_ub, 256);
            const uint32_t gxc = gx < 1024u ? gx : 1024u;
            HostedArgs ha = {};
            if (mode == 2) { ha.hc = w.hosted; ha.rowflag = w.rowflag; ha.blocks = kHostedBlocks; ha.partial_base = gm.x; ha.gx = gxc; ha.n_tab = n_enc; }
            const uint32_t grid_lds = mode == 2 ? march_lds : 0u;
            GridArgs ga = {};
            ga.xyzs = w.xyzs; ga.deltas = w.deltas; ga.offsets = a->offsets; ga.lp = lp; ga.level_stride = N; ga.bound = a->bound; ga.two_bound = 2.0f * a->bound;
            ga.inv_two_bound = exact_reciprocal_or_zero(ga.two_bound);
            ga.gridtype = a->gridtype;
            ga.enc[0] = w.enc; ga.enc[1] = w.enc_pal; ga.enc[2] = w.enc_clip;
            // live timing of the roofline kernel: the launch carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch's begin and end
            // time stamps, what rocprofv3 reports) -- events recorded around the launch are packets of their own and measured 79.7 us where the
            // kernel took 71.0
            hipEvent_t e0 = timing ? next_event() : nullptr, e1 = timing ? next_event() : nullptr;
#define PNR_LAUNCH_GRID(KERNEL, GRID) hipExtLaunchKernelGGL(KERNEL, (ha.blocks ? dim3(ha.blocks + (GRID).x * (GRID).y * (GRID).z) : (GRID)), dim3(256), grid_lds, s, e0, e1, 0, cur, ga, ha)
            if (half_tables && pal) {
                ga.table[0] = pal->embeddings_pair;
                PNR_LAUNCH_GRID(k_frame_grid_h2, dim3(gxc, 16));
            } else if (half_tables) {
                ga.table[0] = a->embeddings;
                PNR_LAUNCH_GRID(k_frame_grid_h1, dim3(gxc, 16));
            } else if (triple_table) {
                ga.table[0] = triple_table;
                PNR_LAUNCH_GRID(k_frame_grid_triple, dim3(gxc, 16));
            } else if (pair_table) {
                ga.table[0] = pair_table;
                PNR_LAUNCH_GRID(k_frame_grid_pair, dim3(gxc, PNR_FRAME_LEVEL_PAIRS_PAL ? 8 : 16));
            } else {
                for (int k = 0; k < 3; k++) ga.table[k] = tables[k];
                PNR_LAUNCH_GRID(k_frame_grid, dim3(gxc, PNR_FRAME_LEVEL_PAIRS == 2 ? 4 : (PNR_FRAME_LEVEL_PAIRS ? 8 : 16), n_enc));
            }
#undef PNR_LAUNCH_GRID
            if (pal) {
                pf.ctl = cur; pf.B = rows_ub;
                if (aux_fused) { pf.rays_alive = alive_in; pf.weights_sum = out_ws; pf.aux_map = out_aux; pf.T_thresh = a->T_thresh; }
                if (pal_composite_fused) {   // the field kernel does the whole compositing step: no composite launch
                    pf.rays_t = w.rays_t; pf.weights_sum_rw = out_ws; pf.depth = out_depth; pf.image = out_image; pf.rays_alive_rw = alive_in; pf.counts_cur = counts_of(iter);
                }
                const int rc = pnr_palette_field_forward(&pf, stream);
                if (rc != PNR_OK) return rc;
            } else if (a->field_precision == PNR_FIELD_FP32)
                hipLaunchKernelGGL((k_frame_field<0, false>), dim3(gx < 512u ? gx : 512u), dim3(kFieldThreads), 0, s, cur, w.enc, N, w.dirs, w.deltas,
                                   a->packed_weights, a->density_scale, enc_scale, w.sigmas, w.rgbs, composite_fused, a->T_thresh, alive_in, w.rays_t, out_ws, out_depth,
                                   out_image, w.scratch, counts_of(iter));
            else if (a->field_precision == PNR_FIELD_F16X2 && a->watch_overflow)
                hipLaunchKernelGGL((k_frame_field<2, true>), dim3(gx < 512u ? gx : 512u), dim3(kFieldThreads), 0, s, cur, w.enc, N, w.dirs, w.deltas,
                                   a->packed_weights, a->density_scale, enc_scale, w.sigmas, w.rgbs, composite_fused, a->T_thresh, alive_in, w.rays_t, out_ws, out_depth,
                                   out_image, w.scratch, counts_of(iter));
            else if (a->field_precision == PNR_FIELD_F16X2)
                hipLaunchKernelGGL((k_frame_field<2, false>), dim3(gx < 512u ? gx : 512u), dim3(kFieldThreads), 0, s, cur, w.enc, N, w.dirs, w.deltas,
                                   a->packed_weights, a->density_scale, enc_scale, w.sigmas, w.rgbs, composite_fused, a->T_thresh, alive_in, w.rays_t, out_ws, out_depth,
                                   out_image, w.scratch, counts_of(iter));
            else if (a->watch_overflow)
                hipLaunchKernelGGL((k_frame_field<1, true>), dim3(gx < 512u ? gx : 512u), dim3(kFieldThreads), 0, s, cur, w.enc, N, w.dirs, w.deltas,
                                   a->packed_weights, a->density_scale, enc_scale, w.sigmas, w.rgbs, composite_fused, a->T_thresh, alive_in, w.rays_t, out_ws, out_depth,
                                   out_image, w.scratch, counts_of(iter));
            else
                hipLaunchKernelGGL((k_frame_field<1, false>), dim3(gx < 512u ? gx : 512u), dim3(kFieldThreads), 0, s, cur, w.enc, N, w.dirs, w.deltas,
                                   a->packed_weights, a->density_scale, enc_scale, w.sigmas, w.rgbs, composite_fused, a->T_thresh, alive_in, w.rays_t, out_ws, out_depth,
                                   out_image, w.scratch, counts_of(iter));
            if (composite_fused != 2 && !pal_composite_fused)   // (the field kernels composite every iteration themselves)
                hipLaunchKernelGGL(k_frame_composite, gm, bm, 0, s, cur, a->T_thresh, alive_in, w.rays_t, w.sigmas, w.rgbs, w.deltas, out_ws, out_depth, out_image,
                                   counts_of(iter), (const float*)w.aux, out_aux, aux_stride, aux_fused, composite_fused);
            prev_partials = gm.x + ha.blocks;
        }
        // The frame's last launch goes out BEHIND the read-back and BEFORE the host waits for it: the kernel looks at the same control block and does
        // nothing unless the frame is done, so a chunk that fell short costs an empty launch -- and when the guess holds (nearly always along a camera
        // path) the host wakes up, returns and prepares the caller's next frame while this launch runs, instead of launching it after waking up
        // (an eighth of the garden frame: 0.40 of 2.4 ms were the host's turnaround between frames).  The in-place finish of an unsorted frame is
        // not idempotent across looks either way: guarded by the same flag.
        launch_last(w.ctl + ((iter - 1) & 1));
        return PNR_OK;
    };
    if (phase != kFinish) {
        if (int rc = enqueue_chunk()) return rc;
        if (phase == kSubmit) {
            pending.on = true; pending.args = pal ? static_cast<const void*>(pal) : static_cast<const void*>(a);
            pending.iter = iter; pending.alive_ub = alive_ub; pending.chunk = chunk; pending.looks = looks; pending.prev_partials = prev_partials; pending.ev_used = ev_used;
            return check_launch();
        }
    }
    for (;;) {
        if (hipEventSynchronize(dev_state.done_ev) != hipSuccess) return PNR_ERR_LAUNCH;     // (polling hipEventQuery instead measured the same: the runtime's wait already spins)
        if (host_ctl->done) break;
        alive_ub = (uint32_t)host_ctl->n_alive;
        if (looks == 0) chunk = predicted_iterations ? 4u : 8u;
        if (++looks >= 4 && chunk < 64) chunk *= 2;
        if (int rc = enqueue_chunk()) return rc;
    }
    predicted_iterations = (uint32_t)host_ctl->iterations;
    if (timing) {  // only the iterations that did work (the tail of the last chunk are no-op launches)
        float total = 0.0f;
        uint32_t counted = 0;
        for (size_t i = 0; i + 1 < ev_used && counted < (uint32_t)host_ctl->iterations; i += 2, counted++) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) total += ms;
        }
        a->kernel_ms[0] = total;
        a->kernel_ms[1] = (float)counted * ((pair_table || triple_table || half_tables) ? 1.0f : (float)n_enc);  // a k_frame_grid launch covers n_enc tables (count table-launches); the pair kernel is one launch for both
    }
    if (a->stats) {
        a->stats[0] = (uint64_t)host_ctl->iterations;
        a->stats[1] = host_ctl->rendered;
        a->stats[2] = host_ctl->rows;
        a->stats[3] = (uint64_t)iter;  // iterations enqueued (>= executed)
        a->stats[4] = (uint64_t)looks + 1;  // host looks at the control block (stream synchronisations) this frame took
        a->stats[5] = (uint64_t)(host_ctl->pad0 != 0);  // an operand of the split-fp16 field left fp16's range (watch_overflow): render again in fp32
    }
    return check_launch();
}

}  // extern "C"

#ifdef PNR_MARCH_TIMING
extern "C" int pnr_debug_march_timing(unsigned long long* out, int iteration) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_march_timing), sizeof(unsigned long long) * pnr::kTimingWaves * 8) != hipSuccess) return -3;
    if (iteration >= 0) hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_march_timing_iter), &iteration, sizeof(int));
    std::vector<unsigned long long> z((size_t)pnr::kTimingWaves * 8, 0ull);
    hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_march_timing), z.data(), z.size() * sizeof(unsigned long long));
    return 0;
}
#endif
#ifdef PNR_HOSTED_TIMING
extern "C" int pnr_debug_hosted_timing(unsigned long long* out, int iteration) {
    hipDeviceSynchronize();
    constexpr size_t kWords = 8 + 8 * 256 + 2 * 16 * 128;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_hosted_timing), sizeof(unsigned long long) * kWords) != hipSuccess) return -3;
    if (iteration >= 0) hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_hosted_timing_iter), &iteration, sizeof(int));
    std::vector<unsigned long long> z(kWords, 0ull);
    hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_hosted_timing), z.data(), z.size() * sizeof(unsigned long long));
    return 0;
}
#endif
#ifdef PNR_MARCH_STATS
extern "C" int pnr_debug_march_kinds(unsigned int* out) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_march_kinds), 64 * 8 * 4) == hipSuccess ? 0 : -3;
}
extern "C" int pnr_debug_march_hist(unsigned int* out) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_march_hist), 64 * 4) == hipSuccess ? 0 : -3;
}
extern "C" int pnr_debug_march_max(unsigned int* out, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_march_max), 256) != hipSuccess) return -3;
    if (reset) { unsigned int z[64] = {}; hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_march_max), z, 256); }
    return 0;
}
extern "C" int pnr_debug_march_stats(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pnr::g_march_stats), 64) != hipSuccess) return -3;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_march_stats), z, 64); }
    return 0;
}
#endif
