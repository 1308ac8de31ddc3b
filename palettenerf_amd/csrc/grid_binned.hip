// grid_binned.hip -- hash-grid table gradient without global atomics in the inner loop (D = 3, C = 2, fp32 tables).
//
// The scatter of gridencoder.cu:216-286 issues 8 corners x 2 channels float atomics per (sample, level): 160 M per 627 k-sample
// training batch.  On MI355X fp32 atomics execute memory-side (pinning a level's atomics to one XCD changes nothing), at
// 17-29 G/s that is 5.5 ms of a 13.6 ms PaletteNeRF step.  Here the table is cut into buckets of kBinRows rows (64 KiB of
// gradient = it fits the LDS of a CU) and the update is reorganised in three sweeps:
//   count    every workgroup histograms the buckets its (sample, level, corner) records fall into (LDS), adds the non-empty bins
//            to the global bucket counts;
//   scatter  after a scan of the counts, the same traversal writes each record (row inside its bucket: 2 bytes, weighted
//            gradient: 8 bytes) to its bucket's contiguous segment (a workgroup reserves its slice per bucket with one returning
//            atomic, ranks inside the slice come from LDS);
//   gather   one workgroup per (bucket, <= kBinChunk records): accumulates its records into a 64 KiB LDS image of the bucket
//            with LDS atomics, then adds the image to the table -- plainly when it owns the bucket, with global atomics on the
//            non-zero rows when a crowded bucket (the dense coarse levels) is split between several workgroups.
// Records cost 10 B each written + read once, the table is touched once per bucket.
// Three kinds of level (round 4): the coarsest (tables of at most two buckets) have no records at all -- workgroups keep fp64 LDS images of them and a
// reduce sums the images in a fixed order (k_coarse_image / k_coarse_reduce); on the mid levels (cells a few samples wide) the samples of a ray inside one
// cell are summed before their eight records are written (COMBINE = 2: a third of all records go); the fine levels write a record per sample and corner.
// 627 k samples: 0.93 -> 0.69 ms for the whole gradient (profiles/EXPERIMENTS.md has the sweep-by-sweep log).
// Sums are formed in a different order than the atomic scatter: same values up to fp32 rounding order, like any atomic run.
#include "pnr_common.hpp"
#include "grid_core.hpp"

namespace pnr {

constexpr uint32_t kBinRows = 8192;     // rows per bucket (x 2 channels x 4 B = 64 KiB)
constexpr uint32_t kBinChunk = 131072;  // records per gather workgroup (a uniformly hit bucket of a 627 k batch holds ~78 k: it stays in one piece)
// samples per thread x threads per workgroup of the count and the scatter sweeps (measured on the 627 k-sample batch, us, hashed + merged levels:
//   count   256 x 4: 47 + 40   512 x 4: 35 + 31   1024 x 4: 31 + 26   512 x 8: 33 + 33   512 x 2: 52 + 41   1024 x 1: 58 + 47   -- a histogram wants few, large workgroups
//   scatter 256 x 4: 284 + 142  512 x 2: 272 + 110  1024 x 1: 263 + 119  512 x 4: 291 + 149  256 x 2: 299 + 133  1024 x 2: 288 + 122  -- the scatter wants few registers per thread)
#ifndef PNR_COUNT_SAMPLES
#define PNR_COUNT_SAMPLES 4
#endif
#ifndef PNR_COUNT_THREADS
#define PNR_COUNT_THREADS 1024
#endif
#ifndef PNR_SCAT_SAMPLES
#define PNR_SCAT_SAMPLES 2
#endif
#ifndef PNR_SCAT_THREADS
#define PNR_SCAT_THREADS 512
#endif
constexpr uint32_t kCountSamples = PNR_COUNT_SAMPLES, kCountThreads = PNR_COUNT_THREADS;
constexpr uint32_t kBinSamples = PNR_SCAT_SAMPLES, kBinThreads = PNR_SCAT_THREADS;
constexpr uint32_t kMaxBucketsPerLaunch = 8192;  // LDS histogram bound (counts of one level's buckets)

struct BinJob { uint32_t row_base, nrows, rec_begin, rec_end, exclusive; };

// first global bucket of `level`: buckets are numbered level by level
__device__ __forceinline__ uint32_t level_bucket_base(const int32_t* __restrict__ offsets, uint32_t level) {
    uint32_t base = 0;
    for (uint32_t l = 0; l < level; l++) base += ((uint32_t)(offsets[l + 1] - offsets[l]) + kBinRows - 1) / kBinRows;
    return base;
}

// the 8 (row, weight) pairs of one sample on one level: gridencoder.cu:100-175 indexing, identical to k_grid_bwd
struct Corners { uint32_t row[8]; float w[8]; bool active; uint32_t cell; /* the cell's lower corner, 10 bits per axis (run merging: levels of at most 1023 cells a side) */ };
__device__ __forceinline__ Corners corners_of(const float* __restrict__ inputs, uint32_t b, uint32_t B, float scale, uint32_t resolution,
                                              uint32_t hashmap_size, uint32_t gridtype, bool align_corners) {
    Corners c;
    c.active = b < B;
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        const float v = c.active ? inputs[(size_t)b * 3 + d] : 0.0f;
        if (v < 0.0f || v > 1.0f) c.active = false;
        pos[d] = fmaf(v, scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1.0f;
        uint32_t pl[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        c.w[idx] = w;
        if (align_corners) c.row[idx] = grid_index<3, 1>(gridtype, true, hashmap_size, resolution, pl);
    }
    c.cell = c.active ? (pg[0] | (pg[1] << 10) | (pg[2] << 20)) : 0xFFFFFFFFu;
    // the row index per kind of level (grid_core.hpp: dense levels need no `%`, hashed levels with a power-of-two size a mask): the general form's
    // 32-bit modulo is ~25 instructions per corner, eight corners, in each of the three sweeps
    if (!align_corners) corner_rows_by_kind<1>(level_kind(gridtype, hashmap_size, resolution), gridtype, hashmap_size, resolution, pg, c.row);
    return c;
}

// A sample whose gradient on this level is exactly zero writes no record (round 5).  Training batches are full of them: composite_rays_train stops a ray at
// the sample where its transmittance falls below T_thresh (raymarching.cu:660-672), every sample behind it gets an exact zero from the composite's
// backward and therefore from the MLP's -- on a trained scene most of a ray.  Adding zeros changes no sum; every sweep applies the same test, so the counts,
// the slices and the records agree.  (A dead sample also ends a merged run: two records where there was one, still the same sums.)
__device__ __forceinline__ float2 drop_dead(Corners& c, const float* __restrict__ grad, uint32_t level, uint32_t b, uint32_t B) {
    float2 g = make_float2(0.0f, 0.0f);
    if (c.active) g = *reinterpret_cast<const float2*>(grad + ((size_t)level * B + b) * 2);
    if (g.x == 0.0f && g.y == 0.0f) { c.active = false; c.cell = 0xFFFFFFFFu; }
    return g;
}

// Runs of consecutive lanes that hit the same table row (coarse levels: samples arrive ordered along rays, ~25 per cell on
// level 0) are merged into ONE record carried by the run's last lane -- same ballot + segmented-scan scheme as k_grid_bwd<COMBINE>.
struct Run { bool tail; int start; };
__device__ __forceinline__ Run run_of(uint32_t key, int lane) {
    const uint32_t prev = __shfl_up(key, 1, PNR_WAVE);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    Run r;
    r.start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));
    r.tail = lane == PNR_WAVE - 1 || ((heads >> (lane + 1)) & 1ull);
    return r;
}
__device__ __forceinline__ float run_sum(float v, const Run& r, int lane) {
#pragma unroll
    for (int off = 1; off < PNR_WAVE; off <<= 1) {
        const float up = __shfl_up(v, off, PNR_WAVE);
        if (lane - off >= r.start) v += up;
    }
    return v;
}

// Mid levels: consecutive samples of a ray that sit in the SAME CELL share all eight rows; their 8 x 2 weighted gradients are summed first and the
// run's last lane writes the eight records.  One comparison of the cell per lane (not one per corner) and a segmented sum on the DPP path (row_shr
// inside a row of 16 lanes: vector instructions, where the coarse levels' merge above costs 12 LDS-pipe shuffles per corner); runs are cut at
// 16-lane boundaries.  With a step of t/128 a sample spends 12, 8, 5.6, 3.9, 2.7, 1.9 steps in a cell of levels 2..7: a third of all records go.
struct CellRun { bool tail; uint32_t k; };     // k: lanes of the same run in front of this one (0..15)
__device__ __forceinline__ CellRun cell_run_of(uint32_t cell, int lane) {
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~cell, (int)cell, 0x111, 0xf, 0xf, false);   // row_shr:1; a row's first lane sees ~cell: a head
    const bool head = prev != cell || cell == 0xFFFFFFFFu;
    const unsigned long long heads = __ballot(head);
    CellRun r;
    r.k = (uint32_t)(lane - (63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)))));
    r.tail = lane == PNR_WAVE - 1 || ((heads >> (lane + 1)) & 1ull);
    return r;
}
template <int D> __device__ __forceinline__ float dpp_row_shr0(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true)); }
__device__ __forceinline__ float cell_run_sum(float v, uint32_t k) {
    float t = dpp_row_shr0<1>(v); v += k >= 1u ? t : 0.0f;
    t = dpp_row_shr0<2>(v); v += k >= 2u ? t : 0.0f;
    t = dpp_row_shr0<4>(v); v += k >= 4u ? t : 0.0f;
    t = dpp_row_shr0<8>(v); v += k >= 8u ? t : 0.0f;
    return v;
}

// One LDS counter update per (wave, distinct bucket) instead of one per lane: on the dense levels a whole wave lands in one or
// two buckets and per-lane atomics on a single LDS word serialise.  Returns the lane's rank inside its bucket's workgroup slice.
__device__ __forceinline__ uint32_t reserve_in_bucket(uint32_t* hist, uint32_t bucket, bool emit, int lane, bool aggregate) {
    if (!aggregate) return emit ? atomicAdd(&hist[bucket], 1u) : 0u;
    uint32_t rank = 0;
    unsigned long long todo = __ballot(emit);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b0 = (uint32_t)__shfl((int)bucket, leader, PNR_WAVE);
        const unsigned long long same = __ballot(emit && bucket == b0);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&hist[b0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, PNR_WAVE);
        if (emit && bucket == b0) rank = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return rank;
}

// sweep 1: bucket counts
template <int COMBINE>   // 0: a record per (sample, corner); 1: runs of equal rows merged per corner (coarse levels); 2: runs of samples in one cell merged (mid levels)
__global__ void __launch_bounds__(kCountThreads) k_bin_count(const float* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets, uint32_t B,
                                                           LevelParams lp, uint32_t gridtype, bool align_corners, uint32_t* __restrict__ counts,
                                                           uint32_t level0) {
    extern __shared__ uint32_t hist[];
    const uint32_t level = level0 + blockIdx.y;
    const int lane = threadIdx.x & (PNR_WAVE - 1);
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t nb = (hashmap_size + kBinRows - 1) / kBinRows;
    for (uint32_t k = threadIdx.x; k < nb; k += kCountThreads) hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kCountSamples; u++) {
        const uint32_t b = (blockIdx.x * kCountSamples + u) * kCountThreads + threadIdx.x;
        Corners c = corners_of(inputs, b, B, lp.scale[level], lp.resolution[level], hashmap_size, gridtype, align_corners);
        drop_dead(c, grad, level, b, B);
        [[maybe_unused]] CellRun cr = {true, 0u};
        if constexpr (COMBINE == 2) cr = cell_run_of(c.cell, lane);
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            bool emit = c.active;
            if constexpr (COMBINE == 1) {
                const Run r = run_of(c.active ? c.row[idx] : 0xFFFFFFFFu, lane);   // wave-collective: every lane takes part
                emit = emit && r.tail;
            }
            if constexpr (COMBINE == 2) emit = emit && cr.tail;
            reserve_in_bucket(hist, c.row[idx] / kBinRows, emit, lane, nb <= 8);
        }
    }
    __syncthreads();
    const uint32_t base = level_bucket_base(offsets, level);
    for (uint32_t k = threadIdx.x; k < nb; k += kCountThreads)
        if (hist[k]) atomicAdd(&counts[base + k], hist[k]);
}

// single workgroup: record offsets of the buckets (exclusive scan), the write cursors, and the gather job list
__global__ void __launch_bounds__(1024) k_bin_plan(const int32_t* __restrict__ offsets, uint32_t L, const uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ rec_off, uint32_t* __restrict__ cursor, BinJob* __restrict__ jobs,
                                                   uint32_t* __restrict__ n_jobs) {
    __shared__ uint32_t wsum[2][1024 / PNR_WAVE];
    __shared__ uint32_t carry[2];
    const uint32_t total_buckets = level_bucket_base(offsets, L);
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    for (uint32_t base = 0; base < total_buckets; base += 1024) {
        const uint32_t k = base + threadIdx.x;
        const uint32_t cnt = k < total_buckets ? counts[k] : 0;
        const uint32_t nj = (cnt + kBinChunk - 1) / kBinChunk;
        const uint32_t inc0 = (uint32_t)wave_inclusive_scan((int)cnt), inc1 = (uint32_t)wave_inclusive_scan((int)nj);
        if (lane == PNR_WAVE - 1) { wsum[0][wave] = inc0; wsum[1][wave] = inc1; }
        __syncthreads();
        uint32_t off0 = carry[0], off1 = carry[1], tot0 = 0, tot1 = 0;
        for (int wv = 0; wv < 1024 / PNR_WAVE; wv++) {
            if (wv < wave) { off0 += wsum[0][wv]; off1 += wsum[1][wv]; }
            tot0 += wsum[0][wv]; tot1 += wsum[1][wv];
        }
        if (k < total_buckets) {
            const uint32_t r0 = off0 + inc0 - cnt, j0 = off1 + inc1 - nj;
            rec_off[k] = r0;
            cursor[k] = r0;
            // which level / bucket-in-level is k?
            uint32_t level = 0, lb = 0;
            for (uint32_t l = 0; l < L; l++) {
                const uint32_t nbl = ((uint32_t)(offsets[l + 1] - offsets[l]) + kBinRows - 1) / kBinRows;
                if (k < lb + nbl) { level = l; break; }
                lb += nbl;
            }
            const uint32_t rows_l = (uint32_t)(offsets[level + 1] - offsets[level]);
            const uint32_t first_row = (k - lb) * kBinRows;
            for (uint32_t j = 0; j < nj; j++) {
                BinJob jb;
                jb.row_base = (uint32_t)offsets[level] + first_row;
                jb.nrows = rows_l - first_row < kBinRows ? rows_l - first_row : kBinRows;
                jb.rec_begin = r0 + j * kBinChunk;
                jb.rec_end = j + 1 < nj ? r0 + (j + 1) * kBinChunk : r0 + cnt;
                jb.exclusive = nj == 1;
                jobs[j0 + j] = jb;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += tot0; carry[1] += tot1; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { rec_off[total_buckets] = carry[0]; n_jobs[0] = carry[1]; n_jobs[1] = 0; /* the gather's job cursor */ }
}

// sweep 2: write the records into their buckets' segments
template <int COMBINE>
__global__ void __launch_bounds__(kBinThreads) k_bin_scatter(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                             const int32_t* __restrict__ offsets, uint32_t B, LevelParams lp, uint32_t gridtype,
                                                             bool align_corners, uint32_t* __restrict__ cursor, uint16_t* __restrict__ rec_row,
                                                             float2* __restrict__ rec_val, uint32_t level0) {
    extern __shared__ uint32_t lds[];   // [nb] counts, then [nb] slice bases
    const uint32_t level = level0 + blockIdx.y;
    const int lane = threadIdx.x & (PNR_WAVE - 1);
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t nb = (hashmap_size + kBinRows - 1) / kBinRows;
    uint32_t* hist = lds;
    uint32_t* slice = lds + nb;
    for (uint32_t k = threadIdx.x; k < nb; k += kBinThreads) hist[k] = 0;
    __syncthreads();
    Corners c[kBinSamples];
    float wy[kBinSamples][8];
    uint32_t rank[kBinSamples][8];
    uint32_t emit_bits[kBinSamples];
#pragma unroll
    for (uint32_t u = 0; u < kBinSamples; u++) {
        const uint32_t b = (blockIdx.x * kBinSamples + u) * kBinThreads + threadIdx.x;
        c[u] = corners_of(inputs, b, B, lp.scale[level], lp.resolution[level], hashmap_size, gridtype, align_corners);
        const float2 g = drop_dead(c[u], grad, level, b, B);
        emit_bits[u] = 0;
        [[maybe_unused]] CellRun cr = {true, 0u};
        if constexpr (COMBINE == 2) cr = cell_run_of(c[u].cell, lane);
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            bool emit = c[u].active;
            float vx = c[u].w[idx] * g.x, vy = c[u].w[idx] * g.y;
            if constexpr (COMBINE == 2) {
                vx = cell_run_sum(c[u].active ? vx : 0.0f, cr.k);
                vy = cell_run_sum(c[u].active ? vy : 0.0f, cr.k);
                emit = emit && cr.tail;
            }
            if constexpr (COMBINE == 1) {
                const Run r = run_of(c[u].active ? c[u].row[idx] : 0xFFFFFFFFu, lane);
                vx = run_sum(c[u].active ? vx : 0.0f, r, lane);
                vy = run_sum(c[u].active ? vy : 0.0f, r, lane);
                emit = emit && r.tail;
            }
            c[u].w[idx] = vx;                                    // from here on: the record's value (x in w[], y in wy[])
            wy[u][idx] = vy;
            rank[u][idx] = reserve_in_bucket(hist, c[u].row[idx] / kBinRows, emit, lane, nb <= 8);
            if (emit) emit_bits[u] |= 1u << idx;
        }
    }
    __syncthreads();
    const uint32_t base = level_bucket_base(offsets, level);
    for (uint32_t k = threadIdx.x; k < nb; k += kBinThreads) slice[k] = hist[k] ? atomicAdd(&cursor[base + k], hist[k]) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kBinSamples; u++) {
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            if (!((emit_bits[u] >> idx) & 1u)) continue;
            const uint32_t bucket = c[u].row[idx] / kBinRows;
            const uint32_t pos = slice[bucket] + rank[u][idx];
            rec_row[pos] = (uint16_t)(c[u].row[idx] % kBinRows);
            rec_val[pos] = make_float2(c[u].w[idx], wy[u][idx]);
        }
    }
}

// sweep 2, staged (round 5): the same records, written COALESCED.  Above, a lane writes its eight records where their buckets' slices are: the 64 lanes of a
// store instruction go to ~64 different cache lines, 8 (+ 2) bytes each -- the sweep ran at 1.4-1.9 TB/s of record bytes, bound by requests, not by bytes
// (profiles/r04_pmc_train.txt: 171 M wait cycles against 41 M active).  Here the workgroup's 8192 records are first ordered by bucket in LDS (96 KiB: the ranks
// inside a bucket are the ones reserve_in_bucket hands out anyway, the buckets' LDS offsets an exclusive scan of the workgroup's histogram) and then copied out
// by position: consecutive lanes write consecutive records of one bucket's slice -- runs of ~128 records, 16 full lines per store instruction.
constexpr uint32_t kStgThreads = 1024;                   // one sample per thread
constexpr uint32_t kStgRecords = kStgThreads * 8;
template <int COMBINE>
__global__ void __launch_bounds__(kStgThreads) k_bin_scatter_staged(const float* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
                                                                    uint32_t B, LevelParams lp, uint32_t gridtype, bool align_corners, uint32_t* __restrict__ cursor,
                                                                    uint16_t* __restrict__ rec_row, float2* __restrict__ rec_val, uint32_t level0, uint32_t nb_max) {
    extern __shared__ uint32_t lds[];   // [nb_max] counts | [nb_max] slice bases (global) | [nb_max + 1] LDS offsets | staged values float2[8192] | rows u16[8192] | buckets u16[8192]
    const uint32_t level = level0 + blockIdx.y;
    const int lane = threadIdx.x & (PNR_WAVE - 1);
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t nb = (hashmap_size + kBinRows - 1) / kBinRows;
    uint32_t* hist = lds;
    uint32_t* slice = lds + nb_max;
    uint32_t* lofs = lds + 2 * nb_max;
    float2* st_val = reinterpret_cast<float2*>(lds + ((3 * nb_max + 1 + 3) & ~3u));
    uint16_t* st_row = reinterpret_cast<uint16_t*>(st_val + kStgRecords);
    uint16_t* st_bkt = st_row + kStgRecords;
    for (uint32_t k = threadIdx.x; k < nb; k += kStgThreads) hist[k] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.x * kStgThreads + threadIdx.x;
    Corners c = corners_of(inputs, b, B, lp.scale[level], lp.resolution[level], hashmap_size, gridtype, align_corners);
    float wy[8];
    uint32_t rank[8], emit_bits = 0;
    {
        const float2 g = drop_dead(c, grad, level, b, B);
        [[maybe_unused]] CellRun cr = {true, 0u};
        if constexpr (COMBINE == 2) cr = cell_run_of(c.cell, lane);
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            bool emit = c.active;
            float vx = c.w[idx] * g.x, vy = c.w[idx] * g.y;
            if constexpr (COMBINE == 2) {
                vx = cell_run_sum(c.active ? vx : 0.0f, cr.k);
                vy = cell_run_sum(c.active ? vy : 0.0f, cr.k);
                emit = emit && cr.tail;
            }
            if constexpr (COMBINE == 1) {
                const Run r = run_of(c.active ? c.row[idx] : 0xFFFFFFFFu, lane);
                vx = run_sum(c.active ? vx : 0.0f, r, lane);
                vy = run_sum(c.active ? vy : 0.0f, r, lane);
                emit = emit && r.tail;
            }
            c.w[idx] = vx;
            wy[idx] = vy;
            rank[idx] = reserve_in_bucket(hist, c.row[idx] / kBinRows, emit, lane, nb <= 8);
            if (emit) emit_bits |= 1u << idx;
        }
    }
    __syncthreads();
    const uint32_t base = level_bucket_base(offsets, level);
    for (uint32_t k = threadIdx.x; k < nb; k += kStgThreads) slice[k] = hist[k] ? atomicAdd(&cursor[base + k], hist[k]) : 0u;
    if (threadIdx.x < PNR_WAVE) {     // exclusive scan of the workgroup's histogram: where a bucket's records start in the staging arrays
        uint32_t carry = 0;
        for (uint32_t k0 = 0; k0 < nb; k0 += PNR_WAVE) {
            const uint32_t k = k0 + (uint32_t)lane, v = k < nb ? hist[k] : 0u;
            const uint32_t inc = (uint32_t)wave_inclusive_scan((int)v);
            if (k < nb) lofs[k] = carry + inc - v;
            carry += (uint32_t)__shfl((int)inc, PNR_WAVE - 1, PNR_WAVE);
        }
        if (lane == 0) lofs[nb] = carry;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        if (!((emit_bits >> idx) & 1u)) continue;
        const uint32_t bucket = c.row[idx] / kBinRows, p = lofs[bucket] + rank[idx];
        st_val[p] = make_float2(c.w[idx], wy[idx]);
        st_row[p] = (uint16_t)(c.row[idx] % kBinRows);
        st_bkt[p] = (uint16_t)bucket;
    }
    __syncthreads();
    const uint32_t total = lofs[nb];
    for (uint32_t p = threadIdx.x; p < total; p += kStgThreads) {
        const uint32_t bucket = st_bkt[p], gp = slice[bucket] + (p - lofs[bucket]);
        rec_row[gp] = st_row[p];
        rec_val[gp] = st_val[p];
    }
}

// sweep 3: accumulate one job's records in LDS, add the bucket image to the table
__global__ void __launch_bounds__(1024) k_bin_gather(const BinJob* __restrict__ jobs, uint32_t* __restrict__ n_jobs /* [0] jobs, [1] cursor */,
                                                     const uint16_t* __restrict__ rec_row, const float2* __restrict__ rec_val,
                                                     float* __restrict__ grad_grid) {
    // fp64 accumulators: on gfx950 ds_add_f64 runs at 1.3 T lane-atomics/s where ds_add_f32 manages 0.2 T/s (profiles/micro/lds_atomics.hip),
    // and the bucket sums come out order-independent to fp32 precision as a bonus; 128 KiB of the CU's 160 KiB LDS
    extern __shared__ double acc[];   // [kBinRows][2]
    // one resident workgroup per CU, jobs handed out through a cursor (round 5: dealt by block index, a quarter of the CUs sat through a fourth round of
    // jobs while the others had finished three)
    __shared__ uint32_t next_job;
    const uint32_t total_jobs = n_jobs[0];
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) next_job = atomicAdd(&n_jobs[1], 1u);
        __syncthreads();
        const uint32_t job = next_job;
        if (job >= total_jobs) break;
        const BinJob jb = jobs[job];
        for (uint32_t i = threadIdx.x; i < jb.nrows * 2; i += 1024) acc[i] = 0.0;
        __syncthreads();
        constexpr uint32_t U = 8;   // records in flight per thread: the loop is a latency chain otherwise (~76 HBM round trips per job)
        uint32_t r = jb.rec_begin + threadIdx.x;
        for (; r + (U - 1) * 1024 < jb.rec_end; r += U * 1024) {
            uint32_t row[U];
            float2 v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { row[u] = rec_row[r + u * 1024]; v[u] = rec_val[r + u * 1024]; }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { atomicAdd(&acc[row[u] * 2], (double)v[u].x); atomicAdd(&acc[row[u] * 2 + 1], (double)v[u].y); }
        }
        for (; r < jb.rec_end; r += 1024) {
            const uint32_t row = rec_row[r];
            const float2 v = rec_val[r];
            atomicAdd(&acc[row * 2], (double)v.x);
            atomicAdd(&acc[row * 2 + 1], (double)v.y);
        }
        __syncthreads();
        float* dst = grad_grid + (size_t)jb.row_base * 2;
        if (jb.exclusive) {
            for (uint32_t i = threadIdx.x; i < jb.nrows * 2; i += 1024) dst[i] += (float)acc[i];
        } else {
            for (uint32_t i = threadIdx.x; i < jb.nrows * 2; i += 1024)
                if (acc[i] != 0.0) unsafeAtomicAdd(dst + i, (float)acc[i]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The coarsest levels (tables of at most 2 x kBinRows rows: 17^3 and 25^3 of the shipped grids) without records.  Every sample lands in the same few
// thousand rows there; as records they were 10 M per level squeezed into one or two buckets (run merging by wave shuffles, crowded cursors, split gather
// jobs: ~100 us per level).  Here a workgroup keeps an fp64 image of (half of) the level in LDS, walks its share of the samples with the lanes of a wave
// ~chunk/64 samples apart (different rays: different cells, no same-address pile-ups), adds the 8 x 2 weighted gradients with LDS atomics, and dumps
// the image as fp32; k_coarse_reduce sums the workgroups' images in a fixed order into the table gradient.  No global atomics, deterministic.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kImgThreads = 1024;
constexpr uint32_t kMaxImgJobs = 8;
constexpr uint32_t kImgBlocks = 256;     // image workgroups in total (one per CU: 128 KiB of LDS each)
struct ImgJobs { uint32_t n, level[kMaxImgJobs], row_lo[kMaxImgJobs]; };
__global__ void __launch_bounds__(kImgThreads) k_coarse_image(const float* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets, uint32_t B,
                                                              LevelParams lp, uint32_t gridtype, bool align_corners, ImgJobs jobs, float* __restrict__ partial /* [n][gridDim.x][kBinRows * 2] */) {
    extern __shared__ double img[];   // [kBinRows][2]
    const uint32_t level = jobs.level[blockIdx.y], row_lo = jobs.row_lo[blockIdx.y];
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t nrows = hashmap_size > row_lo ? (hashmap_size - row_lo < kBinRows ? hashmap_size - row_lo : kBinRows) : 0u;
    for (uint32_t i = threadIdx.x; i < kBinRows * 2; i += kImgThreads) img[i] = 0.0;
    __syncthreads();
    const uint32_t chunk = (B + gridDim.x - 1) / gridDim.x;                    // samples of this workgroup
    const uint32_t per_lane = (chunk + 63) / 64, per_thread = (per_lane + kImgThreads / 64 - 1) / (kImgThreads / 64);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * chunk, last = first + chunk < B ? first + chunk : B;
    const uint32_t lane_first = first + lane * per_lane, lane_last = lane_first + per_lane < last ? lane_first + per_lane : last;
    for (uint32_t j = 0; j < per_thread; j++) {
        const uint32_t b = lane_first + wave * per_thread + j;
        if (b >= lane_last) break;
        Corners c = corners_of(inputs, b, B, lp.scale[level], lp.resolution[level], hashmap_size, gridtype, align_corners);
        const float2 g = drop_dead(c, grad, level, b, B);
        if (!c.active) continue;
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            const uint32_t r = c.row[idx] - row_lo;
            if (r < nrows) { atomicAdd(&img[r * 2], (double)(c.w[idx] * g.x)); atomicAdd(&img[r * 2 + 1], (double)(c.w[idx] * g.y)); }
        }
    }
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kBinRows * 2);
    for (uint32_t i = threadIdx.x; i < nrows * 2; i += kImgThreads) out[i] = (float)img[i];
}
__global__ void __launch_bounds__(256) k_coarse_reduce(const int32_t* __restrict__ offsets, ImgJobs jobs, uint32_t G, const float* __restrict__ partial, float* __restrict__ grad_grid) {
    const uint32_t level = jobs.level[blockIdx.y], row_lo = jobs.row_lo[blockIdx.y];
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t nrows = hashmap_size > row_lo ? (hashmap_size - row_lo < kBinRows ? hashmap_size - row_lo : kBinRows) : 0u;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * 2) return;
    const float* src = partial + (size_t)blockIdx.y * G * (kBinRows * 2) + e;
    float s = 0.0f;
    uint32_t g = 0;
    for (; g + 8 <= G; g += 8) {          // eight loads in flight; summed in workgroup order either way
        float v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = src[(size_t)(g + u) * (kBinRows * 2)];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) s += v[u];
    }
    for (; g < G; g++) s += src[(size_t)g * (kBinRows * 2)];
    grad_grid[((size_t)offsets[level] + row_lo) * 2 + e] += s;
}

struct BinLayout { uint64_t counts, rec_off, cursor, n_jobs, jobs, rec_row, rec_val, img, total; uint32_t bucket_bound, job_bound; };
static BinLayout bin_layout(uint32_t B, uint32_t L, uint64_t total_rows) {
    BinLayout l;
    const uint64_t n_rec = (uint64_t)B * L * 8;
    l.bucket_bound = (uint32_t)(total_rows / kBinRows + L);
    l.job_bound = (uint32_t)(n_rec / kBinChunk + l.bucket_bound);
    auto al = [](uint64_t v) { return (v + 255) & ~uint64_t(255); };
    uint64_t o = 0;
    l.counts = o; o = al(o + (uint64_t)l.bucket_bound * 4);
    l.rec_off = o; o = al(o + ((uint64_t)l.bucket_bound + 1) * 4);
    l.cursor = o; o = al(o + (uint64_t)l.bucket_bound * 4);
    l.n_jobs = o; o = al(o + 8);
    l.jobs = o; o = al(o + (uint64_t)l.job_bound * sizeof(BinJob));
    l.rec_row = o; o = al(o + n_rec * 2);
    l.rec_val = o; o = al(o + n_rec * 8);
    l.img = o; o = al(o + (uint64_t)kImgBlocks * kBinRows * 2 * 4);       // the coarse levels' per-workgroup images
    l.total = o;
    return l;
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_grid_backward_binned_workspace_bytes(uint32_t B, uint32_t L, uint64_t total_rows) { return bin_layout(B, L, total_rows).total; }

int pnr_grid_encode_backward_binned(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                    uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint64_t total_rows,
                                    void* workspace, uint64_t workspace_bytes, pnr_stream_t stream) {
    if (D != 3 || C != 2 || L == 0 || L > kMaxLevels) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings || !workspace || total_rows == 0) return PNR_ERR_INVALID;
    const BinLayout lay = bin_layout(B, L, total_rows);
    if (workspace_bytes < lay.total) return PNR_ERR_INVALID;
    if (lay.bucket_bound > kMaxBucketsPerLaunch || (uint64_t)B * L * 8 >= (1ull << 32)) return PNR_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint32_t* counts = reinterpret_cast<uint32_t*>(ws + lay.counts);
    uint32_t* rec_off = reinterpret_cast<uint32_t*>(ws + lay.rec_off);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(ws + lay.cursor);
    uint32_t* n_jobs = reinterpret_cast<uint32_t*>(ws + lay.n_jobs);
    BinJob* jobs = reinterpret_cast<BinJob*>(ws + lay.jobs);
    uint16_t* rec_row = reinterpret_cast<uint16_t*>(ws + lay.rec_row);
    float2* rec_val = reinterpret_cast<float2*>(ws + lay.rec_val);
    const LevelParams lp = make_level_params(L, S, H);
    if (hipMemsetAsync(counts, 0, (size_t)lay.bucket_bound * 4, s) != hipSuccess) return PNR_ERR_LAUNCH;
    // leading levels whose whole table is at most two LDS images: accumulated in place (k_coarse_image), no records
    ImgJobs ij;
    ij.n = 0;
    uint32_t ni = 0;
    if (g_opt_coarse_image) {
        while (ni < L) {
            const uint64_t side = (uint64_t)lp.resolution[ni] + (align_corners ? 0u : 1u);
            const uint64_t rows_bound = (side * side * side + 7) / 8 * 8;
            const uint32_t halves = (uint32_t)((rows_bound + kBinRows - 1) / kBinRows);
            if (halves > 2 || ij.n + halves > kMaxImgJobs) break;
            for (uint32_t h = 0; h < halves; h++) { ij.level[ij.n] = ni; ij.row_lo[ij.n] = h * kBinRows; ij.n++; }
            ni++;
        }
    }
    if (ij.n) {
        static bool img_attr[kMaxDevices] = {};
        if (!ensure_dynamic_lds(k_coarse_image, kBinRows * 2 * 8, img_attr)) return PNR_ERR_LAUNCH;
        const uint32_t G = kImgBlocks / ij.n;
        float* partial = reinterpret_cast<float*>(ws + lay.img);
        hipLaunchKernelGGL(k_coarse_image, dim3(G, ij.n), dim3(kImgThreads), kBinRows * 2 * 8, s, grad, inputs, offsets, B, lp, gridtype, align_corners != 0, ij, partial);
        hipLaunchKernelGGL(k_coarse_reduce, dim3(cdiv(kBinRows * 2, 256), ij.n), dim3(256), 0, s, offsets, ij, G, partial, grad_embeddings);
        if (ni == L) return check_launch();
    }
    uint32_t nc = ni;   // levels whose cells are wide compared with the sample spacing: run-combined records (as k_grid_bwd<COMBINE>)
    // (with fp64 LDS accumulation only the two coarsest levels still gain from merging runs: 3.47 -> 3.29 ms/step against the former bound of 384)
    while (nc < L && lp.scale[nc] <= 24.0f) nc++;
    uint32_t nm = nc;   // mid levels: samples of one cell merged (cells of at most 1023 a side, wide enough for a step to stay inside for ~2 samples)
    if (g_opt_cell_merge) while (nm < L && lp.scale[nm] <= 256.0f) nm++;
    const uint32_t gx = cdiv(B, kBinThreads * kBinSamples), gxc = cdiv(B, kCountThreads * kCountSamples);
    const uint32_t hist_bytes = lay.bucket_bound * 4;
    const bool ac = align_corners != 0;
    if (nc > ni) hipLaunchKernelGGL(k_bin_count<1>, dim3(gxc, nc - ni), dim3(kCountThreads), hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac, counts, ni);
    if (nm > nc) hipLaunchKernelGGL(k_bin_count<2>, dim3(gxc, nm - nc), dim3(kCountThreads), hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac, counts, nc);
    if (nm < L) hipLaunchKernelGGL(k_bin_count<0>, dim3(gxc, L - nm), dim3(kCountThreads), hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac, counts, nm);
    hipLaunchKernelGGL(k_bin_plan, dim3(1), dim3(1024), 0, s, offsets, L, counts, rec_off, cursor, jobs, n_jobs);
    const uint32_t nb_max = lay.bucket_bound;      // (an upper bound of any level's bucket count: the host does not know the levels' sizes, the offsets live on the device)
    const uint32_t stg_lds = ((3 * nb_max + 1 + 3) & ~3u) * 4 + kStgRecords * (8 + 2 + 2);
    if (g_opt_scatter_staged && stg_lds <= 160 * 1024) {
        static bool stg_attr[3][kMaxDevices] = {};
        if (!ensure_dynamic_lds(k_bin_scatter_staged<0>, stg_lds, stg_attr[0]) || !ensure_dynamic_lds(k_bin_scatter_staged<1>, stg_lds, stg_attr[1]) ||
            !ensure_dynamic_lds(k_bin_scatter_staged<2>, stg_lds, stg_attr[2])) return PNR_ERR_LAUNCH;
        const uint32_t gs = cdiv(B, kStgThreads);
        if (nc > ni) hipLaunchKernelGGL(k_bin_scatter_staged<1>, dim3(gs, nc - ni), dim3(kStgThreads), stg_lds, s, grad, inputs, offsets, B, lp, gridtype, ac, cursor, rec_row, rec_val, ni, nb_max);
        if (nm > nc) hipLaunchKernelGGL(k_bin_scatter_staged<2>, dim3(gs, nm - nc), dim3(kStgThreads), stg_lds, s, grad, inputs, offsets, B, lp, gridtype, ac, cursor, rec_row, rec_val, nc, nb_max);
        if (nm < L) hipLaunchKernelGGL(k_bin_scatter_staged<0>, dim3(gs, L - nm), dim3(kStgThreads), stg_lds, s, grad, inputs, offsets, B, lp, gridtype, ac, cursor, rec_row, rec_val, nm, nb_max);
    } else {
    if (nc > ni) hipLaunchKernelGGL(k_bin_scatter<1>, dim3(gx, nc - ni), dim3(kBinThreads), 2 * hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac, cursor,
                                    rec_row, rec_val, ni);
    if (nm > nc) hipLaunchKernelGGL(k_bin_scatter<2>, dim3(gx, nm - nc), dim3(kBinThreads), 2 * hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac, cursor,
                                    rec_row, rec_val, nc);
    if (nm < L) hipLaunchKernelGGL(k_bin_scatter<0>, dim3(gx, L - nm), dim3(kBinThreads), 2 * hist_bytes, s, grad, inputs, offsets, B, lp, gridtype, ac,
                                   cursor, rec_row, rec_val, nm);
    }
    static bool attr_set[kMaxDevices] = {};
    if (!ensure_dynamic_lds(k_bin_gather, kBinRows * 2 * 8, attr_set)) return PNR_ERR_LAUNCH;
    const uint32_t gather_blocks = lay.job_bound < 256u ? lay.job_bound : 256u;     // (128 KiB of LDS each: one per CU)
    hipLaunchKernelGGL(k_bin_gather, dim3(gather_blocks), dim3(1024), kBinRows * 2 * 8, s, jobs, n_jobs, rec_row, rec_val, grad_embeddings);
    return check_launch();
}

}  // extern "C"
