// raymarch.hip -- occupancy-grid ray marching for gfx950 (MI355X).
//
// What it computes follows the reference's raymarching extension (cited per kernel); how it is
// organised does not: output slots of the training march come from a deterministic two-level
// wave64 prefix sum instead of global atomics, the alive-ray list is compacted on the device with
// the same scan, and all launches take an explicit stream.
#include "pnr_common.hpp"
#include "march_core.hpp"

namespace pnr {

constexpr uint32_t kBlock = 256;  // 4 waves; one ray / element per thread

// one thread per 4x4x4 brick: 64 Morton-consecutive cells = one aligned uint64 of the bitfield
__global__ void __launch_bounds__(256) k_build_mip(const unsigned long long* __restrict__ grid64, uint32_t nbricks, uint32_t words_per_mask,
                                                   uint32_t* __restrict__ mip) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long w = b < nbricks ? grid64[b] : 0ull;
    const unsigned long long any = __ballot(w != 0ull), all = __ballot(b < nbricks && w == ~0ull);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t word = b >> 5;  // first of the two 32-brick words this wave covers
        if (word < words_per_mask) { mip[word] = (uint32_t)any; mip[words_per_mask + word] = (uint32_t)all; }
        if (word + 1 < words_per_mask) { mip[word + 1] = (uint32_t)(any >> 32); mip[words_per_mask + word + 1] = (uint32_t)(all >> 32); }
    }
}

// ------------------------------------------------------------------------------------------
// utils
// ------------------------------------------------------------------------------------------

// reference raymarching.cu:95-148 (near_far_of: march_core.hpp, shared with the frame loops' first kernel)
__global__ void __launch_bounds__(kBlock) k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ aabb, uint32_t N, float min_near,
                                                     float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    near_far_of(rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], aabb, min_near, nears[n], fars[n]);
}

// reference raymarching.cu:166-201
__global__ void __launch_bounds__(kBlock) k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         float radius, uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float RPI = 0.3183098861837907f;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float Cq = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * Cq)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    coords[n * 2] = 2 * atan2f(sqrtf(x * x + z * z), y) * RPI - 1;
    coords[n * 2 + 1] = atan2f(z, x) * RPI;
}

// reference raymarching.cu:217-257
__global__ void __launch_bounds__(kBlock) k_morton3d(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(kBlock) k_morton3d_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];
    coords[n * 3] = (int32_t)gather3((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)gather3((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)gather3((uint32_t)(ind >> 2));
}

// reference raymarching.cu:271-292 ; one byte per thread, 2 x 16-byte loads
__global__ void __launch_bounds__(kBlock) k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4* g = reinterpret_cast<const float4*>(grid) + (size_t)n * 2;
    const float4 a = g[0], b = g[1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ------------------------------------------------------------------------------------------
// two-level exclusive scan over per-thread int values (blocks of kBlock)
// scratch layout (int32): [0]=base0 [1]=base1 [2]=total [3]=unused [4 .. 4+nblocks) block offsets,
// then (training march only) N per-ray counts
// ------------------------------------------------------------------------------------------
constexpr uint32_t kScanHdr = 4;

// block-wide exclusive scan; returns this thread's exclusive prefix, *block_total for everyone
__device__ __forceinline__ int block_exclusive_scan(int v, int* block_total) {
    __shared__ int wave_sums[kBlock / PNR_WAVE];
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    const int incl = wave_inclusive_scan(v);
    if (lane == PNR_WAVE - 1) wave_sums[wave] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < (int)(kBlock / PNR_WAVE); w++) {
        const int s = wave_sums[w];
        if (w < wave) wave_off += s;
        total += s;
    }
    __syncthreads();
    *block_total = total;
    return wave_off + incl - v;
}

// single-block pass over the block sums: in-place exclusive scan + bookkeeping of the counters
__global__ void __launch_bounds__(1024) k_scan_block_sums(int32_t* __restrict__ scratch, uint32_t nblocks, int32_t* __restrict__ counter,
                                                          uint32_t n_items, int32_t* __restrict__ total_out) {
    __shared__ int wsum[1024 / PNR_WAVE];
    __shared__ int carry_s;
    int32_t* sums = scratch + kScanHdr;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const int v = i < nblocks ? sums[i] : 0;
        const int incl = wave_inclusive_scan(v);
        if (lane == PNR_WAVE - 1) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 1024 / PNR_WAVE; w++) { const int s = wsum[w]; if (w < wave) woff += s; tot += s; }
        const int carry = carry_s;
        if (i < nblocks) sums[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int total = carry_s;
        scratch[2] = total;
        if (counter) {  // training march: reserve [counter0, counter0+total) and N ray rows
            scratch[0] = counter[0];
            scratch[1] = counter[1];
            counter[0] += total;
            counter[1] += (int32_t)n_items;
        } else {
            scratch[0] = 0;
            scratch[1] = 0;
        }
        if (total_out) total_out[0] = total;
    }
}

// ------------------------------------------------------------------------------------------
// training march (reference raymarching.cu:315-483), split at the two atomicAdd()s
// ------------------------------------------------------------------------------------------
template <bool MIP, bool POW2>
__global__ void __launch_bounds__(kBlock) k_march_train_count(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, MarchParams p, uint32_t N,
    const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ noises, int32_t* __restrict__ scratch,
    const uint32_t* __restrict__ mip, float* __restrict__ t_store /* [N][max_steps] or null: the sample parameters, for k_march_train_emit */) {
    const uint32_t* mip_lds = stage_mip(mip, MIP ? p.mip_words : 0);
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t max_steps = p.max_steps;
    int num_steps = 0;
    if (n < N) {
        RayCtx c;
        ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, p, grid, mip_lds);
        const BoxHit bh = clip_to_box(c, fars[n]);
    const float far = bh.far;
        float t = nears[n];
        t = fmaf(clampf(t * c.dt_gamma, c.dt_min, c.dt_max), noises[n], t);
        t = skip_to_box<MIP && POW2>(c, bh, t);
        float x, y, z, dt;
        while (t < far && (uint32_t)num_steps < max_steps) {
            if (march_probe<MIP, POW2>(c, t, x, y, z, dt)) {
                if (t_store) t_store[(size_t)n * max_steps + (uint32_t)num_steps] = t;
                num_steps++; t += dt;
            }
        }
        scratch[kScanHdr + gridDim.x + n] = num_steps;  // per-ray counts live behind the block sums
    }
    int total;
    (void)block_exclusive_scan(num_steps, &total);
    if (threadIdx.x == 0) scratch[kScanHdr + blockIdx.x] = total;
}

// The counting pass with every wave working for four rays at a time (march_coop_tail with sample recording): a training batch is a few
// thousand rays, i.e. 64 waves at one ray per lane on 1024 SIMDs, each paced by its longest ray at ~1.3 us per probe -- one probe per sample
// inside an object.  Here a ray's next 16 (then 32, 64 as its neighbours finish) lattice points are probed at once, so the dense interior
// costs one batch per 16-64 samples, and the batch is spread over N / 4 waves.  Same counts and the same t_store as k_march_train_count
// (the probes are the same march_probe() calls; tests/test_gpu_ops.py compares both with the oracle).  The per-chunk sums k_scan_block_sums
// wants are formed by k_chunk_sums.
// NR rays per wave: 1 for training-sized batches (4096 rays: one wave per ray, windows of 64 lattice points; 169 -> 149 us slab, 254 -> 218 us lego),
// 4 above (40 000 rays, constant step: 444 us against 505 with one ray per wave -- enough waves either way, fewer of them to schedule)
template <bool MIP, bool POW2, int NR>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) k_march_train_count_coop(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, MarchParams p, uint32_t N,
    const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ noises, int32_t* __restrict__ counts /* [N] */,
    const uint32_t* __restrict__ mip, float* __restrict__ t_store) {
    __shared__ CoopSharedT<NR> coop[kBlock / PNR_WAVE];
    const uint32_t* mip_lds = stage_mip(mip, MIP ? p.mip_words : 0);
    const uint32_t wave = threadIdx.x / PNR_WAVE, lane = threadIdx.x & (PNR_WAVE - 1);
    const uint32_t n = (blockIdx.x * (kBlock / PNR_WAVE) + wave) * NR + lane;
    const bool keep = lane < (uint32_t)NR && n < N;
    RayCtx c = {};
    float t = 0.0f, far = -FLT_MAX;
    if (keep) {
        ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, p, grid, mip_lds);
        const BoxHit bh = clip_to_box(c, fars[n]);
        far = bh.far;
        t = nears[n];
        t = fmaf(clampf(t * c.dt_gamma, c.dt_min, c.dt_max), noises[n], t);
        t = skip_to_box<MIP && POW2>(c, bh, t);
    }
    const bool active = keep && t < far;
    uint32_t steps = 0;
    if (__ballot(active) != 0ull)
        steps = march_coop_tail<MIP, POW2, true>(coop[wave], p, grid, mip_lds, p.max_steps, active, c, t, far, t, n, 0u, nullptr, nullptr, nullptr, t_store);
    if (keep) counts[n] = (int32_t)steps;
}
__global__ void __launch_bounds__(kBlock) k_chunk_sums(int32_t* __restrict__ scratch, uint32_t N) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    int total;
    (void)block_exclusive_scan(n < N ? scratch[kScanHdr + gridDim.x + n] : 0, &total);
    if (threadIdx.x == 0) scratch[kScanHdr + blockIdx.x] = total;
}

template <bool MIP, bool POW2>
__global__ void __launch_bounds__(kBlock) k_march_train_write(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, MarchParams p, uint32_t N, uint32_t M,
    const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ noises, float* __restrict__ xyzs,
    float* __restrict__ dirs, float* __restrict__ deltas, int32_t* __restrict__ rays, const int32_t* __restrict__ scratch,
    const uint32_t* __restrict__ mip) {
    const uint32_t* mip_lds = stage_mip(mip, MIP ? p.mip_words : 0);
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const int num_steps = n < N ? scratch[kScanHdr + gridDim.x + n] : 0;
    int total;
    const int excl = block_exclusive_scan(num_steps, &total);
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)(scratch[0] + scratch[kScanHdr + blockIdx.x] + excl);
    const uint32_t ray_index = (uint32_t)scratch[1] + n;
    // rays rows: (ray id, offset, count).  Callers zero the counter first (nerf/renderer.py:285-286),
    // so row index == ray id; rows that a non-zero counter[1] would push past N are dropped
    // (the reference would write out of bounds there).
    if (ray_index < N) {
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = num_steps;
    }
    if (num_steps == 0) return;
    if (point_index + (uint32_t)num_steps > M) return;

    RayCtx c;
    ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, p, grid, mip_lds);
    const BoxHit bh = clip_to_box(c, fars[n]);
    const float far = bh.far;
    float t = nears[n];
    t = fmaf(clampf(t * c.dt_gamma, c.dt_min, c.dt_max), noises[n], t);
    float* px = xyzs + (size_t)point_index * 3;
    float* pd = dirs + (size_t)point_index * 3;
    float* pl = deltas + (size_t)point_index * 2;
    float last_t = t, x, y, z, dt;   // delta[1] of the first sample spans the skipped empty space too (raymarching.cu:428)
    t = skip_to_box<MIP && POW2>(c, bh, t);
    int step = 0;
    while (t < far && step < num_steps) {
        if (march_probe<MIP, POW2>(c, t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
        }
    }
}

// Second pass when the count pass kept every sample's ray parameter (t_store): no second walk through the occupancy grid -- a sample's
// position, step and deltas are functions of its t alone (march_probe's first four lines; t_after = t + dt as the walk does), so 16
// lanes per ray write the rows side by side.  The serial pass is paced by each wave's longest ray (~105 us on a 4096-ray batch); this
// one is a streaming write.  Same values bit for bit as k_march_train_write.
__global__ void __launch_bounds__(kBlock) k_march_train_emit(const float* __restrict__ rays_o, const float* __restrict__ rays_d, MarchParams p, uint32_t N,
                                                             uint32_t M, const float* __restrict__ nears, const float* __restrict__ noises,
                                                             const float* __restrict__ t_store, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                             float* __restrict__ deltas, int32_t* __restrict__ rays, const int32_t* __restrict__ scratch) {
    __shared__ uint32_t s_point[kBlock];
    __shared__ int s_steps[kBlock];
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const int num_steps = n < N ? scratch[kScanHdr + gridDim.x + n] : 0;
    int total;
    const int excl = block_exclusive_scan(num_steps, &total);
    const uint32_t point_index = (uint32_t)(scratch[0] + scratch[kScanHdr + blockIdx.x] + excl);
    const uint32_t ray_index = (uint32_t)scratch[1] + n;
    if (blockIdx.y == 0 && n < N && ray_index < N) {
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = num_steps;
    }
    s_point[threadIdx.x] = point_index;
    s_steps[threadIdx.x] = (n < N && point_index + (uint32_t)num_steps <= M) ? num_steps : 0;   // a ray whose rows would not fit is dropped (raymarching.cu:419)
    __syncthreads();
    // 16 workgroups (blockIdx.y) share a 256-ray chunk: each repeats the cheap offset scan above and writes the rows of 16 of its rays
    const uint32_t q = threadIdx.x & 15u;
    {
        const uint32_t r = blockIdx.y * (kBlock / 16) + (threadIdx.x >> 4);
        const uint32_t ray = blockIdx.x * kBlock + r;
        const int steps = s_steps[r];
        if (ray >= N || steps == 0) return;
        const float ox = rays_o[(size_t)ray * 3], oy = rays_o[(size_t)ray * 3 + 1], oz = rays_o[(size_t)ray * 3 + 2];
        const float dx = rays_d[(size_t)ray * 3], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
        float t_first = nears[ray];
        t_first = fmaf(clampf(t_first * p.dt_gamma, p.dt_min, p.dt_max), noises[ray], t_first);   // where the walk started: delta[1] of sample 0 spans the skipped space
        const float* ts = t_store + (size_t)ray * p.max_steps;
        const size_t row0 = s_point[r];
        for (uint32_t k = q; k < (uint32_t)steps; k += 16) {
            const float t0 = ts[k];
            const float dt = clampf(t0 * p.dt_gamma, p.dt_min, p.dt_max);
            float last_t = t_first;
            if (k > 0) { const float tp = ts[k - 1]; last_t = tp + clampf(tp * p.dt_gamma, p.dt_min, p.dt_max); }
            const float t_after = t0 + dt;
            float* px = xyzs + (row0 + k) * 3;
            float* pd = dirs + (row0 + k) * 3;
            float* pl = deltas + (row0 + k) * 2;
            px[0] = clampf(fmaf(t0, dx, ox), -p.bound, p.bound);
            px[1] = clampf(fmaf(t0, dy, oy), -p.bound, p.bound);
            px[2] = clampf(fmaf(t0, dz, oz), -p.bound, p.bound);
            pd[0] = dx; pd[1] = dy; pd[2] = dz;
            pl[0] = dt; pl[1] = t_after - last_t;
        }
    }
}

// reference raymarching.cu:848-882
__global__ void __launch_bounds__(kBlock) k_spread_ray_to_sample(const float* __restrict__ input, const int32_t* __restrict__ rays,
                                                                 uint32_t M, uint32_t N, uint32_t n_channel, float* __restrict__ output) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0) return;
    const float* in = input + (size_t)index * n_channel;
    for (uint32_t step = 0; step < num_steps && offset + step < M; step++) {
        float* out = output + (size_t)(offset + step) * n_channel;
        for (uint32_t i = 0; i < n_channel; i++) out[i] = in[i];
    }
}

// ------------------------------------------------------------------------------------------
// inference march (reference raymarching.cu:907-1011)
// ------------------------------------------------------------------------------------------
#ifndef PNR_OP_MARCH_COOP
#define PNR_OP_MARCH_COOP 1   // 0: the plain SIMT loop (every lane walks its own ray to the end) -- the A/B of the cooperative tail in the drop-in kernel
#endif
// A launch lasts as long as its slowest ray (~1.3 us per probe of one lane while the other 63 idle).  Once at most kCoopRays rays of a wave
// are still marching the whole wave works for them (march_coop_tail, march_core.hpp: the same march_probe() at the ray's next lattice
// points, so every sample is bit for bit what the one-lane walk writes).  Lanes without a ray stay in the loop for that reason.
template <bool MIP, bool POW2>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) k_march_rays(
    uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, MarchParams p, const uint8_t* __restrict__ grid,
    const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
    const float* __restrict__ noises, const uint32_t* __restrict__ mip, uint32_t fill_rows) {
    __shared__ CoopShared coop[kBlock / PNR_WAVE];
    const uint32_t* mip_lds = stage_mip(mip, MIP ? p.mip_words : 0);
    const uint32_t wave = threadIdx.x / PNR_WAVE;
    // fill_rows != 0 (pnr_march_rays_fill): the buffers arrive uninitialised -- every slot a ray leaves unfilled and the alignment rows
    // [n_alive * n_step, fill_rows) are zeroed here, so the caller's three zero-fill launches (raymarching.py:384-386) are not needed
    if (fill_rows)
        for (uint32_t r = n_alive * n_step + blockIdx.x * kBlock + threadIdx.x; r < fill_rows; r += gridDim.x * kBlock) {
            xyzs[(size_t)r * 3] = 0.0f; xyzs[(size_t)r * 3 + 1] = 0.0f; xyzs[(size_t)r * 3 + 2] = 0.0f;
            dirs[(size_t)r * 3] = 0.0f; dirs[(size_t)r * 3 + 1] = 0.0f; dirs[(size_t)r * 3 + 2] = 0.0f;
            deltas[(size_t)r * 2] = 0.0f; deltas[(size_t)r * 2 + 1] = 0.0f;
        }
    for (uint32_t base = blockIdx.x * kBlock; base < n_alive; base += gridDim.x * kBlock) {   // (workgroup-uniform trip count)
        const uint32_t n = base + threadIdx.x;
        const bool keep = n < n_alive;
        RayCtx c = {};
        float t = 0.0f, far = -FLT_MAX, last_t = 0.0f;
        uint32_t step = 0;
        if (keep) {
            const int index = rays_alive[n];
            ctx_init(c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, p, grid, mip_lds);
            t = rays_t[index];
            const BoxHit bh = clip_to_box(c, fars[index]);
            far = bh.far;
            t = fmaf(clampf(t * c.dt_gamma, c.dt_min, c.dt_max), noises ? noises[n] : 0.0f, t);  // noise is slot-indexed (quirk 5)
            last_t = t;
            t = skip_to_box<MIP && POW2>(c, bh, t);
        }
        bool active = keep && t < far && n_step != 0;
        for (;;) {
            const unsigned long long am = __ballot(active);
            if (am == 0ull) break;
            if (PNR_OP_MARCH_COOP && p.coop && __popcll(am) <= kCoopRays) break;
            if (active) {
                float x, y, z, dt;
                if (march_probe<MIP, POW2>(c, t, x, y, z, dt)) {
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
        if (PNR_OP_MARCH_COOP && p.coop && __ballot(active) != 0ull)
            step = march_coop_tail<MIP, POW2>(coop[wave], p, grid, mip_lds, n_step, active, c, t, far, last_t, n, step, xyzs, dirs, deltas);
        if (fill_rows && keep)
            for (; step < n_step; step++) {
                const size_t row = (size_t)n * n_step + step;
                float* px = xyzs + row * 3;
                float* pd = dirs + row * 3;
                float* pl = deltas + row * 2;
                px[0] = 0.0f; px[1] = 0.0f; px[2] = 0.0f; pd[0] = 0.0f; pd[1] = 0.0f; pd[2] = 0.0f; pl[0] = 0.0f; pl[1] = 0.0f;
            }
    }
}

// ------------------------------------------------------------------------------------------
// stable compaction of the alive list (replaces the host boolean mask, nerf/renderer.py:376)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_alive_count(uint32_t n, const int32_t* __restrict__ alive, int32_t* __restrict__ scratch) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const int keep = (i < n && alive[i] >= 0) ? 1 : 0;
    // one ballot + popcount per wave, then 4 partials per block
    __shared__ int wsum[kBlock / PNR_WAVE];
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & (PNR_WAVE - 1)) == 0) wsum[threadIdx.x / PNR_WAVE] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < (int)(kBlock / PNR_WAVE); w++) t += wsum[w];
        scratch[kScanHdr + blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(kBlock) k_alive_write(uint32_t n, const int32_t* __restrict__ alive_in, int32_t* __restrict__ alive_out,
                                                        const int32_t* __restrict__ scratch) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const int id = i < n ? alive_in[i] : -1;
    const int keep = id >= 0 ? 1 : 0;
    __shared__ int wsum[kBlock / PNR_WAVE];
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const int rank = __popcll(m & ((1ull << lane) - 1ull));  // stable in-wave rank
    if (keep) alive_out[scratch[kScanHdr + blockIdx.x] + woff + rank] = id;
}

// single workgroup: reduce the 'any' mask to the world-space box of occupied bricks (+2 cells per cascade)
__global__ void __launch_bounds__(1024) k_mip_bounds(uint32_t* __restrict__ mip, uint32_t words_per_mask, uint32_t nbricks, uint32_t C, uint32_t H,
                                                     float bound) {
    __shared__ float red[6][1024 / PNR_WAVE];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const uint32_t bricks_per_level = nbricks / C;
    for (uint32_t wd = threadIdx.x; wd < words_per_mask; wd += blockDim.x) {
        uint32_t bits = mip[wd];
        while (bits) {
            const uint32_t b = wd * 32 + (uint32_t)__ffs((int)bits) - 1;
            bits &= bits - 1;
            if (b >= nbricks) break;
            const uint32_t level = b / bricks_per_level, local = b % bricks_per_level;
            const float mb = fminf(scalbnf(1.0f, (int)level), bound);
            const float cell = 2.0f * mb / (float)H;
            const uint32_t bc[3] = {gather3(local), gather3(local >> 1), gather3(local >> 2)};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                lo[a] = fminf(lo[a], ((float)(4 * bc[a]) / (float)H * 2.0f - 1.0f) * mb - 2.0f * cell);
                hi[a] = fmaxf(hi[a], ((float)(4 * bc[a] + 4) / (float)H * 2.0f - 1.0f) * mb + 2.0f * cell);
            }
        }
    }
    const int lane = threadIdx.x & (PNR_WAVE - 1), wave = threadIdx.x / PNR_WAVE;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = lo[a], h = hi[a];
        for (int off = PNR_WAVE / 2; off > 0; off >>= 1) { l = fminf(l, __shfl_xor(l, off, PNR_WAVE)); h = fmaxf(h, __shfl_xor(h, off, PNR_WAVE)); }
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    // Power-of-two bound and H: every cascade's cell grid refines the coarsest one (cell = 2 bound / H), so faces rounded
    // outward to that grid are cell boundaries of ALL cascades -- the property skip_to_box() needs.  [6] = 1 flags it,
    // [7] = the coarse cell size.  Otherwise the box is left as is and only clips the far end.
    const bool aligned = (H & (H - 1)) == 0 && (__float_as_uint(bound) & 0x7fffffu) == 0 && bound > 0.0f;
    const float coarse = 2.0f * bound / (float)H;
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < (int)(blockDim.x / PNR_WAVE); w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        if (aligned && fabsf(v) <= 2.0f * bound)   // (an empty grid keeps its +-FLT_MAX sentinels)
            v = (threadIdx.x < 3 ? floorf((v + bound) / coarse) : ceilf((v + bound) / coarse)) * coarse - bound;  // exact: powers of two
        reinterpret_cast<float*>(mip + 2 * words_per_mask)[threadIdx.x] = v;  // empty grid: min = +FLT_MAX > max = -FLT_MAX => every ray "misses"
    }
    if (threadIdx.x == 6) mip[2 * words_per_mask + 6] = aligned ? 1u : 0u;
    if (threadIdx.x == 7) reinterpret_cast<float*>(mip + 2 * words_per_mask)[7] = coarse;
}

// nerf/utils.py:53-149, deterministic core (pixel ids -> rays)
// four values per thread: one 16-byte load, one 4-byte store
__global__ void __launch_bounds__(kBlock) k_image_to_uint8(const float* __restrict__ src, uint64_t n, int to_srgb, uint8_t* __restrict__ dst) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i0 >= n) return;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(src + i0) & 15u) == 0) {
        const float4 q = *reinterpret_cast<const float4*>(src + i0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        for (int k = 0; k < 4; k++) if (i0 + k < n) v[k] = src[i0 + k];
    }
    uint8_t b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float x = v[k];
        if (to_srgb) x = x < 0.0031308f ? 12.92f * x : 1.055f * powf(x, 0.41666f) - 0.055f;
        b[k] = (uint8_t)(int)(x * 255.0f);
    }
    if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(dst + i0) & 3u) == 0)
        *reinterpret_cast<uchar4*>(dst + i0) = make_uchar4(b[0], b[1], b[2], b[3]);
    else
        for (int k = 0; k < 4; k++) if (i0 + k < n) dst[i0 + k] = b[k];
}

__global__ void __launch_bounds__(kBlock) k_get_rays(const float* __restrict__ poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t W,
                                                     const long long* __restrict__ inds, uint32_t N, float* __restrict__ rays_o,
                                                     float* __restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const uint32_t p = inds ? (uint32_t)inds[(size_t)b * N + n] : n;
    const float xs = (((float)(p % W) + 0.5f) - cx) / fx;
    const float ys = (((float)(p / W) + 0.5f) - cy) / fy;
    const float nrm = sqrtf(fmaf(xs, xs, fmaf(ys, ys, 1.0f)));
    const float d0 = xs / nrm, d1 = ys / nrm, d2 = 1.0f / nrm;
    const float* P = poses + (size_t)b * 16;
    float* o = rays_o + ((size_t)b * N + n) * 3;
    float* d = rays_d + ((size_t)b * N + n) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d[k] = fmaf(d2, P[k * 4 + 2], fmaf(d1, P[k * 4 + 1], d0 * P[k * 4 + 0]));
        o[k] = P[k * 4 + 3];
    }
}

}  // namespace pnr

// ==========================================================================================
// C ABI
// ==========================================================================================
int g_opt_block_skip = getenv("PNR_NO_BLOCK_SKIP") ? 0 : 1;
int g_opt_coop_march = getenv("PNR_NO_COOP_MARCH") ? 0 : 1;
int g_opt_hosted_tail = getenv("PNR_NO_HOSTED_TAIL") ? 0 : 1;   // frame loops: march stragglers finished inside the lookup launch (frame.hip: hosted_march_tail)
int g_opt_march_budget = getenv("PNR_MARCH_BUDGET") ? atoi(getenv("PNR_MARCH_BUDGET")) : 2;     // probe rounds a later march launch spends on a ray before it hands it over
int g_opt_march_budget0 = getenv("PNR_MARCH_BUDGET0") ? atoi(getenv("PNR_MARCH_BUDGET0")) : 0;  // the same for a frame's first launch; 0 = that launch keeps the in-wave cooperative tail
int g_opt_march_blocks = getenv("PNR_MARCH_BLOCKS") ? atoi(getenv("PNR_MARCH_BLOCKS")) : 0;    // workgroup cap of a budgeted march launch; 0 (or 65536) = automatic: the 1 280 resident workgroups once a frame has twice as many chunks (frame.hip)
int g_opt_aux_fusion = getenv("PNR_NO_AUX_FUSION") ? 0 : 1;
int g_opt_composite_fusion = getenv("PNR_NO_COMPOSITE_FUSION") ? 0 : (getenv("PNR_COMPOSITE_FUSION") ? atoi(getenv("PNR_COMPOSITE_FUSION")) : 2);   // 2: the NeRF frame loop has no composite launch
int g_opt_iteration_margin = getenv("PNR_ITERATION_MARGIN") ? atoi(getenv("PNR_ITERATION_MARGIN")) : 0;   // measured 0 / 1 / 2 / 4 on the moving-camera bench: 4.239 / 4.247 / 4.277 / 4.265 ms -- a look costs less than a spare iteration
int g_opt_palette_waves12 = getenv("PNR_PALETTE_WAVES8") ? 0 : 1;   // specialised PaletteNeRF field kernel: 12-wave workgroups (three waves per SIMD)
int g_opt_dynamic_tiles = getenv("PNR_DYNAMIC_TILES") ? 1 : 0;   // measured: garden 14.6 -> 20.2 ms with it on (one contended counter, scattered tiles): off
int g_opt_grid_fast = getenv("PNR_NO_GRID_FAST") ? 0 : 1;   // pnr_grid_encode_forward: the D = 3, C = 2 kernel (k_grid_fwd_d3c2) instead of the generic one (A/B; same bits)
int g_opt_train_coop = getenv("PNR_NO_TRAIN_COOP") ? 0 : 1;   // training march: wave-cooperative counting pass (k_march_train_count_coop)
int g_opt_mlp_f16x3 = getenv("PNR_MLP_FP32") ? 0 : 1;   // training MLP launches (mlp.hip): split-fp16 products on the fp16 matrix pipe (1) / exact fp32 MFMA (0)
int g_opt_coarse_image = getenv("PNR_NO_COARSE_IMAGE") ? 0 : 1;   // binned table gradient: the coarsest levels accumulated as LDS images (grid_binned.hip: k_coarse_image)
int g_opt_scatter_staged = getenv("PNR_NO_SCATTER_STAGED") ? 0 : 1;   // binned table gradient: records ordered by bucket in LDS and written coalesced (grid_binned.hip: k_bin_scatter_staged)
int g_opt_cell_merge = getenv("PNR_NO_CELL_MERGE") ? 0 : 1;   // binned table gradient: mid levels merge the samples of one cell before writing records
int g_opt_flex_coop = getenv("PNR_NO_FLEX_COOP") ? 0 : 1;   // composite_rays_flex (n_step <= 8): the workgroup-cooperative kernel (coalesced rows) instead of one thread per ray; same bits
int g_opt_grid_nt = getenv("PNR_GRID_NT") ? atoi(getenv("PNR_GRID_NT")) : 0;   // experiment: non-temporal stores (1) / input loads (2) in k_grid_fwd_d3c2
int g_opt_adam_variant = 0;   // experiment switch of adam.hip (which multiply-adds are contracted); 0 = torch's kernels on this platform

using namespace pnr;

extern "C" {

int pnr_abi_version(void) { return 7; }

int pnr_set_option(const char* name, int value) {
    if (!name) return PNR_ERR_INVALID;
    if (!strcmp(name, "block_skip")) { g_opt_block_skip = value != 0; return PNR_OK; }
    if (!strcmp(name, "coop_march")) { g_opt_coop_march = value != 0; return PNR_OK; }
    if (!strcmp(name, "hosted_tail")) { g_opt_hosted_tail = value != 0; return PNR_OK; }
    if (!strcmp(name, "march_budget")) { g_opt_march_budget = value < 1 ? 1 : (value > 1024 ? 1024 : value); return PNR_OK; }
    if (!strcmp(name, "march_budget0")) { g_opt_march_budget0 = value < 0 ? 0 : (value > 1024 ? 1024 : value); return PNR_OK; }
    if (!strcmp(name, "march_blocks")) { g_opt_march_blocks = value < 0 ? 0 : value; return PNR_OK; }
    if (!strcmp(name, "aux_fusion")) { g_opt_aux_fusion = value != 0; return PNR_OK; }
    if (!strcmp(name, "composite_fusion")) { g_opt_composite_fusion = value < 0 ? 0 : (value > 2 ? 2 : value); return PNR_OK; }
    if (!strcmp(name, "palette_waves12")) { g_opt_palette_waves12 = value != 0; return PNR_OK; }
    if (!strcmp(name, "dynamic_tiles")) { g_opt_dynamic_tiles = value != 0; return PNR_OK; }
    if (!strcmp(name, "grid_fast")) { g_opt_grid_fast = value != 0; return PNR_OK; }
    if (!strcmp(name, "train_coop")) { g_opt_train_coop = value != 0; return PNR_OK; }
    if (!strcmp(name, "mlp_f16x3")) { g_opt_mlp_f16x3 = value != 0; return PNR_OK; }
    if (!strcmp(name, "coarse_image")) { g_opt_coarse_image = value != 0; return PNR_OK; }
    if (!strcmp(name, "cell_merge")) { g_opt_cell_merge = value != 0; return PNR_OK; }
    if (!strcmp(name, "scatter_staged")) { g_opt_scatter_staged = value != 0; return PNR_OK; }
    if (!strcmp(name, "flex_coop")) { g_opt_flex_coop = value != 0; return PNR_OK; }
    if (!strcmp(name, "grid_nt")) { g_opt_grid_nt = value & 3; return PNR_OK; }
    if (!strcmp(name, "adam_variant")) { g_opt_adam_variant = value & 7; return PNR_OK; }
    if (!strcmp(name, "iteration_margin")) { g_opt_iteration_margin = value < 0 ? 0 : (value > 64 ? 64 : value); return PNR_OK; }
    return PNR_ERR_INVALID;
}

const char* pnr_error_string(int code) {
    switch (code) {
        case PNR_OK: return "ok";
        case PNR_ERR_INVALID: return "invalid argument (null pointer or bad size)";
        case PNR_ERR_UNSUPPORTED: return "unsupported configuration";
        case PNR_ERR_LAUNCH: return "HIP kernel launch failed";
        case PNR_ERR_ALIGNMENT: return "an activation array does not start on a 16-byte boundary or holds fewer than four floats (pnr_mlp_* move tiles as 16-byte requests: copy such a view)";
        default: return "unknown error";
    }
}

int pnr_get_rays(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t* inds, uint32_t N,
                 float* rays_o, float* rays_d, pnr_stream_t stream) {
    if (B == 0 || N == 0) return PNR_OK;
    if (!poses || !rays_o || !rays_d || H == 0 || W == 0 || fx == 0.0f || fy == 0.0f) return PNR_ERR_INVALID;
    if (!inds && N != H * W) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_get_rays, dim3(cdiv(N, kBlock), B), dim3(kBlock), 0, as_stream(stream), poses, B, fx, fy, cx, cy, W,
                       reinterpret_cast<const long long*>(inds), N, rays_o, rays_d);
    return check_launch();
}

int pnr_image_to_uint8(const float* src, uint64_t n, int linear_to_srgb, uint8_t* dst, pnr_stream_t stream) {
    if (n == 0) return PNR_OK;
    if (!src || !dst) return PNR_ERR_INVALID;
    const uint64_t blocks = (n + kBlock * 4 - 1) / (kBlock * 4);
    if (blocks > 0x7fffffffull) return PNR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_image_to_uint8, dim3((uint32_t)blocks), dim3(kBlock), 0, as_stream(stream), src, n, linear_to_srgb, dst);
    return check_launch();
}

uint64_t pnr_scan_scratch_bytes(uint32_t N) { return ((uint64_t)kScanHdr + cdiv(N, kBlock) + N + 4) * 4; }

int pnr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears,
                           float* fars, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!rays_o || !rays_d || !aabb || !nears || !fars) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_near_far, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch();
}

int pnr_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!rays_o || !rays_d || !coords) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_sph_from_ray, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, radius, N, coords);
    return check_launch();
}

int pnr_morton3d(const int32_t* coords, uint32_t N, int32_t* indices, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!coords || !indices) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_morton3d, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), coords, N, indices);
    return check_launch();
}

int pnr_morton3d_invert(const int32_t* indices, uint32_t N, int32_t* coords, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!coords || !indices) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_morton3d_invert, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), indices, N, coords);
    return check_launch();
}

int pnr_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!grid || !bitfield) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_packbits, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), grid, N, density_thresh, bitfield);
    return check_launch();
}

uint64_t pnr_occupancy_mip_bytes(uint32_t C, uint32_t H) {
    return ((uint64_t)2 * (((uint64_t)C * H * H * H / 64 + 31) / 32) + 8) * 4;  // any mask, all mask, occupied box
}

int pnr_build_occupancy_mip(const uint8_t* grid, uint32_t C, uint32_t H, float bound, void* mip, pnr_stream_t stream) {
    if (!grid || !mip) return PNR_ERR_INVALID;
    if (C == 0 || C > 16 || H == 0 || (H % 4) != 0 || (reinterpret_cast<uintptr_t>(grid) & 7)) return PNR_ERR_UNSUPPORTED;
    const uint32_t nbricks = (uint32_t)((uint64_t)C * H * H * H / 64);
    const uint32_t words = (nbricks + 31) / 32;
    hipLaunchKernelGGL(k_build_mip, dim3(cdiv(nbricks, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const unsigned long long*>(grid),
                       nbricks, words, static_cast<uint32_t*>(mip));
    hipLaunchKernelGGL(k_mip_bounds, dim3(1), dim3(1024), 0, as_stream(stream), static_cast<uint32_t*>(mip), words, nbricks, C, H, bound);
    return check_launch();
}

// LDS needed by a march launch that uses the mip; 0 when the mip is absent or does not fit (then the plain path runs)
static uint32_t mip_lds_bytes(const MarchParams& p) { return p.mip_words ? (2 * p.mip_words + 8) * 4 : 0; }
static bool mip_usable(const void* mip, uint32_t C, uint32_t H) {
    return mip && (H % 4) == 0 && pnr_occupancy_mip_bytes(C, H) <= 64 * 1024;
}

int pnr_march_rays_train_mip(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                             uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                             float* deltas, int32_t* rays, int32_t* counter, const float* noises, void* scratch, const void* mip,
                             float* t_store, pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !rays || !counter || !noises || !scratch) return PNR_ERR_INVALID;
    if (M > 0 && (!xyzs || !dirs || !deltas)) return PNR_ERR_INVALID;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    int32_t* sc = static_cast<int32_t*>(scratch);
    const uint32_t nb = cdiv(N, kBlock);
    const bool use_mip = mip_usable(mip, C, H);
    const bool pow2 = is_pow2f(bound) && (H & (H - 1)) == 0;
    const MarchParams p = make_march_params(bound, dt_gamma, max_steps, C, H, use_mip);
    const uint32_t lds = mip_lds_bytes(p);
    const uint32_t* m = static_cast<const uint32_t*>(mip);
    // measured (profiles/scratch/train_coop_ab.py, us per call incl. the emit pass, one ray per lane -> cooperative): 4096 rays 206 -> 162 (slab, dt_gamma 1/128),
    // 726 -> 256 (lego, 0), 507 -> 235 (slab, 0); 40 000 rays 704 -> 441 and 760 -> 695 with a constant step, but 247 -> 300 and 232 -> 239 with a growing one
    // (its windows are walked point by point and there are enough waves already): cooperative for training-sized batches or a constant step
    const bool coop_count = p.coop != 0 && g_opt_train_coop != 0 && (uint64_t)N * max_steps < (1ull << 31) && (dt_gamma == 0.0f || N <= 16384u);
#define PNR_LAUNCH_TRAIN(MIPV, P2V)                                                                                                           \
    if (coop_count) {                                                                                                                          \
        if (N <= 8192u)                                                                                                                        \
            hipLaunchKernelGGL((k_march_train_count_coop<MIPV, P2V, 1>), dim3(cdiv(N, kBlock / PNR_WAVE)), dim3(kBlock), lds, s, rays_o,       \
                               rays_d, grid, p, N, nears, fars, noises, sc + kScanHdr + nb, m, t_store);                                       \
        else                                                                                                                                   \
            hipLaunchKernelGGL((k_march_train_count_coop<MIPV, P2V, 4>), dim3(cdiv(N, (kBlock / PNR_WAVE) * 4)), dim3(kBlock), lds, s, rays_o, \
                               rays_d, grid, p, N, nears, fars, noises, sc + kScanHdr + nb, m, t_store);                                       \
        hipLaunchKernelGGL(k_chunk_sums, dim3(nb), dim3(kBlock), 0, s, sc, N);                                                                 \
    } else                                                                                                                                     \
        hipLaunchKernelGGL((k_march_train_count<MIPV, P2V>), dim3(nb), dim3(kBlock), lds, s, rays_o, rays_d, grid, p, N, nears, fars, noises, sc, m, \
                           t_store);                                                                                                           \
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, sc, nb, counter, N, (int32_t*)nullptr);                                    \
    if (t_store)                                                                                                                               \
        hipLaunchKernelGGL(k_march_train_emit, dim3(nb, 16), dim3(kBlock), 0, s, rays_o, rays_d, p, N, M, nears, noises, t_store, xyzs, dirs, deltas, \
                           rays, sc);                                                                                                          \
    else                                                                                                                                       \
        hipLaunchKernelGGL((k_march_train_write<MIPV, P2V>), dim3(nb), dim3(kBlock), lds, s, rays_o, rays_d, grid, p, N, M, nears, fars, noises, \
                           xyzs, dirs, deltas, rays, sc, m)
    if (use_mip && pow2) { PNR_LAUNCH_TRAIN(true, true); }
    else if (use_mip) { PNR_LAUNCH_TRAIN(true, false); }
    else if (pow2) { PNR_LAUNCH_TRAIN(false, true); }
    else { PNR_LAUNCH_TRAIN(false, false); }
#undef PNR_LAUNCH_TRAIN
    return check_launch();
}

int pnr_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs,
                         float* dirs, float* deltas, int32_t* rays, int32_t* counter, const float* noises, void* scratch,
                         pnr_stream_t stream) {
    return pnr_march_rays_train_mip(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                                    noises, scratch, nullptr, nullptr, stream);
}

int pnr_spread_ray_to_sample(const float* input, const int32_t* rays, uint32_t M, uint32_t N, uint32_t n_channel, float* output,
                             pnr_stream_t stream) {
    if (N == 0 || M == 0) return PNR_OK;
    if (n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
    if (!input || !rays || !output) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_spread_ray_to_sample, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), input, rays, M, N, n_channel, output);
    return check_launch();
}

int pnr_march_rays_mip(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                       const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                       const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises, const void* mip,
                       pnr_stream_t stream) {
    return pnr_march_rays_fill(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises, mip, 0, stream);
}

int pnr_march_rays_fill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                        const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                        const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises, const void* mip,
                        uint32_t fill_rows, pnr_stream_t stream) {
    (void)nears;
    if (fill_rows != 0 && fill_rows < n_alive * n_step) return PNR_ERR_INVALID;
    if ((n_alive == 0 || n_step == 0) && fill_rows == 0) return PNR_OK;
    if (n_alive != 0 && (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars)) return PNR_ERR_INVALID;
    if (!xyzs || !dirs || !deltas) return PNR_ERR_INVALID;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0) return PNR_ERR_INVALID;
    const bool use_mip = mip_usable(mip, C, H);
    const bool pow2 = is_pow2f(bound) && (H & (H - 1)) == 0;
    const MarchParams p = make_march_params(bound, dt_gamma, max_steps, C, H, use_mip);
    const uint32_t lds = mip_lds_bytes(p);
    const uint32_t nb = cdiv(n_alive ? n_alive : 1u, kBlock);
    const uint32_t grid_dim = use_mip ? (nb < 2048u ? nb : 2048u) : nb;  // with the mip staged per workgroup, keep workgroups persistent
    const dim3 g(grid_dim), b(kBlock);
    hipStream_t s = as_stream(stream);
    const uint32_t* m = static_cast<const uint32_t*>(mip);
    if (use_mip && pow2)
        hipLaunchKernelGGL((k_march_rays<true, true>), g, b, lds, s, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, p, grid, fars, xyzs, dirs, deltas, noises, m, fill_rows);
    else if (use_mip)
        hipLaunchKernelGGL((k_march_rays<true, false>), g, b, lds, s, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, p, grid, fars, xyzs, dirs, deltas, noises, m, fill_rows);
    else if (pow2)
        hipLaunchKernelGGL((k_march_rays<false, true>), g, b, lds, s, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, p, grid, fars, xyzs, dirs, deltas, noises, m, fill_rows);
    else
        hipLaunchKernelGGL((k_march_rays<false, false>), g, b, lds, s, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, p, grid, fars, xyzs, dirs, deltas, noises, m, fill_rows);
    return check_launch();
}

int pnr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                   const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                   const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises,
                   pnr_stream_t stream) {
    if (n_alive != 0 && n_step != 0 && !noises) return PNR_ERR_INVALID;
    return pnr_march_rays_mip(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs,
                              deltas, noises, nullptr, stream);
}

int pnr_compact_alive(uint32_t n_alive, const int32_t* rays_alive_in, int32_t* rays_alive_out, int32_t* n_alive_out, void* scratch,
                      pnr_stream_t stream) {
    if (!n_alive_out || !scratch) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    int32_t* sc = static_cast<int32_t*>(scratch);
    if (n_alive == 0) { return hipMemsetAsync(n_alive_out, 0, sizeof(int32_t), s) == hipSuccess ? PNR_OK : PNR_ERR_LAUNCH; }
    if (!rays_alive_in || !rays_alive_out) return PNR_ERR_INVALID;
    const uint32_t nb = cdiv(n_alive, kBlock);
    hipLaunchKernelGGL(k_alive_count, dim3(nb), dim3(kBlock), 0, s, n_alive, rays_alive_in, sc);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, sc, nb, (int32_t*)nullptr, n_alive, n_alive_out);
    hipLaunchKernelGGL(k_alive_write, dim3(nb), dim3(kBlock), 0, s, n_alive, rays_alive_in, rays_alive_out, sc);
    return check_launch();
}

}  // extern "C"
