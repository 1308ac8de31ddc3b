// raymarch.hip -- occupancy-grid ray marching for gfx950 (MI355X).
//
// What it computes follows the reference's raymarching extension (cited per kernel); how it is
// organised does not: output slots of the training march come from a deterministic two-level
// wave64 prefix sum instead of global atomics, the alive-ray list is compacted on the device with
// the same scan, and all launches take an explicit stream.
#include "pnr_common.hpp"
#include <float.h>

namespace pnr {

constexpr uint32_t kBlock = 256;  // 4 waves; one ray / element per thread

// ------------------------------------------------------------------------------------------
// per-ray constants + the march state machine (reference raymarching.cu:336-349, 362-403)
// ------------------------------------------------------------------------------------------
struct RayCtx {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, fC, fH;
    uint32_t H, H3;
    const uint8_t* __restrict__ grid;
};

__device__ __forceinline__ void ctx_init(RayCtx& c, const float* __restrict__ o, const float* __restrict__ d, float bound,
                                         float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                         const uint8_t* __restrict__ grid) {
    c.ox = o[0]; c.oy = o[1]; c.oz = o[2];
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    c.rdx = 1.0f / c.dx; c.rdy = 1.0f / c.dy; c.rdz = 1.0f / c.dz;
    c.bound = bound; c.dt_gamma = dt_gamma;
    const float two_sqrt3 = 2.0f * 1.7320508075688772f;
    c.dt_min = two_sqrt3 / (float)max_steps;
    c.dt_max = two_sqrt3 * (float)(1 << (C - 1)) / (float)H;
    c.rH = 1.0f / (float)H; c.fC = (float)C; c.fH = (float)H;
    c.H = H; c.H3 = H * H * H; c.grid = grid;
}

// Probe the cell containing the point at parameter t.  Occupied: returns true and the sample
// (x,y,z,dt), t untouched.  Empty: advances t past the cell (do..while of the reference) and
// returns false.
__device__ __forceinline__ bool march_probe(const RayCtx& c, float& t, float& x, float& y, float& z, float& dt) {
    const float t0 = t;
    x = clampf(fmaf(t0, c.dx, c.ox), -c.bound, c.bound);
    y = clampf(fmaf(t0, c.dy, c.oy), -c.bound, c.bound);
    z = clampf(fmaf(t0, c.dz, c.oz), -c.bound, c.bound);
    dt = clampf(t0 * c.dt_gamma, c.dt_min, c.dt_max);
    const int lp = mip_from_pos(x, y, z, c.fC), ld = mip_from_dt(dt, c.fH, c.fC);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), c.bound);
    const float mip_rbound = 1.0f / mip_bound;
    const float hi = (float)(c.H - 1);
    // double intermediate exactly as the reference's `0.5 * (x * mip_rbound + 1) * H`
    const int nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
    const int ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
    const int nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)c.H), 0.0f, hi);
    const uint32_t index = (uint32_t)level * c.H3 + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const bool occ = c.grid[index >> 3] & (1u << (index & 7u));
    if (occ) return true;
    const float tx = fmaf(fmaf(fmaf(0.5f, signf(c.dx), (float)nx + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -x) * c.rdx;
    const float ty = fmaf(fmaf(fmaf(0.5f, signf(c.dy), (float)ny + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -y) * c.rdy;
    const float tz = fmaf(fmaf(fmaf(0.5f, signf(c.dz), (float)nz + 0.5f) * c.rH, 2.0f, -1.0f), mip_bound, -z) * c.rdz;
    const float tt = t0 + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    float tc = t0;
    do { tc += clampf(tc * c.dt_gamma, c.dt_min, c.dt_max); } while (tc < tt);
    t = tc;
    return false;
}

// ------------------------------------------------------------------------------------------
// utils
// ------------------------------------------------------------------------------------------

// reference raymarching.cu:95-148
__global__ void __launch_bounds__(kBlock) k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ aabb, uint32_t N, float min_near,
                                                     float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1.0f / rays_d[n * 3], rdy = 1.0f / rays_d[n * 3 + 1], rdz = 1.0f / rays_d[n * 3 + 2];
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
    if (near > far) { tmp = near; near = far; far = tmp; }
    float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
    if (ny > fy) { tmp = ny; ny = fy; fy = tmp; }
    if (near > fy || ny > far) { nears[n] = fars[n] = FLT_MAX; return; }
    if (ny > near) near = ny;
    if (fy < far) far = fy;
    float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
    if (nz > fz) { tmp = nz; nz = fz; fz = tmp; }
    if (near > fz || nz > far) { nears[n] = fars[n] = FLT_MAX; return; }
    if (nz > near) near = nz;
    if (fz < far) far = fz;
    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
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
__global__ void __launch_bounds__(kBlock) k_march_train_count(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, float bound,
    float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* __restrict__ nears,
    const float* __restrict__ fars, const float* __restrict__ noises, int32_t* __restrict__ scratch) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    int num_steps = 0;
    if (n < N) {
        RayCtx c;
        ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t = nears[n];
        t = fmaf(clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t);
        float x, y, z, dt;
        while (t < far && (uint32_t)num_steps < max_steps) {
            if (march_probe(c, t, x, y, z, dt)) { num_steps++; t += dt; }
        }
        scratch[kScanHdr + gridDim.x + n] = num_steps;  // per-ray counts live behind the block sums
    }
    int total;
    (void)block_exclusive_scan(num_steps, &total);
    if (threadIdx.x == 0) scratch[kScanHdr + blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) k_march_train_write(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, float bound,
    float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* __restrict__ nears,
    const float* __restrict__ fars, const float* __restrict__ noises, float* __restrict__ xyzs, float* __restrict__ dirs,
    float* __restrict__ deltas, int32_t* __restrict__ rays, const int32_t* __restrict__ scratch) {
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
    ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
    const float far = fars[n];
    float t = nears[n];
    t = fmaf(clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t);
    float* px = xyzs + (size_t)point_index * 3;
    float* pd = dirs + (size_t)point_index * 3;
    float* pl = deltas + (size_t)point_index * 2;
    float last_t = t, x, y, z, dt;
    int step = 0;
    while (t < far && step < num_steps) {
        if (march_probe(c, t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
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
__global__ void __launch_bounds__(kBlock) k_march_rays(
    uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
    uint32_t C, uint32_t H, const uint8_t* __restrict__ grid, const float* __restrict__ fars, float* __restrict__ xyzs,
    float* __restrict__ dirs, float* __restrict__ deltas, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    RayCtx c;
    ctx_init(c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
    float* px = xyzs + (size_t)n * n_step * 3;
    float* pd = dirs + (size_t)n * n_step * 3;
    float* pl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    const float far = fars[index];
    t = fmaf(clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t);  // noise is slot-indexed (quirk 5)
    float last_t = t, x, y, z, dt;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        if (march_probe(c, t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
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

}  // namespace pnr

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace pnr;

extern "C" {

int pnr_abi_version(void) { return 1; }

const char* pnr_error_string(int code) {
    switch (code) {
        case PNR_OK: return "ok";
        case PNR_ERR_INVALID: return "invalid argument (null pointer or bad size)";
        case PNR_ERR_UNSUPPORTED: return "unsupported configuration";
        case PNR_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
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

int pnr_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs,
                         float* dirs, float* deltas, int32_t* rays, int32_t* counter, const float* noises, void* scratch,
                         pnr_stream_t stream) {
    if (N == 0) return PNR_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !rays || !counter || !noises || !scratch) return PNR_ERR_INVALID;
    if (M > 0 && (!xyzs || !dirs || !deltas)) return PNR_ERR_INVALID;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    int32_t* sc = static_cast<int32_t*>(scratch);
    const uint32_t nb = cdiv(N, kBlock);
    hipLaunchKernelGGL(k_march_train_count, dim3(nb), dim3(kBlock), 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H,
                       nears, fars, noises, sc);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, sc, nb, counter, N, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_march_train_write, dim3(nb), dim3(kBlock), 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
                       nears, fars, noises, xyzs, dirs, deltas, rays, sc);
    return check_launch();
}

int pnr_spread_ray_to_sample(const float* input, const int32_t* rays, uint32_t M, uint32_t N, uint32_t n_channel, float* output,
                             pnr_stream_t stream) {
    if (N == 0 || M == 0) return PNR_OK;
    if (n_channel > PNR_CHANNEL_MAXIMUM) return PNR_ERR_UNSUPPORTED;
    if (!input || !rays || !output) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_spread_ray_to_sample, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, as_stream(stream), input, rays, M, N, n_channel, output);
    return check_launch();
}

int pnr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                   const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                   const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises,
                   pnr_stream_t stream) {
    (void)nears;
    if (n_alive == 0 || n_step == 0) return PNR_OK;
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars || !xyzs || !dirs || !deltas || !noises) return PNR_ERR_INVALID;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_march_rays, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
    return check_launch();
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
