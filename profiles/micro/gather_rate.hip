// Gather ceiling of MI355X for the hash-grid lookup's access pattern: every lane of a wave reads a different, random row of a table.
//   rows of 4 / 8 / 16 bytes; table of 4 MB (fits one XCD's L2) / 50 MB (the 16-level fp32 hash table) / 512 MB (HBM);
//   `pair` = two loads to adjacent rows (the x, x+1 corners of a cell when x is even).
// Reports lane-loads per clock per CU (2.4 GHz nominal) and GB/s of useful bytes.
// hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <typename T> __device__ __forceinline__ float first(const T& v);
template <> __device__ __forceinline__ float first<float>(const float& v) { return v; }
template <> __device__ __forceinline__ float first<float2>(const float2& v) { return v.x + v.y; }
template <> __device__ __forceinline__ float first<float4>(const float4& v) { return v.x + v.y + v.z + v.w; }

template <typename T, int UNROLL, bool PAIR>
__global__ void __launch_bounds__(256) k(const T* __restrict__ table, uint32_t mask, uint32_t loads_per_lane, uint32_t seed, float* __restrict__ out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    float acc = 0.0f;
    for (uint32_t i = 0; i < loads_per_lane; i += UNROLL) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            s = s * 1664525u + 1013904223u;             // LCG: a fresh random row per load
            uint32_t r = (s >> 4) & mask;
            if (PAIR && (u & 1)) r = (((s - 1013904223u) * 4000846301u >> 4) & mask) ^ 1u;   // the previous load's row ^ 1 (1664525^-1 mod 2^32 = 4000846301)
            v[u] = table[r];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += first<T>(v[u]);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename T, bool PAIR>
void run(const char* name, const void* table, size_t table_bytes, float* out) {
    const uint32_t rows = (uint32_t)(table_bytes / sizeof(T));
    uint32_t mask = 1; while ((mask << 1) <= rows) mask <<= 1; mask -= 1;
    const uint32_t blocks = 256 * 8, loads = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, 8, PAIR>), dim3(blocks), dim3(256), 0, 0, (const T*)table, mask, loads, 1u, out);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<T, 8, PAIR>), dim3(blocks), dim3(256), 0, 0, (const T*)table, mask, loads, 7u + i, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_loads = (double)blocks * 256 * loads * reps;
    const double sec = ms * 1e-3;
    printf("%-34s table %6.0f MB: %7.1f G lane-loads/s = %5.2f per clock per CU, %7.1f GB/s useful\n", name, table_bytes / 1048576.0, lane_loads / sec / 1e9,
           lane_loads / sec / (256 * 2.4e9), lane_loads * sizeof(T) / sec / 1e9);
}

int main() {
    const size_t big = 512ull << 20;
    void* table; float* out;
    (void)hipMalloc(&table, big); (void)hipMalloc(&out, 64);
    (void)hipMemset(table, 0, big);
    for (size_t mb : {4ull, 50ull, 512ull}) {
        const size_t bytes = mb << 20;
        run<float, false>("4-byte rows (half2 row)", table, bytes, out);
        run<float2, false>("8-byte rows (fp32 row)", table, bytes, out);
        run<float2, true>("8-byte rows, every 2nd adjacent", table, bytes, out);
        run<float4, false>("16-byte rows (two tables interleaved)", table, bytes, out);
    }
    // reference point: the lookup kernel of the lego frame moves 335 k samples x 128 corner rows in ~62 us = 0.69 T lane-loads/s
    return 0;
}
