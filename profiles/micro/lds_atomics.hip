// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_f64 to random addresses of a 64 KiB image
// (the access pattern of k_bin_gather, csrc/grid_binned.hip).  hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename T, int WORDS>
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ idx, uint32_t n_per_block, T* __restrict__ out) {
    extern __shared__ unsigned char raw[];
    T* acc = reinterpret_cast<T*>(raw);
    for (uint32_t i = threadIdx.x; i < WORDS; i += 1024) acc[i] = T(0);
    __syncthreads();
    const uint32_t* p = idx + (size_t)blockIdx.x * n_per_block;
    for (uint32_t r = threadIdx.x; r + 7 * 1024 < n_per_block; r += 8 * 1024) {
        uint32_t a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = p[r + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&acc[a[u] % WORDS], T(1));
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[7];
}

template <typename T, int WORDS>
void run(const char* name, const uint32_t* idx, uint32_t n_per_block, uint32_t blocks) {
    T* out;
    hipMalloc(&out, blocks * sizeof(T));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<T, WORDS>), hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, WORDS>), dim3(blocks), dim3(1024), WORDS * sizeof(T), 0, idx, n_per_block, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<T, WORDS>), dim3(blocks), dim3(1024), WORDS * sizeof(T), 0, idx, n_per_block, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double atomics = (double)blocks * (n_per_block / 8192 * 8192) * 5;
    printf("%-10s %8.1f us per launch, %6.1f G lane-atomics/s (%u workgroups x %u)\n", name, ms / 5 * 1e3, atomics / (ms * 1e-3) / 1e9, blocks, n_per_block);
    hipFree(out);
}

int main() {
    const uint32_t blocks = 512, n = 131072;
    uint32_t* h = (uint32_t*)malloc((size_t)blocks * n * 4);
    uint32_t s = 12345;
    for (size_t i = 0; i < (size_t)blocks * n; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
    uint32_t* d;
    hipMalloc(&d, (size_t)blocks * n * 4);
    hipMemcpy(d, h, (size_t)blocks * n * 4, hipMemcpyHostToDevice);
    run<float, 16384>("f32", d, n, blocks);
    run<uint32_t, 16384>("u32", d, n, blocks);
    run<unsigned long long, 8192>("u64", d, n, blocks);
    run<double, 8192>("f64", d, n, blocks);
    run<int, 16384>("i32", d, n, blocks);
    return 0;
}
