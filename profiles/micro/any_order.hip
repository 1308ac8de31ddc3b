// any_order.hip -- does this runtime honour hipExtAnyOrderLaunch (AQL barrier bit cleared) on gfx950?
// Sequence on ONE stream:  A = spin(T) ; B = stamp [normal | any-order] ; C = stamp (normal)
// Reported (100 MHz wall clock, microseconds relative to A's start): A.end, B.start, C.start.
//   ordered:    B.start >= A.end
//   any-order:  B.start <  A.end (B overlaps A) and C.start >= max(A.end, B.end)
// Build: hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void k_spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall_clock64();
}
__global__ void k_stamp(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall_clock64();
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(d, 0, 64, s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 20000ull /* 200 us */, d);
            hipExtLaunchKernelGGL(k_stamp, dim3(64), dim3(256), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, 2000ull /* 20 us */, d + 2);
            hipLaunchKernelGGL(k_stamp, dim3(64), dim3(256), 0, s, 100ull, d + 4);
            hipStreamSynchronize(s);
            unsigned long long h[8];
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            const double u = 0.01;
            printf("%s  A [0, %.1f]  B [%.1f, %.1f]  C [%.1f, %.1f] us\n", mode ? "any-order" : "ordered  ", (h[1] - h[0]) * u, ((long long)(h[2] - h[0])) * u,
                   ((long long)(h[3] - h[0])) * u, ((long long)(h[4] - h[0])) * u, ((long long)(h[5] - h[0])) * u);
        }
    }
    // cost of a launch pair: 200 x (A 5 us ; B 5 us) ordered vs B any-order
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < 200; i++) {
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 500ull, d);
            hipExtLaunchKernelGGL(k_stamp, dim3(64), dim3(256), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, 500ull, d + 2);
        }
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("200 x (A 5us; B 5us) %s: %.1f us per pair\n", mode ? "B any-order" : "ordered", ms * 1000.0 / 200);
    }
    return 0;
}
