// How should the ~20 vector instructions of an activation split be placed around the 3 (or 2 x 3) matrix instructions of a K = 16 block?
// Per "block": 3 x v_mfma_f32_32x32x16_f16 per accumulator and 20 plain VALU (v_fma_f32) that do not depend on them.
//   mode 0  chain:      [M M M on ONE accumulator] [20 VALU]                       (the fused fields' structure up to round 3)
//   mode 1  threaded:   M v*7 M v*7 M v*6, ONE accumulator                          (what hipcc's scheduler writes when left alone)
//   mode 2  pair-serial:[M0 M1 M0 M1 M0 M1 on TWO accumulators] [40 VALU]           (two output tiles from one split, VALU behind)
//   mode 3  pair-threaded: M0 v*7 M1 v*7 M0 v*7 M1 v*7 M0 v*6 M1 v*6                 (two accumulators alternating, VALU in the gaps)
//   mode 4  pair-threaded, 5 per gap + 10 behind
//   mode 5  MFMA only (pair), mode 6 VALU only (40)
// WAVES per SIMD = blockDim / 256 (one workgroup per CU).  Prints ns and cycles (at 2.4 GHz) per block per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_interleave.hip -o mfma_interleave && ./mfma_interleave
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define V(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(ad))
#define V5(i) V(v[(i) % 16]); V(v[(i + 1) % 16]); V(v[(i + 2) % 16]); V(v[(i + 3) % 16]); V(v[(i + 4) % 16])
#define V6(i) V5(i); V(v[(i + 5) % 16])
#define V7(i) V6(i); V(v[(i + 6) % 16])
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void k(int iters, float* __restrict__ out, float seed) {
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(seed + threadIdx.x * 1e-3f + j); b[j] = (_Float16)(seed - j); }
    f16v c0 = {}, c1 = {};
    float v[16];
    for (int j = 0; j < 16; j++) v[j] = seed + j + threadIdx.x;
    const float m = seed * 0.5f + 1.0f, ad = seed + 0.25f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { MFMA(c0, a, b); MFMA(c0, a, b); MFMA(c0, a, b); SB; V7(0); V7(7); V6(14); SB; }
        if (MODE == 1) { MFMA(c0, a, b); SB; V7(0); SB; MFMA(c0, a, b); SB; V7(7); SB; MFMA(c0, a, b); SB; V6(14); SB; }
        if (MODE == 2) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c0, a, b); MFMA(c1, a, b); SB; V7(0); V7(7); V6(14); V7(4); V7(11); V6(2); SB; }
        if (MODE == 3) { MFMA(c0, a, b); SB; V7(0); SB; MFMA(c1, a, b); SB; V7(7); SB; MFMA(c0, a, b); SB; V7(14); SB; MFMA(c1, a, b); SB; V7(5); SB; MFMA(c0, a, b); SB; V6(12); SB; MFMA(c1, a, b); SB; V6(2); SB; }
        if (MODE == 4) { MFMA(c0, a, b); SB; V5(0); SB; MFMA(c1, a, b); SB; V5(5); SB; MFMA(c0, a, b); SB; V5(10); SB; MFMA(c1, a, b); SB; V5(15); SB; MFMA(c0, a, b); SB; V5(4); SB; MFMA(c1, a, b); SB; V5(9); V5(14); V5(3); SB; }
        if (MODE == 5) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c0, a, b); MFMA(c1, a, b); SB; }
        if (MODE == 6) { V7(0); V7(7); V6(14); V7(4); V7(11); V6(2); SB; }
    }
    float s = 0.0f;
    for (int j = 0; j < 16; j++) s += v[j] + c0[j] + c1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int iters, float* out, double blocks_per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, out, 0.0f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, out, 0.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms / 5 * 1e6 / iters;                 // per loop iteration of a SIMD (its waves run concurrently)
    const double waves = threads / 256.0;
    printf("%-58s %2.0f waves/SIMD: %7.1f ns per iteration = %6.1f cycles per K-block per wave-slot (x%.0f waves: %6.1f per block)\n", name, waves, ns, ns * 2.4 / blocks_per_iter,
           waves, ns * 2.4 / blocks_per_iter / waves);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 4000;
    for (int threads : {256, 768}) {
        run<0>("0 chain: MMM (one acc) | 20 VALU", threads, iters, out, 1);
        run<1>("1 threaded, one acc: M v7 M v7 M v6", threads, iters, out, 1);
        run<2>("2 pair-serial: M0M1M0M1M0M1 | 40 VALU  (2 blocks)", threads, iters, out, 2);
        run<3>("3 pair-threaded: M0 v7 M1 v7 M0 v7 M1 v7 M0 v6 M1 v6 (2 blocks)", threads, iters, out, 2);
        run<4>("4 pair-threaded 5 per gap + 15 behind (2 blocks)", threads, iters, out, 2);
        run<5>("5 MFMA only, pair (2 blocks)", threads, iters, out, 2);
        run<6>("6 VALU only, 40 (2 blocks)", threads, iters, out, 2);
    }
    return 0;
}
