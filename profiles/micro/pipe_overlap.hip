// Do the matrix pipe and the VALU of ONE SIMD overlap on gfx950, and what does a VALU instruction cost?
//   mode 0  every wave: MFMA only (v_mfma_f32_32x32x16_f16, 4 independent accumulators)
//   mode 1  every wave: VALU only (v_fma_f32, 16 independent chains)
//   mode 2  waves 0-3 MFMA, waves 4-7 VALU  (waves w and w+4 of a 512-thread workgroup share a SIMD)
//   mode 3  every wave: 1 MFMA followed by 8 independent VALU, repeated (in-stream interleave)
//   mode 4  every wave: a phase of 16 MFMA then a phase of 128 VALU that DEPENDS on nothing (phase-split, like a layer loop)
//   mode 5  as 4, but the VALU phase consumes the MFMA results and feeds the next MFMA phase (true layer dependency)
// hipcc --offload-arch=gfx950 -O3 pipe_overlap.hip -o pipe_overlap && ./pipe_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define VFMA(x, m, ad) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(ad))   // one plain (unpacked) VALU instruction

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, float* __restrict__ out, float seed) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(seed + threadIdx.x * 1e-3f + j); b[j] = (_Float16)(seed - j); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v[16];
    for (int j = 0; j < 16; j++) v[j] = seed + j + threadIdx.x;
    const float m = seed * 0.5f + 1.0f, ad = seed + 0.25f;
    const bool do_mfma = MODE == 0 || (MODE == 2 && wave < 4) || MODE >= 3;
    const bool do_valu = MODE == 1 || (MODE == 2 && wave >= 4) || MODE >= 3;
    for (int it = 0; it < iters; it++) {
        if (MODE <= 2) {
            if (do_mfma) {
#pragma unroll
                for (int r = 0; r < 4; r++) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c2, a, b); MFMA(c3, a, b); }
            }
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int j = 0; j < 16; j++) VFMA(v[j], m, ad);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if ((r & 3) == 0) MFMA(c0, a, b); else if ((r & 3) == 1) MFMA(c1, a, b); else if ((r & 3) == 2) MFMA(c2, a, b); else MFMA(c3, a, b);
#pragma unroll
                for (int j = 0; j < 8; j++) VFMA(v[(r & 1) * 8 + j], m, ad);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; r++) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c2, a, b); MFMA(c3, a, b); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int j = 0; j < 16; j++) VFMA(v[j], m, ad);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c2, a, b); MFMA(c3, a, b); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = fmaf(v[j], m, c0[j] + c1[j] + c2[j] + c3[j]);      // 64 VALU consuming the accumulators
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 16; j++) VFMA(v[j], m, ad);                                // + 64 more
#pragma unroll
            for (int j = 0; j < 8; j++) { a[j] = (_Float16)v[j]; b[j] = (_Float16)v[8 + j]; }          // feeds the next phase
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.0f;
    for (int j = 0; j < 16; j++) s += v[j] + c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out, double mfma_per_wave_iter, double valu_per_wave_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, iters, out, 0.0f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, iters, out, 0.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / 5 * 1e3;
    // per SIMD: waves per SIMD = blocks_per_cu * 2
    const double bpc = blocks / 256.0;
    printf("%-44s blocks/CU %.0f: %8.1f us  | per SIMD per iteration: %.0f ns  (MFMA issued %.0f, VALU issued %.0f per SIMD-iteration)\n", name, bpc, us,
           us * 1e3 / iters, mfma_per_wave_iter * bpc, valu_per_wave_iter * bpc);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 512 * sizeof(float));
    const int iters = 2000;
    for (int blocks : {256, 512}) {
        // per wave-iteration: 16 MFMA (= 512 matrix-pipe cycles at 32 cycles each), 128 VALU
        run<0>("0 all waves MFMA (16/iter/wave)", blocks, iters, out, 2 * 16, 0);
        run<1>("1 all waves VALU (128/iter/wave)", blocks, iters, out, 0, 2 * 128);
        run<2>("2 waves 0-3 MFMA, 4-7 VALU (same SIMDs)", blocks, iters, out, 16, 128);
        run<3>("3 in-stream 1 MFMA : 8 VALU", blocks, iters, out, 2 * 16, 2 * 128);
        run<4>("4 phase-split 16 MFMA | 128 VALU, independent", blocks, iters, out, 2 * 16, 2 * 128);
        run<5>("5 phase-split, VALU depends on MFMA and back", blocks, iters, out, 2 * 16, 2 * 128 + 24);
    }
    return 0;
}
