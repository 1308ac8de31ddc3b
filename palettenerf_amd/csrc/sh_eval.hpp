// sh_eval.hpp -- shared evaluation helpers of the real SH basis (see shencoder.hip for the derivation).
#pragma once
#include <hip/hip_runtime.h>

namespace pnr {

#include "sh_tables.inc"

__device__ __forceinline__ float sh_poly(const ShPoly& p, float z, float z2) {
    float r = p.c[p.n - 1];
    for (int i = p.n - 2; i >= 0; i--) r = fmaf(r, z2, p.c[i]);
    return p.par ? r * z : r;
}

// all DEG*DEG basis values at the direction (x,y,z)
template <int DEG>
__device__ __forceinline__ void sh_eval(float x, float y, float z, float out[DEG * DEG]) {
    const float z2 = z * z;
    float re[DEG], im[DEG];
    re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < DEG; m++) {
        re[m] = x * re[m - 1] - y * im[m - 1];
        im[m] = fmaf(x, im[m - 1], y * re[m - 1]);
    }
#pragma unroll
    for (int l = 0; l < DEG; l++) {
#pragma unroll
        for (int m = 0; m <= l; m++) {
            const float q = sh_poly(SH_Q[l][m], z, z2);
            out[l * l + l + m] = re[m] * q;
            if (m) out[l * l + l - m] = im[m] * q;
        }
    }
}

}  // namespace pnr
