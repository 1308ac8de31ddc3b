// shencoder.hip -- real spherical-harmonics direction encoder (degree <= 8) for gfx950.
//
// The reference tabulates 64 hand-expanded polynomials plus 192 derivative polynomials
// (shencoder/src/shencoder.cu:49-354).  Here the same basis is evaluated from its product form
//   Y_l^m = K * A_m(x,y) * Q_l^m(z),   A_m = Re/Im (x+iy)^m,   Q_l^m = d^m P_l/dz^m
// with generated coefficient tables (gen_sh_tables.py): 8 complex powers by recurrence, one short
// Horner chain in z^2 per (l,m).  Everything is unrolled at compile time per degree; outputs are
// written as 16-byte vectors.
#include "pnr_common.hpp"
#include "sh_eval.hpp"

namespace pnr {

template <int DEG, bool GRAD>
__global__ void __launch_bounds__(256) k_sh_fwd(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t D,
                                                float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
    const float z2 = z * z;
    float re[DEG], im[DEG];
    re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < DEG; m++) {
        re[m] = x * re[m - 1] - y * im[m - 1];
        im[m] = fmaf(x, im[m - 1], y * re[m - 1]);
    }
    float out[C2];
    float gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
#pragma unroll
    for (int l = 0; l < DEG; l++) {
#pragma unroll
        for (int m = 0; m <= l; m++) {
            const float q = sh_poly(SH_Q[l][m], z, z2);
            const int ip = l * l + l + m, in_ = l * l + l - m;
            out[ip] = re[m] * q;
            if (m) out[in_] = im[m] * q;
            if constexpr (GRAD) {
                const float dq = sh_poly(SH_DQ[l][m], z, z2);
                const float fm = (float)m;
                gx[ip] = m ? fm * re[m - 1] * q : 0.0f;
                gy[ip] = m ? -fm * im[m - 1] * q : 0.0f;
                gz[ip] = re[m] * dq;
                if (m) {
                    gx[in_] = fm * im[m - 1] * q;
                    gy[in_] = fm * re[m - 1] * q;
                    gz[in_] = im[m] * dq;
                }
            }
        }
    }
    float* o = outputs + (size_t)b * C2;
    if constexpr (C2 % 4 == 0) {
#pragma unroll
        for (int i = 0; i < C2; i += 4) *reinterpret_cast<float4*>(o + i) = make_float4(out[i], out[i + 1], out[i + 2], out[i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < C2; i++) o[i] = out[i];
    }
    if constexpr (GRAD) {
        float* dx = dy_dx + (size_t)b * D * C2;  // [B, D, C2] : d/dx block, d/dy block, d/dz block
#pragma unroll
        for (int i = 0; i < C2; i++) { dx[i] = gx[i]; dx[C2 + i] = gy[i]; dx[2 * C2 + i] = gz[i]; }
    }
}

// reference shencoder.cu:358-382 ; '+=' into caller-zeroed grad_inputs
__global__ void __launch_bounds__(256) k_sh_bwd(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D;
    const float* g = grad + (size_t)b * C2;
    const float* dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], dd[ch], acc);
    grad_inputs[t] = acc;
}

template <int DEG>
static int launch_sh(const float* inputs, float* outputs, uint32_t B, uint32_t D, float* dy_dx, hipStream_t s) {
    const dim3 grid(cdiv(B, 256)), block(256);
    if (dy_dx) hipLaunchKernelGGL((k_sh_fwd<DEG, true>), grid, block, 0, s, inputs, outputs, B, D, dy_dx);
    else hipLaunchKernelGGL((k_sh_fwd<DEG, false>), grid, block, 0, s, inputs, outputs, B, D, dy_dx);
    return check_launch();
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx, pnr_stream_t stream) {
    if (D != 3) return PNR_ERR_UNSUPPORTED;        // "SH encoder only support input dim == 3" (sphere_harmonics.py:69)
    if (C < 1 || C > 8) return PNR_ERR_UNSUPPORTED; // degree in [1, 8] (sphere_harmonics.py:70)
    if (B == 0) return PNR_OK;
    if (!inputs || !outputs) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 1: return launch_sh<1>(inputs, outputs, B, D, dy_dx, s);
        case 2: return launch_sh<2>(inputs, outputs, B, D, dy_dx, s);
        case 3: return launch_sh<3>(inputs, outputs, B, D, dy_dx, s);
        case 4: return launch_sh<4>(inputs, outputs, B, D, dy_dx, s);
        case 5: return launch_sh<5>(inputs, outputs, B, D, dy_dx, s);
        case 6: return launch_sh<6>(inputs, outputs, B, D, dy_dx, s);
        case 7: return launch_sh<7>(inputs, outputs, B, D, dy_dx, s);
        default: return launch_sh<8>(inputs, outputs, B, D, dy_dx, s);
    }
}

int pnr_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx, float* grad_inputs,
                           pnr_stream_t stream) {
    (void)inputs;
    if (D != 3) return PNR_ERR_UNSUPPORTED;
    if (C < 1 || C > 8) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!grad || !dy_dx || !grad_inputs) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_sh_bwd, dim3(cdiv(B * D, 256)), dim3(256), 0, as_stream(stream), grad, B, D, C * C, dy_dx, grad_inputs);
    return check_launch();
}

}  // extern "C"
