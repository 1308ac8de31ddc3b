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

// [SH(dir) | tail] rows in one pass: color_net's input torch.cat([encoder_dir(d), geo_feat]) (nerf/network.py:109-115, palette/network.py:248-249)
// without the [B, C2] intermediate and the concatenation's round trip.  The SH values are k_sh_fwd<DEG, false>'s, operation for operation.
// A workgroup's 256 rows are put together in LDS (row stride W = C2 + t) and leave as one contiguous block of 16-byte stores.
template <int DEG>
__global__ void __launch_bounds__(256) k_sh_cat_fwd(const float* __restrict__ inputs, const float* __restrict__ tail, uint32_t t,
                                                    float* __restrict__ outputs, uint32_t B, uint32_t tail_stride /* floats per source row */,
                                                    uint32_t tail_off /* first source column */, float* __restrict__ sigma /* or null: exp(source column 0) */) {
    extern __shared__ float tile[];
    constexpr int C2 = DEG * DEG;
    const uint32_t W = C2 + t, row0 = blockIdx.x * 256, b = row0 + threadIdx.x, nrows = B - row0 < 256u ? B - row0 : 256u;
    if (b < B) {
        const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
        const float z2 = z * z;
        float re[DEG], im[DEG];
        re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
        for (int m = 1; m < DEG; m++) {
            re[m] = x * re[m - 1] - y * im[m - 1];
            im[m] = fmaf(x, im[m - 1], y * re[m - 1]);
        }
        float* o = tile + threadIdx.x * W;
#pragma unroll
        for (int l = 0; l < DEG; l++) {
#pragma unroll
            for (int m = 0; m <= l; m++) {
                const float q = sh_poly(SH_Q[l][m], z, z2);
                o[l * l + l + m] = re[m] * q;
                if (m) o[l * l + l - m] = im[m] * q;
            }
        }
    }
    const float* tsrc = tail + (size_t)row0 * tail_stride;
    for (uint32_t i = threadIdx.x; i < nrows * t; i += 256) {
        const uint32_t r = i / t, c = i - r * t;
        tile[r * W + C2 + c] = tsrc[(size_t)r * tail_stride + tail_off + c];
    }
    if (sigma && threadIdx.x < nrows) sigma[row0 + threadIdx.x] = expf(tsrc[(size_t)threadIdx.x * tail_stride]);   // trunc_exp's forward (activation.py:11-13)
    __syncthreads();
    float* dst = outputs + (size_t)row0 * W;      // 256 W floats per workgroup: 16-byte aligned for every W
    const uint32_t total = nrows * W, quads = total / 4;
    for (uint32_t i = threadIdx.x; i < quads; i += 256)
        reinterpret_cast<float4*>(dst)[i] = make_float4(tile[i * 4], tile[i * 4 + 1], tile[i * 4 + 2], tile[i * 4 + 3]);
    for (uint32_t i = quads * 4 + threadIdx.x; i < total; i += 256) dst[i] = tile[i];
}

template <int DEG>
static int launch_sh_cat(const float* inputs, const float* tail, uint32_t t, float* outputs, uint32_t B, hipStream_t s, uint32_t tail_stride = 0, uint32_t tail_off = 0,
                         float* sigma = nullptr) {
    hipLaunchKernelGGL((k_sh_cat_fwd<DEG>), dim3(cdiv(B, 256)), dim3(256), 256 * (DEG * DEG + t) * sizeof(float), s, inputs, tail, t, outputs, B,
                       tail_stride ? tail_stride : t, tail_off, sigma);
    return check_launch();
}
// grad_h[b] = [dsigma[b] * exp(clamp(h[b][0], -15, 15)), dout[b][C2 : C2 + hw - 1]]: trunc_exp's backward (activation.py:14-17) and the column slice, one pass
__global__ void __launch_bounds__(256) k_sigma_geo_cat_bwd(const float* __restrict__ h, uint32_t hw, const float* __restrict__ dsigma, const float* __restrict__ dout, uint32_t C2,
                                                           uint32_t B, float* __restrict__ grad_h) {
    // blockDim = (columns rounded up to a power of two, 256 / that): consecutive lanes walk a row, no division
    const uint32_t c = threadIdx.x, b = blockIdx.x * blockDim.y + threadIdx.y;
    if (c >= hw || b >= B) return;
    const size_t i = (size_t)b * hw + c;
    float v = 0.0f;
    if (c == 0) { if (dsigma) v = dsigma[b] * expf(fminf(fmaxf(h[i], -15.0f), 15.0f)); }
    else if (dout) v = dout[(size_t)b * (C2 + hw - 1) + C2 + c - 1];
    grad_h[i] = v;
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

int pnr_sigma_geo_cat_forward(const float* h, uint32_t hw, const float* dirs, uint32_t C, uint32_t B, float* sigma, float* out, pnr_stream_t stream) {
    if (C < 1 || C > 8 || hw < 2 || C * C + hw - 1 > 64) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!h || !dirs || !sigma || !out) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 1: return launch_sh_cat<1>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        case 2: return launch_sh_cat<2>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        case 3: return launch_sh_cat<3>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        case 4: return launch_sh_cat<4>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        case 5: return launch_sh_cat<5>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        case 6: return launch_sh_cat<6>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);
        default: return launch_sh_cat<7>(dirs, h, hw - 1, out, B, s, hw, 1, sigma);   // C == 8 has no room for a tail
    }
}

int pnr_sigma_geo_cat_backward(const float* h, uint32_t hw, const float* dsigma, const float* dout, uint32_t C, uint32_t B, float* grad_h, pnr_stream_t stream) {
    if (C < 1 || C > 8 || hw < 2 || C * C + hw - 1 > 64) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!h || !grad_h) return PNR_ERR_INVALID;
    if ((uint64_t)B * hw >= (1ull << 32)) return PNR_ERR_UNSUPPORTED;
    uint32_t bx = 2;
    while (bx < hw) bx <<= 1;
    hipLaunchKernelGGL(k_sigma_geo_cat_bwd, dim3(cdiv(B, 256 / bx)), dim3(bx, 256 / bx), 0, as_stream(stream), h, hw, dsigma, dout, C * C, B, grad_h);
    return check_launch();
}

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

int pnr_sh_encode_cat_forward(const float* inputs, const float* tail, uint32_t tail_cols, float* outputs, uint32_t B, uint32_t C, pnr_stream_t stream) {
    if (C < 1 || C > 8 || tail_cols == 0 || C * C + tail_cols > 64) return PNR_ERR_UNSUPPORTED;   // one LDS tile of 256 x 64 floats at most
    if (B == 0) return PNR_OK;
    if (!inputs || !tail || !outputs) return PNR_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 1: return launch_sh_cat<1>(inputs, tail, tail_cols, outputs, B, s);
        case 2: return launch_sh_cat<2>(inputs, tail, tail_cols, outputs, B, s);
        case 3: return launch_sh_cat<3>(inputs, tail, tail_cols, outputs, B, s);
        case 4: return launch_sh_cat<4>(inputs, tail, tail_cols, outputs, B, s);
        case 5: return launch_sh_cat<5>(inputs, tail, tail_cols, outputs, B, s);
        case 6: return launch_sh_cat<6>(inputs, tail, tail_cols, outputs, B, s);
        default: return launch_sh_cat<7>(inputs, tail, tail_cols, outputs, B, s);   // C == 8 has no room for a tail (64 + t > 64)
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
