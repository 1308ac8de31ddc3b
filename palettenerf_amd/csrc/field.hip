// field.hip -- fused tiny-MLP evaluation of the NeRF field on the gfx950 matrix cores.
//
// Computes, per sample, what nerf/network.py:95-124 computes after the hash-grid lookup:
//     h     = W_s1 . relu(W_s0 . enc)                 (32 -> 64 -> 16, no biases)
//     sigma = exp(h[0]);  geo = h[1:16]
//     rgb   = sigmoid(W_c2 . relu(W_c1 . relu(W_c0 . [SH16(d) ; geo])))   (31 -> 64 -> 64 -> 3)
// in ONE kernel: five GEMMs, SH encoding, concat, four activations and the exp -- the reference
// issues 5 GEMM + ~9 elementwise launches and round-trips every activation through HBM.
//
// MI355X mapping (this is not a GEMM-library call):
//   * the products are evaluated TRANSPOSED, D[feature][sample] = W[feature][k] . act[k][sample], with
//     v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, MI355X_MICROARCH.md): weights are the A operand,
//     activations the B operand.  In that orientation the D fragment of one layer (lane = sample,
//     registers = features) IS the B fragment of the next layer: lanes 0-31 feed k=0, lanes 32-63
//     feed k=1 of every MFMA with the accumulator register they already hold.  Activations never
//     leave the register file -- no LDS round trip, no shuffles, no transposes between layers.
//   * the price is a permuted reduction order over k, paid once on the weights: pnr_nerf_field_pack
//     lays every layer out as [row tile][k step][lane] so that an A fragment is one conflict-free
//     ds_read_b32 of 64 consecutive floats.  48 KiB of LDS per 512-thread workgroup holds all five
//     layers; workgroups are persistent (grid-stride over 256-sample tiles) so the staging is
//     amortised.
//   * layer-1 B operands are read straight from the level-major encoder output [L,B,2]: the 64 lanes
//     of a wave read 32 samples x 2 channels = 256 contiguous bytes per level.
// One wave = 32 samples; 192 MFMAs per wave-tile (18 688 useful + padding FLOP per sample).
#include "pnr_common.hpp"
#include "field_core.hpp"

namespace pnr {

// One thread per packed element: decode (layer, row tile, step, lane) and fetch the weight.
__global__ void __launch_bounds__(256) k_nerf_field_pack(const float* __restrict__ Ws0, const float* __restrict__ Ws1,
                                                         const float* __restrict__ Wc0, const float* __restrict__ Wc1,
                                                         const float* __restrict__ Wc2, float* __restrict__ packed, int limit) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= limit) return;
    const int lane = e & 63, i = lane & 31, h = lane >> 5;
    float v = 0.0f;
    if (e < kS1) {                       // sigma_net[0]  W[64][32]; k order is the natural one
        const int q = (e - kS0) >> 6, tile = q / 16, s = q % 16;
        v = Ws0[(tile * 32 + i) * 32 + 2 * s + h];
    } else if (e < kC0) {                // sigma_net[1]  W[16][64]
        const int s = (e - kS1) >> 6, k = (s / 16) * 32 + frag_row(s % 16, h);
        if (i < 16) v = Ws1[i * 64 + k];
    } else if (e < kC1) {                // color_net[0]  W[64][31] ; columns: 0..15 SH, 16..30 geo_feat 1..15
        const int q = (e - kC0) >> 6, tile = q / 16, s = q % 16;
        int col;
        if (s < 8) col = s + 8 * h;      // steps 0-7: SH coefficient s (lower half-wave) / 8+s (upper)
        else { const int g = frag_row(s - 8, h); col = g >= 1 ? 15 + g : -1; }  // geo feature g; g == 0 is the sigma logit: zero weight
        if (col >= 0) v = Wc0[(tile * 32 + i) * 31 + col];
    } else if (e < kC2) {                // color_net[1]  W[64][64]
        const int q = (e - kC1) >> 6, tile = q / 32, s = q % 32, k = (s / 16) * 32 + frag_row(s % 16, h);
        v = Wc1[(tile * 32 + i) * 64 + k];
    } else {                             // color_net[2]  W[3][64]
        const int s = (e - kC2) >> 6, k = (s / 16) * 32 + frag_row(s % 16, h);
        if (i < 3) v = Wc2[i * 64 + k];
    }
    packed[e] = v;
}

// pack kernel of the split-fp16 blob: one thread per (block, lane, element)
__global__ void __launch_bounds__(256) k_nerf_field_pack_f16x3(const float* __restrict__ Ws0, const float* __restrict__ Ws1,
                                                               const float* __restrict__ Wc0, const float* __restrict__ Wc1,
                                                               const float* __restrict__ Wc2, unsigned char* __restrict__ packed, int limit) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= limit) return;
    const int q = e / 512, lane = (e / 8) & 63, j = e & 7, i = lane & 31, h = lane >> 5;
    float v = 0.0f;
    if (q < 4) { const int rt = q / 2, kb = q % 2; v = Ws0[(rt * 32 + i) * 32 + f16_col_S0(kb, h, j)]; }
    else if (q < 8) { const int kb = q - 4; if (i < 16) v = Ws1[i * 64 + f16_col_from_frag(kb, h, j)]; }
    else if (q < 12) { const int rt = (q - 8) / 2, kb = (q - 8) % 2, col = f16_col_C0(kb, h, j); if (col >= 0) v = Wc0[(rt * 32 + i) * 31 + col]; }
    else if (q < 20) { const int rt = (q - 12) / 4, kb = (q - 12) % 4; v = Wc1[(rt * 32 + i) * 64 + f16_col_from_frag(kb, h, j)]; }
    else v = 0.0f;      // (blocks 20 .. 23: color_net[2] is a vector head now -- k_pack_vec_head writes its 768 bytes over the start of block 20)
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16* blk = reinterpret_cast<_Float16*>(packed + (size_t)q * kF16BlockBytes);
    blk[lane * 8 + j] = hi;
    blk[512 + lane * 8 + j] = lo;
}

constexpr int kFieldThreads = 512;  // 8 waves x 32 samples = 256 samples per workgroup tile

template <int PREC>
__global__ void __launch_bounds__(kFieldThreads) k_nerf_field_fwd(const float* __restrict__ enc /* [16][B][2] */, const float* __restrict__ dirs /* [B][3] */,
                                                                  const float* __restrict__ packed, uint32_t B, float* __restrict__ sigmas,
                                                                  float* __restrict__ rgbs, float enc_scale) {
    __shared__ float w[kPackedFloats];
    for (int i = threadIdx.x * 4; i < kPackedFloats; i += kFieldThreads * 4)
        *reinterpret_cast<float4*>(&w[i]) = *reinterpret_cast<const float4*>(&packed[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const uint32_t ntiles = (B + 255) / 256;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t n = tile * 256 + wave * 32 + (lane & 31);
        const bool valid = n < B;
        const uint32_t nc = valid ? n : (B - 1);
        const float dx = dirs[(size_t)nc * 3], dy = dirs[(size_t)nc * 3 + 1], dz = dirs[(size_t)nc * 3 + 2];
        const FieldOut o = nerf_field_tile<PREC>(w, lane, valid, enc, B, nc, dx, dy, dz, enc_scale);
        if (valid && h == 0) {
            sigmas[n] = expf(o.sigma_logit);                      // trunc_exp forward (activation.py:9)
            rgbs[(size_t)n * 3] = 1.0f / (1.0f + expf(-o.o0));    // sigmoid (nerf/network.py:122)
            rgbs[(size_t)n * 3 + 1] = 1.0f / (1.0f + expf(-o.o1));
            rgbs[(size_t)n * 3 + 2] = 1.0f / (1.0f + expf(-o.o2));
        }
    }
}

// sigma_net alone: sigma = exp(logit) and, optionally, the 15 geometry features per sample ([B,15] row-major, what the colour heads of
// both models take).  Only the first 16 KiB (fp32) / 8 blocks (f16x3) of the packed blob are read.
template <int PREC>
__global__ void __launch_bounds__(kFieldThreads) k_nerf_density_fwd(const float* __restrict__ enc /* [16][B][2] */, const float* __restrict__ packed, uint32_t B,
                                                                    float scale, float* __restrict__ sigmas, float* __restrict__ geo, float enc_scale) {
    constexpr int kFloats = PREC == 0 ? kC0 : 8 * kF16BlockBytes / 4;
    __shared__ float w[kFloats];
    for (int i = threadIdx.x * 4; i < kFloats; i += kFieldThreads * 4)
        *reinterpret_cast<float4*>(&w[i]) = *reinterpret_cast<const float4*>(&packed[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const uint32_t ntiles = (B + 255) / 256;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t n = tile * 256 + wave * 32 + (lane & 31);
        const bool valid = n < B;
        const uint32_t nc = valid ? n : (B - 1);
        const f32x16 g = nerf_density_tile<PREC>(w, lane, valid, enc, B, nc, enc_scale);
        if (!valid) continue;
        if (h == 0) sigmas[n] = scale * expf(g[0]);   // trunc_exp forward (activation.py:9), times the caller's density_scale (1 = plain sigma)
        if (geo) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int row = frag_row(r, 0) + 4 * h;   // 0..15
                if (row >= 1) geo[(size_t)n * 15 + row - 1] = g[r];
            }
        }
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

uint64_t pnr_nerf_field_packed_bytes(void) { return (uint64_t)kPackedFloats * 4; }

int pnr_nerf_field_pack(const float* w_sigma0, const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                        float* packed, int precision, pnr_stream_t stream) {
    if (!w_sigma0 || !w_sigma1 || !packed) return PNR_ERR_INVALID;
    const bool sigma_only = !w_color0 && !w_color1 && !w_color2;     // the density-only users (pnr_nerf_density_forward, pnr_occupancy_update) read that part alone
    if (!sigma_only && (!w_color0 || !w_color1 || !w_color2)) return PNR_ERR_INVALID;
    if (precision == PNR_FIELD_FP32)
        hipLaunchKernelGGL(k_nerf_field_pack, dim3(cdiv(kPackedFloats, 256)), dim3(256), 0, as_stream(stream), w_sigma0, w_sigma1, w_color0, w_color1,
                           w_color2, packed, sigma_only ? kC0 : kPackedFloats);
    else if (precision == PNR_FIELD_F16X3 || precision == PNR_FIELD_F16X2) {
        hipLaunchKernelGGL(k_nerf_field_pack_f16x3, dim3(cdiv(kF16Blocks * 512, 256)), dim3(256), 0, as_stream(stream), w_sigma0, w_sigma1, w_color0,
                           w_color1, w_color2, reinterpret_cast<unsigned char*>(packed), (sigma_only ? 8 : kF16Blocks) * 512);
        if (!sigma_only) hipLaunchKernelGGL(k_pack_vec_head, dim3(1), dim3(192), 0, as_stream(stream), w_color2, packed + 20 * kF16BlockBytes / 4);
    } else
        return PNR_ERR_UNSUPPORTED;
    return check_launch();
}

int pnr_nerf_field_forward(const float* enc, const float* dirs, const float* packed, uint32_t B, float* sigmas, float* rgbs, int precision,
                           float enc_scale, pnr_stream_t stream) {
    if (!(enc_scale > 0.0f)) enc_scale = 1.0f;
    if (precision != PNR_FIELD_FP32 && precision != PNR_FIELD_F16X3 && precision != PNR_FIELD_F16X2) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!enc || !dirs || !packed || !sigmas || !rgbs) return PNR_ERR_INVALID;
    const uint32_t ntiles = cdiv(B, 256);
    const uint32_t grid = ntiles < 512u ? ntiles : 512u;  // 2 persistent workgroups per CU
    if (precision == PNR_FIELD_FP32)
        hipLaunchKernelGGL(k_nerf_field_fwd<0>, dim3(grid), dim3(kFieldThreads), 0, as_stream(stream), enc, dirs, packed, B, sigmas, rgbs, enc_scale);
    else if (precision == PNR_FIELD_F16X2)
        hipLaunchKernelGGL(k_nerf_field_fwd<2>, dim3(grid), dim3(kFieldThreads), 0, as_stream(stream), enc, dirs, packed, B, sigmas, rgbs, enc_scale);
    else
        hipLaunchKernelGGL(k_nerf_field_fwd<1>, dim3(grid), dim3(kFieldThreads), 0, as_stream(stream), enc, dirs, packed, B, sigmas, rgbs, enc_scale);
    return check_launch();
}

int pnr_nerf_density_forward(const float* enc, const float* packed, uint32_t B, float scale, float* sigmas, float* geo_feat, int precision,
                             float enc_scale, pnr_stream_t stream) {
    if (!(enc_scale > 0.0f)) enc_scale = 1.0f;
    if (precision == PNR_FIELD_F16X2) precision = PNR_FIELD_F16X3;   // sigma feeds the occupancy grid and the training composites: always the split form
    if (precision != PNR_FIELD_FP32 && precision != PNR_FIELD_F16X3) return PNR_ERR_UNSUPPORTED;
    if (B == 0) return PNR_OK;
    if (!enc || !packed || !sigmas) return PNR_ERR_INVALID;
    const uint32_t ntiles = cdiv(B, 256);
    const uint32_t grid = ntiles < 1024u ? ntiles : 1024u;
    if (precision == PNR_FIELD_FP32)
        hipLaunchKernelGGL(k_nerf_density_fwd<0>, dim3(grid), dim3(kFieldThreads), 0, as_stream(stream), enc, packed, B, scale, sigmas, geo_feat, enc_scale);
    else
        hipLaunchKernelGGL(k_nerf_density_fwd<1>, dim3(grid), dim3(kFieldThreads), 0, as_stream(stream), enc, packed, B, scale, sigmas, geo_feat, enc_scale);
    return check_launch();
}

}  // extern "C"
