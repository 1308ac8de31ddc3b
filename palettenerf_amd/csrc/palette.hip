// palette.hip -- RGB <-> HSV conversion used by PaletteNeRF's regional editing, for gfx950.
// Follows the reference's palette extension (palette/src/palette.cu:45-133): H in [0,360),
// S and V in percent, equality test |a-b| < 1e-9, branch order r, g, b.
#include "pnr_common.hpp"
#include "hsv_core.hpp"

namespace pnr {

__global__ void __launch_bounds__(256) k_rgb_to_hsv(uint32_t n, const float* __restrict__ input, float* __restrict__ output) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float h, s, v;
    rgb_to_hsv_px(input[i * 3], input[i * 3 + 1], input[i * 3 + 2], h, s, v);
    output[i * 3] = h; output[i * 3 + 1] = s; output[i * 3 + 2] = v;
}

__global__ void __launch_bounds__(256) k_hsv_to_rgb(uint32_t n, const float* __restrict__ input, float* __restrict__ output) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r, g, b;
    hsv_to_rgb_px(input[i * 3], input[i * 3 + 1], input[i * 3 + 2], r, g, b);
    output[i * 3] = r; output[i * 3 + 1] = g; output[i * 3 + 2] = b;
}

// weighted RGB histogram (reference: CPU C++ compute_RGB_histogram, palette/src/bindings.cpp:40-91).
// Bin index = bits of clamp(c, 0, 0.999) * 2^bpc per channel, r most significant; fp64 accumulators
// (global_atomic_add_f64).  Bin centres are written by the first 2^(3 bpc) threads.
__global__ void __launch_bounds__(256) k_rgb_histogram(const float* __restrict__ rgb, const float* __restrict__ weights, uint32_t n, int bpc,
                                                       double* __restrict__ bin_weights, float* __restrict__ bin_centers) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t num_bins = 1u << (3 * bpc);
    if (i < num_bins) {
        uint32_t code = i;
        for (int k = 0; k < 3; k++) {
            const float c = (float)(code & ((1u << bpc) - 1u));
            bin_centers[i * 3 + (2 - k)] = (c + 0.5f) / (float)(1 << bpc);
            code >>= bpc;
        }
    }
    for (uint32_t p = i; p < n; p += gridDim.x * blockDim.x) {
        uint32_t index = 0;
        for (int k = 0; k < 3; k++) {
            const float c = fmaxf(0.0f, fminf(0.999f, rgb[(size_t)p * 3 + k]));
            index = (index << bpc) + (uint32_t)(c * (float)(1 << bpc));
        }
        unsafeAtomicAdd(&bin_weights[index], (double)weights[p]);
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" {

int pnr_rgb_to_hsv(uint32_t n, const float* input, float* output, pnr_stream_t stream) {
    if (n == 0) return PNR_OK;
    if (!input || !output) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_rgb_to_hsv, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), n, input, output);
    return check_launch();
}
int pnr_hsv_to_rgb(uint32_t n, const float* input, float* output, pnr_stream_t stream) {
    if (n == 0) return PNR_OK;
    if (!input || !output) return PNR_ERR_INVALID;
    hipLaunchKernelGGL(k_hsv_to_rgb, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), n, input, output);
    return check_launch();
}

int pnr_rgb_histogram(const float* colors_rgb, const float* weights, uint32_t n, int bits_per_channel, double* bin_weights, float* bin_centers,
                      pnr_stream_t stream) {
    if (bits_per_channel < 1 || bits_per_channel > 8) return PNR_ERR_UNSUPPORTED;   // the reference asserts 1 <= bpc <= 8 (palette/utils.py:136)
    if (!bin_weights || !bin_centers || (n && (!colors_rgb || !weights))) return PNR_ERR_INVALID;
    const uint32_t num_bins = 1u << (3 * bits_per_channel);
    if (hipMemsetAsync(bin_weights, 0, sizeof(double) * num_bins, as_stream(stream)) != hipSuccess) return PNR_ERR_LAUNCH;
    uint32_t blocks = cdiv(n > num_bins ? n : num_bins, 256);
    const uint32_t min_blocks = cdiv(num_bins, 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < min_blocks) blocks = min_blocks;
    hipLaunchKernelGGL(k_rgb_histogram, dim3(blocks), dim3(256), 0, as_stream(stream), colors_rgb, weights, n, bits_per_channel, bin_weights, bin_centers);
    return check_launch();
}

}  // extern "C"
