// checksum.hip -- order-independent 64-bit checksums of device buffers in one launch.
//
// The fused kernels run on blobs DERIVED from a model's parameters (MFMA-ordered weights, interleaved / half hash tables).  torch tells
// when a parameter changed through its version counter -- except for writes through `.data` (torch_ema's copy_to / restore around an
// evaluation, nerf/utils.py:829-839, 959-961; `p.data.uniform_()`), which change neither identity nor version.  The frame loops therefore
// checksum the sources of their blobs once per frame (this kernel: a few KiB of weights in full, the 50 MB tables on a stride) and compare
// with the checksums taken when the blobs were built; a mismatch rebuilds the blobs and renders the frame again (palettenerf_amd/fused.py).
#include "pnr_common.hpp"

namespace pnr {

constexpr uint32_t kChecksumMax = 24;
struct ChecksumArgs { const uint32_t* ptr[kChecksumMax]; uint64_t words[kChecksumMax]; uint32_t stride[kChecksumMax]; };

__global__ void __launch_bounds__(256) k_checksum(ChecksumArgs a, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long red[4];
    const uint32_t b = blockIdx.y;
    const uint32_t* __restrict__ p = a.ptr[b];
    const uint64_t stride = a.stride[b], n = (a.words[b] + stride - 1) / stride;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t w = p[i * stride];
        acc += ((unsigned long long)(w ^ ((uint32_t)i * 0x9E3779B1u)) + 0x632BE59BD9B4E019ull) * (0xD6E8FEB86659FD93ull + 2ull * i);   // position-dependent, summed: any order
    }
    for (int off = PNR_WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, PNR_WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[b], red[0] + red[1] + red[2] + red[3]);
}

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_checksum(const void* const* buffers, const uint64_t* nbytes, const uint32_t* word_stride, uint32_t count, uint64_t* out, pnr_stream_t stream) {
    if (count == 0) return PNR_OK;
    if (!buffers || !nbytes || !out || count > kChecksumMax) return PNR_ERR_INVALID;
    ChecksumArgs a = {};
    uint64_t most = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (!buffers[i] || (nbytes[i] & 3) || (reinterpret_cast<uintptr_t>(buffers[i]) & 3)) return PNR_ERR_INVALID;
        a.ptr[i] = static_cast<const uint32_t*>(buffers[i]);
        a.words[i] = nbytes[i] / 4;
        a.stride[i] = word_stride && word_stride[i] ? word_stride[i] : 1u;
        const uint64_t n = (a.words[i] + a.stride[i] - 1) / a.stride[i];
        if (n > most) most = n;
    }
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(out, 0, sizeof(uint64_t) * count, s) != hipSuccess) return PNR_ERR_LAUNCH;
    const uint32_t blocks = (uint32_t)((most + 1023) / 1024 < 1 ? 1 : ((most + 1023) / 1024 > 256 ? 256 : (most + 1023) / 1024));
    hipLaunchKernelGGL(k_checksum, dim3(blocks, count), dim3(256), 0, s, a, reinterpret_cast<unsigned long long*>(out));
    return check_launch();
}
