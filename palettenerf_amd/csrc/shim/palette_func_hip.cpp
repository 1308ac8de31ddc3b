// palette_func_hip.cpp -- reference-side binding of the C ABI for the `_palette_func` extension ("native boundary").
//
// The reference builds `_palette_func` from palette/src/bindings.cpp + palette/src/palette.cu (palette/backend.py:32-39).  bindings.cpp is
// plain C++ / pybind11 (compute_RGB_histogram, bindings.cpp:40-91, plus the two m.def lines for the HSV kernels); palette.cu is CUDA.
// This file is what a maintainer drops in INSTEAD of palette.cu on an MI355X box: the two functions palette/src/palette_func.h declares,
// with the same signatures, forwarding to libpnr_hip.so's pnr_rgb_to_hsv / pnr_hsv_to_rgb (include/pnr.h).  bindings.cpp itself is compiled
// unmodified from where it lies, so the module exports exactly the reference's pybind surface.
//
// Dtype: the reference dispatches over float/double/half but its Python wrapper forces fp32 (palette/utils.py:259,279 custom_fwd
// cast_inputs=torch.float32); other dtypes are rejected here with the error type TORCH_CHECK gives (RuntimeError in Python).
// Stream: the reference launches on the legacy default stream (no stream argument at palette.cu:139,147); so does this binding.
#include <stdint.h>
#include <torch/torch.h>

#include "../../../include/pnr.h"

static void check(const at::Tensor& t, const char* name, uint32_t n) {
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be a float32 tensor (the Python wrapper casts to fp32)");
    TORCH_CHECK(t.is_contiguous(), name, " must be a contiguous tensor");
    TORCH_CHECK(t.device().is_cuda(), name, " must be a CUDA(HIP) tensor");
    TORCH_CHECK((uint64_t)t.numel() >= (uint64_t)n * 3, name, " holds fewer than n_rays x 3 values");
}

void rgb_to_hsv(const uint32_t n_rays, const at::Tensor input, at::Tensor output) {
    check(input, "input", n_rays);
    check(output, "output", n_rays);
    const int rc = pnr_rgb_to_hsv(n_rays, input.data_ptr<float>(), output.data_ptr<float>(), nullptr);
    TORCH_CHECK(rc == PNR_OK, "pnr_rgb_to_hsv: ", pnr_error_string(rc));
}

void hsv_to_rgb(const uint32_t n_rays, const at::Tensor input, at::Tensor output) {
    check(input, "input", n_rays);
    check(output, "output", n_rays);
    const int rc = pnr_hsv_to_rgb(n_rays, input.data_ptr<float>(), output.data_ptr<float>(), nullptr);
    TORCH_CHECK(rc == PNR_OK, "pnr_hsv_to_rgb: ", pnr_error_string(rc));
}
