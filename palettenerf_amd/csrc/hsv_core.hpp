// hsv_core.hpp -- RGB <-> HSV per pixel, shared by the stand-alone operators (palette.hip) and the RegionEdit epilogue of the fused
// PaletteNeRF field (palette_field.hip).  Follows palette/src/palette.cu:45-133: H in [0,360), S and V in percent, equality test
// |a-b| < 1e-9 (palette.cu:19), branch order r, g, b.
#pragma once
#include "pnr_common.hpp"

namespace pnr {

__device__ __forceinline__ void rgb_to_hsv_px(float r, float g, float b, float& h, float& s, float& v) {
    const float c_max = fmaxf(fmaxf(r, g), b), c_min = fminf(fminf(r, g), b), diff = c_max - c_min;
    if ((double)fabsf(diff) < 1e-9) h = 0.0f;
    else if ((double)fabsf(c_max - r) < 1e-9) h = (float)fmod((double)(60.0f * ((g - b) / diff) + 360.0f), 360.0);
    else if ((double)fabsf(c_max - g) < 1e-9) h = (float)fmod((double)(60.0f * ((b - r) / diff) + 120.0f), 360.0);
    else h = (float)fmod((double)(60.0f * ((r - g) / diff) + 240.0f), 360.0);
    if ((double)fabsf(c_max) < 1e-9) s = 0.0f; else s = (diff / c_max) * 100.0f;
    v = c_max * 100.0f;
}

__device__ __forceinline__ void hsv_to_rgb_px(float h, float s, float v, float& r, float& g, float& b) {
    const float c = s / 100.0f * v / 100.0f;
    const float x = c * (1.0f - fabsf((float)fmod((double)(h / 60.0f), 2.0) - 1.0f));
    const float m = v / 100.0f - c;
    r = 0.0f; g = 0.0f; b = 0.0f;
    if (h >= 0.0f && h < 60.0f) { r = c; g = x; }
    else if (h >= 60.0f && h < 120.0f) { r = x; g = c; }
    else if (h >= 120.0f && h < 180.0f) { g = c; b = x; }
    else if (h >= 180.0f && h < 240.0f) { g = x; b = c; }
    else if (h >= 240.0f && h < 300.0f) { r = x; b = c; }
    else { r = c; b = x; }
    r += m; g += m; b += m;
}

}  // namespace pnr
