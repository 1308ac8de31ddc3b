// lattice.hpp -- exact closed form of the march's t-stepping loop (host + device; no HIP dependencies so that
// tests/native/ can compile it with g++).
//
// The reference advances the ray parameter only by   t <- fl(t + dt(t)),  dt(t) = clamp(t * dt_gamma, dt_min, dt_max)
// (raymarching.cu:389 after a sample, :399-401 `do { t += dt } while (t < tt)` past an empty cell).  The values a ray
// can visit therefore form a fixed sequence -- the ray's "t lattice" -- that does not depend on the occupancy grid;
// the grid only decides which lattice points are probed.  Skipping empty space exactly means landing on the right
// lattice point, i.e. evaluating that do/while loop for a distant `tt`.  For a constant step d (dt_gamma == 0, or the
// clamp saturated over the whole range) the loop has a closed form per binade:
//   for tc in [2^e, 2^(e+1)) with ulp u, fl(tc + d) = tc + D*u with D = round(d / u), as long as d/u is not an exact
//   tie and the sum stays inside the binade  =>  k steps add exactly k*D*u.
// lattice_advance() walks binade by binade with integer arithmetic and performs the step that crosses a binade
// boundary (and every step in a tie binade) as a real fp32 addition, so its result is bit-identical to the loop.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define PNR_HD __host__ __device__ __forceinline__
#else
#define PNR_HD inline
#endif

namespace pnr {

PNR_HD uint32_t lat_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
PNR_HD float lat_float(uint32_t b) { return __builtin_bit_cast(float, b); }

// Exactly:   do { prev = tc; tc += d; } while (tc < tt);   q = tc;       (d > 0 constant; fp32, round to nearest even)
PNR_HD void lattice_advance(float tc, float d, float tt, float& q, float& prev) {
    const uint32_t db = lat_bits(d);
    const int ed = (int)((db >> 23) & 0xffu);
    const uint32_t md = (db & 0x7fffffu) | 0x800000u;
    const bool d_ok = d > 0.0f && ed > 0 && ed < 255;
    for (;;) {
        const uint32_t tb = lat_bits(tc);
        const int e = (int)((tb >> 23) & 0xffu);
        const int s = e - ed;
        bool fast = d_ok && (tb >> 31) == 0 && e > 0 && e < 254 && s >= 1 && s <= 23;
        uint32_t D = 0;
        if (fast) {
            const uint32_t half = 1u << (s - 1), rem = md & ((1u << s) - 1u);
            if (rem == half) fast = false;  // d/u is an exact tie: the rounding direction depends on the parity of tc/u
            else D = (md + half) >> s;      // >= 1 because md >= 2^23 and s <= 23
        }
        if (!fast) {  // one plain step
            prev = tc;
            tc += d;
            if (!(tc < tt)) { q = tc; return; }
            continue;
        }
        const uint32_t Tc = (tb & 0x7fffffu) | 0x800000u;                       // tc = Tc * 2^(e-150)
        const float binade_end = lat_float((uint32_t)(e + 1) << 23);             // 2^(e+1-127)
        uint32_t TT;                                                             // target in units of u, saturated at the binade end
        if (!(tt < binade_end)) TT = 1u << 24;                                   // beyond this binade (or NaN: the loop below ends it)
        else if (!(tt > tc)) TT = Tc;
        else TT = (lat_bits(tt) & 0x7fffffu) | 0x800000u;                        // tc < tt < 2^(e+1): same binade
        if (tt != tt) { prev = tc; q = tc + d; return; }                         // NaN target: one step, `tc < tt` is false
        const uint32_t room = ((1u << 24) - 1u - Tc) / D;                        // steps that stay inside the binade
        uint32_t need = TT > Tc ? (TT - Tc + D - 1u) / D : 1u;
        if (need == 0) need = 1;
        if (need <= room) {
            prev = lat_float(((uint32_t)e << 23) | ((Tc + (need - 1u) * D) & 0x7fffffu));
            q = lat_float(((uint32_t)e << 23) | ((Tc + need * D) & 0x7fffffu));
            return;
        }
        // not reached inside this binade: go to its last lattice point (still < tt), then one real addition crosses over
        const float last = lat_float(((uint32_t)e << 23) | ((Tc + room * D) & 0x7fffffu));
        prev = last;
        tc = last + d;
        if (!(tc < tt)) { q = tc; return; }
    }
}

// Exactly:   for (i = 0; i < k; i++) tc += d;      (d > 0 constant) -- the k-th lattice point after tc, same binade-wise integer
// arithmetic as lattice_advance(): the wave-cooperative march tail hands lane k of a wave the k-th point of one ray's lattice.
PNR_HD float lattice_steps(float tc, float d, uint32_t k) {
    const uint32_t db = lat_bits(d);
    const int ed = (int)((db >> 23) & 0xffu);
    const uint32_t md = (db & 0x7fffffu) | 0x800000u;
    const bool d_ok = d > 0.0f && ed > 0 && ed < 255;
    while (k > 0) {
        const uint32_t tb = lat_bits(tc);
        const int e = (int)((tb >> 23) & 0xffu);
        const int s = e - ed;
        bool fast = d_ok && (tb >> 31) == 0 && e > 0 && e < 254 && s >= 1 && s <= 23;
        uint32_t D = 0;
        if (fast) {
            const uint32_t half = 1u << (s - 1), rem = md & ((1u << s) - 1u);
            if (rem == half) fast = false;
            else D = (md + half) >> s;
        }
        if (!fast) { tc += d; k--; continue; }
        const uint32_t Tc = (tb & 0x7fffffu) | 0x800000u;
        const uint32_t room = ((1u << 24) - 1u - Tc) / D;                         // steps that stay inside the binade
        if (k <= room) return lat_float(((uint32_t)e << 23) | ((Tc + k * D) & 0x7fffffu));
        tc = lat_float(((uint32_t)e << 23) | ((Tc + room * D) & 0x7fffffu));       // the binade's last lattice point ...
        tc += d;                                                                  // ... and one real addition across the boundary
        k -= room + 1u;
    }
    return tc;
}

// The general lattice walk (any dt_gamma): the reference loop itself, without probing.
PNR_HD void lattice_walk(float tc, float dt_gamma, float dt_min, float dt_max, float tt, float& q, float& prev) {
    do {
        prev = tc;
        const float dt = fminf(dt_max, fmaxf(dt_min, tc * dt_gamma));
        tc += dt;
    } while (tc < tt);
    q = tc;
}

}  // namespace pnr
