// CPU check of palettenerf_amd/csrc/lattice.hpp: lattice_advance() against the literal do/while loop it replaces.
// Built by tests/test_host_logic.py with g++ -O2 -ffp-contract=off.
#include "../../palettenerf_amd/csrc/lattice.hpp"
#include <math.h>
#include <stdint.h>

static uint64_t rng_state;
static uint32_t rng() {  // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 2685821237ull * 1000003ull) >> 32);
}
static float uniform(float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f); }

extern "C" int lattice_case(float tc, float d, float tt, float* q, float* prev, float* q_ref, float* prev_ref) {
    pnr::lattice_advance(tc, d, tt, *q, *prev);
    float t = tc, p = tc;
    do { p = t; t += d; } while (t < tt);
    *q_ref = t; *prev_ref = p;
    return pnr::lat_bits(*q) == pnr::lat_bits(*q_ref) && pnr::lat_bits(*prev) == pnr::lat_bits(*prev_ref);
}

// mode 0: generic d in [1e-4, 0.1); mode 1: d = 2*sqrt(3)/max_steps for max_steps in {256,512,1024,2048,4096} and 2*sqrt(3)*2^k/H;
// mode 2: d with few mantissa bits (exercises exact ties); returns the number of mismatching cases
extern "C" int lattice_fuzz(uint64_t seed, int n, int mode, float* bad /* [3] first failing (tc, d, tt) */) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 1;
    int mismatches = 0;
    for (int i = 0; i < n; i++) {
        float d;
        if (mode == 0) d = expf(uniform(logf(1e-4f), logf(0.1f)));
        else if (mode == 1) {
            const float two_sqrt3 = 2.0f * 1.7320508075688772f;
            const int k = (int)(rng() % 9u);
            d = k < 5 ? two_sqrt3 / (float)(256 << k) : two_sqrt3 * (float)(1 << (k - 5)) / 128.0f;
        } else {
            const uint32_t m = 0x800000u | ((rng() & 0x7u) << 20);  // 3 fraction bits
            d = ldexpf((float)m, -24 - (int)(rng() % 10u) - 3);
        }
        const float tc = expf(uniform(logf(0.01f), logf(40.0f)));
        const uint32_t kind = rng() % 8u;
        float tt;
        if (kind == 0) tt = tc - uniform(0.0f, 1.0f);            // target behind: exactly one step
        else if (kind == 1) tt = tc + d * uniform(0.0f, 3.0f);   // a cell or so
        else tt = tc + uniform(0.0f, kind < 5 ? 0.5f : 6.0f);    // long skips, several binades
        float q, p, qr, pr;
        if (!lattice_case(tc, d, tt, &q, &p, &qr, &pr)) {
            if (mismatches == 0 && bad) { bad[0] = tc; bad[1] = d; bad[2] = tt; }
            mismatches++;
        }
    }
    return mismatches;
}

// lattice_steps() against the literal loop: k plain additions.  Returns the number of mismatching cases.
extern "C" int lattice_steps_fuzz(uint64_t seed, int n, int mode, float* bad /* [3] first failing (tc, d, k) */) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 7;
    int mismatches = 0;
    for (int i = 0; i < n; i++) {
        float d;
        if (mode == 0) d = expf(uniform(logf(1e-4f), logf(0.1f)));
        else if (mode == 1) {
            const float two_sqrt3 = 2.0f * 1.7320508075688772f;
            const int k = (int)(rng() % 9u);
            d = k < 5 ? two_sqrt3 / (float)(256 << k) : two_sqrt3 * (float)(1 << (k - 5)) / 128.0f;
        } else {
            const uint32_t m = 0x800000u | ((rng() & 0x7u) << 20);
            d = ldexpf((float)m, -24 - (int)(rng() % 10u) - 3);
        }
        const float tc = expf(uniform(logf(0.01f), logf(40.0f)));
        const uint32_t k = rng() % ((rng() & 3u) ? 65u : 3000u);
        float t = tc;
        for (uint32_t j = 0; j < k; j++) t += d;
        const float got = pnr::lattice_steps(tc, d, k);
        if (pnr::lat_bits(got) != pnr::lat_bits(t)) {
            if (mismatches == 0 && bad) { bad[0] = tc; bad[1] = d; bad[2] = (float)k; }
            mismatches++;
        }
    }
    return mismatches;
}
