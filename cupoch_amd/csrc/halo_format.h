// halo_format.h -- the layout of a leaf's halo (leaf_halo.h builds it, nn_search.h reads it): lines, reaches and how
// the reaches are packed into the two spare words of the leaf's region record.
#pragma once
#include "device_utils.h"

namespace mi {

constexpr int kHaloLines = 8;            // halo lines per leaf (leaf_halo.h): the 64 nearest points of other leaves, 8 per line, in rings
constexpr int kHaloLineFloats = 32;      // x[8] y[8] z[8] slot[8]: a leaf line with the points' slots in its fourth row
constexpr float kHaloUnit = 1.0f / 64.0f;
// How many of the eight lines are STORED (a build-time knob for one measurement, VERDICT r4 next-6: -DMI_HALO_STORED=4
// keeps the nearest 32 points, 512 B per leaf; the reaches of the lines that are not stored equal the last stored
// one's, so a cube that needs more walks).  8 in the shipped library.
#ifndef MI_HALO_STORED
#define MI_HALO_STORED 8
#endif
constexpr int kHaloStored = MI_HALO_STORED;
static_assert(kHaloStored >= 1 && kHaloStored <= kHaloLines, "MI_HALO_STORED: 1 .. 8");

// The reaches of a leaf's eight halo lines travel in the two spare words of its region record, as 6-bit
// fractions q_k of the bound (the reach a line has when no point lies behind it): word A = q0 .. q4 from bit 0,
// the low two bits of q7 on top; word B = q5, q6 from bit 0, the high four bits of q7 from bit 12, the bound
// -- the upper 16 bits of an fp32, rounded down -- on top; reach k = bound / 64 * q_k, rounded down when packed.
__host__ __device__ __forceinline__ uint32_t halo_reach_fraction(uint32_t wa, uint32_t wb, int k) {
    return (k < 5) ? ((wa >> (6 * k)) & 63u) : ((k < 7) ? ((wb >> (6 * (k - 5))) & 63u) : ((wa >> 30) | (((wb >> 12) & 15u) << 2)));
}
// How many lines a cube that pokes out of the region by `over` has to read: 1 .. 8, or 9: beyond them all.
__device__ __forceinline__ uint32_t halo_lines_needed(float wa_f, float wb_f, float over) {
    const uint32_t wa = __float_as_uint(wa_f), wb = __float_as_uint(wb_f);
    const float unit = __uint_as_float(wb & 0xffff0000u) * kHaloUnit;
    uint32_t n = 1u;
#pragma unroll
    for (int k = 0; k < kHaloLines; ++k)
        n += (over < unit * (float)halo_reach_fraction(wa, wb, k)) ? 0u : 1u;  // (NaN: every line and then the walk)
    return n;
}

}  // namespace mi
