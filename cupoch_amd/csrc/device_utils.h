// device_utils.h -- shared device helpers for the gfx950 ICP kernels.
//
// Everything here assumes wave64 (CDNA4).  Compiled with -ffp-contract=off:
// fused multiply-adds appear only where written explicitly, so the per-point
// fp32 expressions are exactly the ones DESIGN.md documents (and the CPU
// oracle evaluates):  d2 = fma(dz,dz, fma(dy,dy, dx*dx)),
//                     R*p+t = fma(R02,z, fma(R01,y, R00*x)) + t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The hand-offs that replace a release fence by "returnless atomics drained with s_waitcnt vmcnt(0) before the ticket"
// (kd_build.h tree_scale, reduce.h) rely on the gfx9 family's counters: stores and returnless atomics are tracked by
// vmcnt there (gfx10+ has a separate vscnt).  This library is written for gfx950 only; anything else must not compile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmi_icp is written for gfx950 (CDNA4) only: wave64, gfx9 vmcnt semantics, 160 KB LDS"
#endif

namespace mi {

constexpr int kWave = 64;
constexpr int kLeaf = 8;          // points per LBVH leaf
constexpr int kLeafFloats = 32;   // x[8] y[8] z[8] region[8] = one 128-B line
// The fourth row of a leaf line is the leaf's REGION record: lo.xyz, A | hi.xyz, B (kd_build.h); A, B: the reaches of its
// halo lines (leaf_halo.h).  (Up to round 3 it held the points' original indices and the regions had an array of their
// own: the seeded search then touched two cache lines per leaf -- 160 bytes of HBM traffic instead of 128.  The indices,
// needed only where results leave the library, are an array indexed by slot now: tidx.)
constexpr int kLeafRegFloats = 8;            // floats of a region record
constexpr int kLeafRegOffset = 24;           // its place inside the leaf line
constexpr int kLeafRegStride = kLeafFloats;  // from one leaf's record to the next: `lreg` pointers are tblk + kLeafRegOffset

// Row-major 3x4 rigid transform passed by value as a kernel argument (SGPRs).
struct Xform {
    float r00, r01, r02, t0;
    float r10, r11, r12, t1;
    float r20, r21, r22, t2;
};

// Pointers in the constant address space: a wave-uniform load through one of
// these is selected as a scalar (s_load_*) instruction.
typedef const __attribute__((address_space(4))) float* cfloat_p;
typedef const __attribute__((address_space(4))) uint32_t* cuint_p;

__device__ __forceinline__ float sq3(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ void rotate(const Xform& T, float x, float y, float z, float& ox,
                                       float& oy, float& oz) {
    ox = __builtin_fmaf(T.r02, z, __builtin_fmaf(T.r01, y, T.r00 * x));
    oy = __builtin_fmaf(T.r12, z, __builtin_fmaf(T.r11, y, T.r10 * x));
    oz = __builtin_fmaf(T.r22, z, __builtin_fmaf(T.r21, y, T.r20 * x));
}

__device__ __forceinline__ void xform_point(const Xform& T, float x, float y, float z,
                                            float& ox, float& oy, float& oz) {
    rotate(T, x, y, z, ox, oy, oz);
    ox += T.t0;
    oy += T.t1;
    oz += T.t2;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// One DPP hop of a 64-bit value (two v_mov_b32 with a DPP modifier); lanes without a
// source -- or rows masked out by ROW_MASK -- receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_hop(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int kWaveSumLane = 63;  // wave_sum leaves the total in the LAST lane

// fp64 wave reduction on the DPP network: row_shr 1,2,4,8 scan each row of 16 lanes,
// row_bcast:15 / row_bcast:31 carry the row totals across; 18 VALU instructions and no
// LDS traffic (the ds_bpermute form -- 12 LDS ops per value -- made the 30-value epilogue
// of reduce_kernel cost ~19 us per CU).
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_hop<0x111, 0xf>(v);  // row_shr:1
    v += dpp_hop<0x112, 0xf>(v);  // row_shr:2
    v += dpp_hop<0x114, 0xf>(v);  // row_shr:4
    v += dpp_hop<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_hop<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_hop<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;                     // valid in lane 63
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

// Workgroups are dealt to XCDs round-robin (block b -> XCD b % 8).  Remap so
// that each XCD walks one contiguous eighth of the (Morton-ordered) work list
// and neighbouring packets share that XCD's L2.  Speed only; any placement is
// correct.  Returns false when the logical index is past the end.
__device__ __forceinline__ bool xcd_remap(uint32_t nblocks, uint32_t& logical) {
    const uint32_t b = blockIdx.x;
    const uint32_t chunk = (nblocks + 7u) >> 3;  // grid = chunk * 8
    logical = (b & 7u) * chunk + (b >> 3);
    return logical < nblocks;
}

}  // namespace mi
