// traverse.h -- wave-uniform traversal of the 8-wide LBVH shared by the 1-NN and
// k-NN kernels.
//
// Tree.  Leaves are the 128-B lines of 8 Morton-consecutive points.  Above them
// sits a complete 8-ary tree: node ids are digit strings with a leading 1 (root
// = 1, child c of id = 8*id + c), so the level-k ids are [8^k, 2*8^k) and the
// ancestor j levels up is id >> 3j.  Node id owns one 256-B RECORD holding the
// boxes of its 8 children; records are stored level after level (full levels
// above, the used prefix of the last one), record_index(id) below.  The
// children of a last-level node (id >= leaf_first = 8^(m-1)) are the leaves
// 8*(id-leaf_first)+c.  Empty child slots carry the inverted box (+inf,-inf).
//
// A record is 4 sibling pairs of 12 floats {Amin.x,Bmin.x, Amin.y,Bmin.y, Amin.z,
// Bmin.z, Amax.x,Bmax.x, Amax.y,Bmax.y, Amax.z,Bmax.z} = 192 B, padded to 256 B.
// Three s_load_dwordx16 fetch it in ONE round trip (they are issued together and
// waited for once), and each lane tests two boxes per packed fp32 instruction
// (v_pk_add_f32 on {A,B} register pairs).
//
// Why 8-wide: the traversal is bound by dependent memory round trips (one wave
// = one outstanding record; measured ~1000 cycles per step at 10M points), not
// by ALU, so each round trip should resolve as many levels as possible: one
// record replaces three binary levels.
//
// A wave walks the tree once for its 64 queries.  All traversal state is
// wave-uniform and lives in SGPRs: the current node id and `pend`, a stack of
// 8-bit masks (one per level on the current path) of the sibling slots that
// were hit but not visited yet.  No per-lane stack, no LDS, no divergent memory
// access.
//
// Node test: each lane keeps the cube [lo, hi] = query -/+ rb, where rb is an upper
// bound of sqrt(its current best d2), and a box is entered when it overlaps the
// cube of ANY lane (Linf-distance(query, box) < rb).  Linf <= L2, so this never
// culls a box that holds a closer point; rb = sqrt(best) * (1 + 2^-21) and lo/hi
// are widened by one ulp, which keeps the test conservative under fp32 rounding
// (culled  =>  |q-p|_inf >= rb  =>  fl(d*d) >= best for every point p of the box).
// The overlap test is 6 compares per box, issued as a v_cmpx chain that narrows
// EXEC (measured on MI355X: plain fp32 VALU ~2.6 cycles/wave-instruction,
// v_pk_*_f32 ~5, v_max3 ~4 -- compares are the cheapest way to test a box).
// The exact fp32 d2 comparison happens only on leaf points.
#pragma once
#include "device_utils.h"

namespace mi {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f2* cf2_p;
typedef const __attribute__((address_space(4))) f16v* cf16_p;

constexpr int kRecordFloats = 64;  // 256 B: 4 sibling pairs of 12 floats + 16 floats of padding
constexpr int kPairStride = 12;
constexpr int kMaxLevels = 9;      // 8-bit masks in a 64-bit stack: 8 pushes

// (h - 1) / 7 for h = 8^k, exact via the inverse of 7 modulo 2^32
__host__ __device__ __forceinline__ uint32_t full_levels_below(uint32_t h) {
    return (h - 1u) * 0xB6DB6DB7u;
}

// storage index of node id's record: ids of level k are [8^k, 2*8^k)
__host__ __device__ __forceinline__ uint32_t record_index(uint32_t id) {
    const uint32_t h = 1u << (31 - __builtin_clz(id));
    return id - h + full_levels_below(h);
}

// per-lane search cube around the query
struct Cube {
    float lox, loy, loz, hix, hiy, hiz;
};

__device__ __forceinline__ float widen_up(float s) { return __builtin_fmaf(fabsf(s), 1.1920929e-7f, s); }
__device__ __forceinline__ float widen_down(float s) { return __builtin_fmaf(fabsf(s), -1.1920929e-7f, s); }

// best_d2 = +inf -> everything; best_d2 < 0 (invalid lane) -> nothing
__device__ __forceinline__ void set_cube(Cube& c, float qx, float qy, float qz, float best_d2) {
    const float rb = (best_d2 >= 0.0f) ? __builtin_amdgcn_sqrtf(best_d2) * 1.0000005f : -INFINITY;
    c.hix = widen_up(qx + rb);
    c.hiy = widen_up(qy + rb);
    c.hiz = widen_up(qz + rb);
    c.lox = widen_down(qx - rb);
    c.loy = widen_down(qy - rb);
    c.loz = widen_down(qz - rb);
}

// hit = 2*hit + (any lane's cube overlaps box), for boxes B then A of one sibling pair.
// A v_cmpx chain per box: EXEC shrinks to the lanes that still overlap; what is left
// non-zero is "some lane hits".  EXEC is saved/restored inside the statement.
__device__ __forceinline__ void pair_hits(uint32_t& hit, const Cube& c, float amnx, float amny,
                                          float amnz, float amxx, float amxy, float amxz, float bmnx,
                                          float bmny, float bmnz, float bmxx, float bmxy, float bmxz) {
    uint64_t sv;
    asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "v_cmpx_lt_f32_e32 %[bmnx], %[hix]\n\t"
            "v_cmpx_lt_f32_e32 %[bmny], %[hiy]\n\t"
            "v_cmpx_lt_f32_e32 %[bmnz], %[hiz]\n\t"
            "v_cmpx_gt_f32_e32 %[bmxx], %[lox]\n\t"
            "v_cmpx_gt_f32_e32 %[bmxy], %[loy]\n\t"
            "v_cmpx_gt_f32_e32 %[bmxz], %[loz]\n\t"
            "s_cmp_lg_u64 exec, 0\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_addc_u32 %[hit], %[hit], %[hit]\n\t"
            "v_cmpx_lt_f32_e32 %[amnx], %[hix]\n\t"
            "v_cmpx_lt_f32_e32 %[amny], %[hiy]\n\t"
            "v_cmpx_lt_f32_e32 %[amnz], %[hiz]\n\t"
            "v_cmpx_gt_f32_e32 %[amxx], %[lox]\n\t"
            "v_cmpx_gt_f32_e32 %[amxy], %[loy]\n\t"
            "v_cmpx_gt_f32_e32 %[amxz], %[loz]\n\t"
            "s_cmp_lg_u64 exec, 0\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_addc_u32 %[hit], %[hit], %[hit]\n\t"
            : [hit] "+s"(hit), [sv] "=&s"(sv)
            : [amnx] "s"(amnx), [amny] "s"(amny), [amnz] "s"(amnz), [amxx] "s"(amxx), [amxy] "s"(amxy),
              [amxz] "s"(amxz), [bmnx] "s"(bmnx), [bmny] "s"(bmny), [bmnz] "s"(bmnz), [bmxx] "s"(bmxx),
              [bmxy] "s"(bmxy), [bmxz] "s"(bmxz), [lox] "v"(c.lox), [loy] "v"(c.loy), [loz] "v"(c.loz),
              [hix] "v"(c.hix), [hiy] "v"(c.hiy), [hiz] "v"(c.hiz)
            : "vcc", "scc");
}

// leaf(L): processes leaf L for every lane and may shrink the lane's cube.
// Returns the number of records visited (census).
template <class LeafFn>
__device__ __forceinline__ uint32_t traverse_wide(const float* records_g, uint32_t leaf_first,
                                                  const Cube& cube, uint32_t max_steps,
                                                  LeafFn&& leaf) {
    const cf16_p recs = (cf16_p)(uintptr_t)records_g;
    uint32_t id = 1u, steps = 0u;
    uint64_t pend = 0ull;
    while (steps++ < max_steps) {
        id = __builtin_amdgcn_readfirstlane(id);
        const cf16_p rec = recs + (size_t)record_index(id) * (kRecordFloats / 16);
        // the whole record in one round trip: 3 x s_load_dwordx16, then one wait
        const f16v r0 = rec[0], r1 = rec[1], r2 = rec[2];
        __builtin_amdgcn_sched_barrier(0);
        float w[48];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            w[e] = r0[e];
            w[16 + e] = r1[e];
            w[32 + e] = r2[e];
        }
        uint32_t hit = 0u;
#pragma unroll
        for (int p = 3; p >= 0; --p) {  // children 7..0, so that child c ends up in bit c
            const float* b = w + p * kPairStride;
            pair_hits(hit, cube, b[0], b[2], b[4], b[6], b[8], b[10], b[1], b[3], b[5], b[7], b[9], b[11]);
        }
        bool pop = true;
        if (id >= leaf_first) {  // children are leaves; later ones were tested before earlier
                                 // ones shrank the bounds: at worst a wasted leaf
            const uint32_t lbase = (id - leaf_first) * 8u;
            while (hit) {
                const uint32_t c = (uint32_t)__builtin_ctz(hit);
                hit &= hit - 1u;
                leaf(lbase + c);
            }
        } else if (hit) {
            const uint32_t c = (uint32_t)__builtin_ctz(hit);
            pend = (pend << 8) | (uint64_t)(hit & (hit - 1u));
            id = id * 8u + c;
            pop = false;
        }
        if (pop) {
            if (pend == 0ull) break;
            const uint32_t j = (uint32_t)__builtin_ctzll(pend) >> 3;  // levels to climb
            id >>= 3u * j;
            pend >>= 8u * j;
            const uint32_t m = (uint32_t)pend & 255u;
            id = (id & ~7u) | (uint32_t)__builtin_ctz(m);  // next pending sibling at that level
            pend = (pend & ~255ull) | (uint64_t)(m & (m - 1u));
        }
    }
    return steps;
}

}  // namespace mi
