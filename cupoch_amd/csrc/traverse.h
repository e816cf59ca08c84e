// traverse.h -- wave-uniform traversal of the implicit 8-ary tree shared by the 1-NN and
// k-NN kernels.
//
// Tree.  Leaves are the 128-B lines of 8 consecutive slots of the target's kd order.  Above them
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
// waited for once).  Floats 48..53 hold the node's REGION -- the part of space free of points
// of any other node -- and float 54 the flag that it is valid (lbvh.h store_own; the points'
// box and 0 otherwise), for the early stop of the seeded search below.
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
// v_pk_*_f32 ~5, v_max3 ~4 -- compares are the cheapest way to test a box); each
// lane collects its 8 results in a VGPR and one DPP OR-reduction per record turns
// them into the wave-uniform hit mask.
// The exact fp32 d2 comparison happens only on leaf points.
#pragma once
#include <type_traits>

#include "device_utils.h"

namespace mi {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f2* cf2_p;
typedef const __attribute__((address_space(4))) f16v* cf16_p;

constexpr int kRecordFloats = 64;  // 256 B: 4 sibling pairs of 12 floats + 16 floats of padding
constexpr int kPairStride = 12;
constexpr int kMaxLevels = 9;      // 8-bit masks in a 64-bit stack: 8 pushes
constexpr int kRecordCap = 56, kRecordCap2 = 57;  // floats of the ROOT record's padding: the walk's cap and its square (kd_build.h tree_scale)
constexpr int kRecordNear2 = 58;                  // ... and the squared radius of a first round from the root (nn_search.h)

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

// One box: a v_cmpx chain narrows EXEC to the lanes whose cube overlaps it (on gfx9-family
// parts v_cmpx also leaves that mask in VCC); EXEC is restored and the lane's own
// history is shifted left with the hit as carry-in.  One SALU instruction per box: the
// earlier s_cmp_lg_u64 / s_addc_u32 tail made the loop SALU-bound.
#define MI_BOX_CHAIN(mnx, mny, mnz, mxx, mxy, mxz)      \
    "v_cmpx_lt_f32_e32 %[" #mnx "], %[hix]\n\t"          \
    "v_cmpx_lt_f32_e32 %[" #mny "], %[hiy]\n\t"          \
    "v_cmpx_lt_f32_e32 %[" #mnz "], %[hiz]\n\t"          \
    "v_cmpx_gt_f32_e32 %[" #mxx "], %[lox]\n\t"          \
    "v_cmpx_gt_f32_e32 %[" #mxy "], %[loy]\n\t"          \
    "v_cmpx_gt_f32_e32 %[" #mxz "], %[loz]\n\t"          \
    "s_mov_b64 exec, %[sv]\n\t"                          \
    "v_addc_co_u32_e32 %[vm], vcc, %[vm], %[vm], vcc\n\t"

// vm = 4*vm + 2*(this lane's cube overlaps box B) + (... box A), for one sibling pair.
// `sv` must hold the wave's EXEC.
__device__ __forceinline__ void pair_hits(uint32_t& vm, uint64_t sv, const Cube& c, float amnx, float amny,
                                          float amnz, float amxx, float amxy, float amxz, float bmnx,
                                          float bmny, float bmnz, float bmxx, float bmxy, float bmxz) {
    asm volatile(MI_BOX_CHAIN(bmnx, bmny, bmnz, bmxx, bmxy, bmxz) MI_BOX_CHAIN(amnx, amny, amnz, amxx, amxy, amxz)
                 : [vm] "+v"(vm)
                 : [sv] "s"(sv), [amnx] "s"(amnx), [amny] "s"(amny), [amnz] "s"(amnz), [amxx] "s"(amxx),
                   [amxy] "s"(amxy), [amxz] "s"(amxz), [bmnx] "s"(bmnx), [bmny] "s"(bmny), [bmnz] "s"(bmnz),
                   [bmxx] "s"(bmxx), [bmxy] "s"(bmxy), [bmxz] "s"(bmxz), [lox] "v"(c.lox), [loy] "v"(c.loy),
                   [loz] "v"(c.loz), [hix] "v"(c.hix), [hiy] "v"(c.hiy), [hiz] "v"(c.hiz)
                 : "vcc");
}

// OR of the lanes' masks, wave-uniform: an inclusive OR-scan along each row of 16 lanes on
// the DPP network, the row results carried across, read from the last lane.  The s_nops are
// the wait states gfx9 requires (VALU write of EXEC -> DPP: 5; VALU write of a VGPR -> DPP
// read of it: 2); inline asm is not covered by the compiler's hazard recogniser.
__device__ __forceinline__ uint32_t wave_or_mask(uint32_t vm) {
    uint32_t out;
    asm volatile(
            "s_nop 3\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "s_nop 1\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "s_nop 1\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "s_nop 1\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "s_nop 1\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_or_b32_dpp %[vm], %[vm], %[vm] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_readlane_b32 %[out], %[vm], 63\n\t"
            : [out] "=s"(out), [vm] "+v"(vm));
    return out;
}

// Returns the number of records visited (census).
//
// The loop is bound by the CU's single scalar ALU (measured: <= 0.97 SALU
// instructions per cycle per CU, shared by the 4 SIMDs, against 1.43 VALU
// wave-instructions; profiles/r01_ubench_salu_rate.txt), so the wave-uniform
// bookkeeping is kept to a minimum:
//  * the record's byte offset is 32-bit (id + off) << 8 with `off` =
//    record_index(id) - id tracked per level (8*off + 1 one level down, an
//    arithmetic shift back up), used as the SGPR offset of the s_loads;
//  * no step limit: the walk consumes a finite stack of sibling bits and cannot
//    cycle, whatever the boxes contain;
//  * pops take the next sibling from the same find-first-bit that found the level.
//
// leaf_rec(lbase, vm, hit) is called once per visited LEAF-LEVEL record: lbase = index of
// its first leaf, vm = this lane's 8-bit mask of the leaves its cube overlaps, hit = the
// OR over the wave.
//
// START NODE.  `start` is the node the walk begins at: 1 (the root) for a plain top-down
// search -- what the unseeded first pass and the k-NN kernels use -- or any other node, in
// which case its subtree is searched first and the walk then climbs: at every ancestor the
// other seven children are tested (and searched where hit), until the root is done -- or
// until ALL lanes' cubes lie inside the REGION of the subtree just completed (record floats
// 48..54, lbvh.h store_own): every point outside the subtree lies on or beyond a face of
// that region, hence outside every cube, hence cannot be closer than what the lanes already
// hold.  The seeded ICP search uses the refinement of this further down (traverse_seeded:
// several starts, lanes retire one by one).
__device__ __forceinline__ bool cubes_inside(const Cube& c, uint64_t full_exec, float mnx, float mny, float mnz,
                                             float mxx, float mxy, float mxz) {
    uint32_t all;
    asm volatile(
            "v_cmpx_le_f32_e32 %[mnx], %[lox]\n\t"
            "v_cmpx_le_f32_e32 %[mny], %[loy]\n\t"
            "v_cmpx_le_f32_e32 %[mnz], %[loz]\n\t"
            "v_cmpx_ge_f32_e32 %[mxx], %[hix]\n\t"
            "v_cmpx_ge_f32_e32 %[mxy], %[hiy]\n\t"
            "v_cmpx_ge_f32_e32 %[mxz], %[hiz]\n\t"
            "s_cmp_eq_u64 exec, %[sv]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_cselect_b32 %[all], 1, 0\n\t"
            : [all] "=s"(all)
            : [sv] "s"(full_exec), [mnx] "s"(mnx), [mny] "s"(mny), [mnz] "s"(mnz), [mxx] "s"(mxx), [mxy] "s"(mxy),
              [mxz] "s"(mxz), [lox] "v"(c.lox), [loy] "v"(c.loy), [loz] "v"(c.loz), [hix] "v"(c.hix),
              [hiy] "v"(c.hiy), [hiz] "v"(c.hiz)
            : "vcc", "scc");
    return all != 0u;
}

// The hit child (bit of `hit`, wave-uniform) that the most lanes' cubes overlap (vm: this lane's 8 bits); the lowest
// such child among equals.
__device__ __forceinline__ uint32_t most_wanted_child(uint32_t vm, uint32_t hit) {
    uint32_t first = (uint32_t)__builtin_ctz(hit);
    if ((hit & (hit - 1u)) != 0u) {
        uint32_t most = 0u;
#pragma unroll
        for (uint32_t c = 0; c < 8u; ++c) {
            if (((hit >> c) & 1u) == 0u) continue;  // wave-uniform
            const uint32_t n = (uint32_t)__popcll(__ballot(((vm >> c) & 1u) != 0u));
            if (n > most) {
                most = n;
                first = c;
            }
        }
    }
    return first;
}

template <class LeafRecFn>
__device__ __forceinline__ uint32_t traverse_from(const float* records_g, uint32_t leaf_first, uint32_t start,
                                                  const Cube& cube, LeafRecFn&& leaf_rec) {
    typedef const __attribute__((address_space(4))) char* cchar_p;
    const cchar_p base = (cchar_p)(uintptr_t)records_g;
    uint32_t id = __builtin_amdgcn_readfirstlane(start), steps = 0u;
    // off = record_index(id) - id of id's level: -1 at the root, (8^k - 1)/7 - 8^k at level k
    const uint32_t start_h = 1u << (31 - __builtin_clz(id));
    int32_t off = (int32_t)(full_levels_below(start_h) - start_h);
    uint32_t top = id;      // the subtree being completed
    int32_t top_off = off;
    uint32_t skip = 8u;     // child of `top` that is already done (8: none)
    uint64_t pend = 0ull;
    float ownv = 0.0f;  // lane k < 8: float 48 + k of top's record
    const uint64_t full_exec = __builtin_amdgcn_read_exec();
    for (;;) {
        ++steps;
        id = __builtin_amdgcn_readfirstlane(id);
        off = __builtin_amdgcn_readfirstlane(off);
        const uint32_t byte_off = (id + (uint32_t)off) << 8;  // kRecordFloats * 4 = 256
        const cf16_p rec = (cf16_p)(base + byte_off);
        // the whole record in one round trip: 3 x s_load_dwordx16, then one wait.  The search is
        // bound by these dependent round trips (~1 us each from L2 / Infinity Cache), so on the
        // climbing path -- id == top -- the same round trip also brings top's own box and touches
        // the four lines of its PARENT's record, which the next step of the climb will read.
        const f16v r0 = rec[0], r1 = rec[1], r2 = rec[2];
        if (id == top) {
            // ONE vector load, consumed only when the subtree is complete: lanes 0..7 fetch top's
            // own box + flag (floats 48..55 of its record), the other lanes touch the four lines
            // of the parent's record.  (As scalar loads these sat in 12 SGPRs across the whole
            // descent and made the compiler spill and wait.)
            const uint32_t poff = (id > 1u) ? (((id >> 3) + (uint32_t)(off >> 3)) << 8) : byte_off;
            const uint32_t ln = (uint32_t)lane_id();
            const uint32_t boff = (ln < 8u) ? (byte_off + 192u + 4u * ln) : (poff + 64u * (ln & 3u));
            ownv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(records_g) + boff);
        }
        __builtin_amdgcn_sched_barrier(0);
        float w[48];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            w[e] = r0[e];
            w[16 + e] = r1[e];
            w[32 + e] = r2[e];
        }
        uint32_t vm = 0u;
#pragma unroll
        for (int p = 3; p >= 0; --p) {  // children 7..0, so that child c ends up in bit c
            const float* b = w + p * kPairStride;
            pair_hits(vm, full_exec, cube, b[0], b[2], b[4], b[6], b[8], b[10], b[1], b[3], b[5], b[7], b[9],
                      b[11]);
        }
        uint32_t hit = wave_or_mask(vm);
        if (id == top && skip < 8u) hit &= ~(1u << skip);  // climbing: that child's subtree is done
        if (id < leaf_first) {
            if (hit) {
                // Descend into the hit child that the MOST lanes' cubes overlap -- the one the packet's queries sit
                // in -- and remember the others: what is found there shrinks the cubes, and a sibling that was hit
                // by the large cubes costs one record when its turn comes instead of its subtree.  (In child order
                // the walk from the root went through half of the far subtrees before the near one.)
                const uint32_t first = most_wanted_child(vm, hit);
                pend = (pend << 8) | (uint64_t)(hit & ~(1u << first));
                id = id * 8u + first;
                off = off * 8 + 1;
                continue;
            }
        } else if (hit) {
            // (a callback that takes a fourth argument also gets the record's 8 child boxes)
            if constexpr (std::is_invocable_v<LeafRecFn&, uint32_t, uint32_t, uint32_t, const float(&)[48]>)
                leaf_rec((id - leaf_first) * 8u, vm, hit, w);
            else
                leaf_rec((id - leaf_first) * 8u, vm, hit);
        }
        if (pend != 0ull) {
            const uint32_t z = (uint32_t)__builtin_ctzll(pend);  // lowest pending sibling, 8 bits per level
            const uint32_t j3 = (z >> 3) * 3u;                   // 3 * levels to climb
            pend >>= (z & 56u);
            id = ((id >> j3) & ~7u) | (z & 7u);
            off >>= j3;  // arithmetic: off_k = 8^j off_(k-j) + (8^j-1)/7, the remainder drops out
            // clear that sibling: after the shift it sits in the low byte
            const uint32_t lo = (uint32_t)pend;
            pend = (pend & 0xffffffff00000000ull) | (uint64_t)(lo & (lo - 1u));
            continue;
        }
        // ---- the subtree of `top` is complete
        top = __builtin_amdgcn_readfirstlane(top);
        top_off = __builtin_amdgcn_readfirstlane(top_off);
        if (top == 1u) break;
        // own box + flag of `top` (fetched when top was visited)
        if (__builtin_amdgcn_readlane(__float_as_int(ownv), 6) != 0 &&
            cubes_inside(cube, full_exec, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 0)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 1)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 2)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 3)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 4)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 5))))
            break;
        skip = top & 7u;
        top >>= 3;
        top_off >>= 3;  // off_(k-1) = (off_k - 1) / 8, exact; the arithmetic shift floors to it
        id = top;
        off = top_off;
    }
    return steps;
}

// SEEDED SEARCH, several starts.  my_node: the leaf-level node of the lane's previous match
// (0: none).  A packet of 64 match-ordered queries usually straddles two or three leaf-level
// nodes; starting at one of them and climbing to their common ancestor costs the ancestor's
// record (and more when the boundary is a high-level one).  Instead the walk starts at the
// seed node of the first lane, and when that subtree is complete:
//   * every lane whose cube lies inside the completed subtree's box RETIRES (flagged boxes
//     only, see above): nothing outside the subtree can be closer than what it holds, and
//     everything inside that overlaps its cube has been queued.  Its cube becomes empty,
//     so it takes no further part in box tests;
//   * no lane left: done;
//   * a lane is left whose own seed node has not been a start yet (at most kSeedStarts
//     starts): that node's record is next -- all remaining lanes are tested against it;
//   * otherwise the climb of traverse_from, from the last start.  Leaf-level nodes that
//     were starts are masked out of their parents' hit masks: every lane still active was
//     tested against them with a cube at least as large as its present one.
// Converged iterations end after one record per distinct seed node (2.x instead of 4.0
// records per packet on the 10M bench).
constexpr uint32_t kSeedStarts = 3u;

template <class LeafRecFn>
__device__ __forceinline__ uint32_t traverse_seeded(const float* records_g, uint32_t leaf_first, uint32_t my_node,
                                                    Cube& cube, bool& retired, LeafRecFn&& leaf_rec) {
    typedef const __attribute__((address_space(4))) char* cchar_p;
    const cchar_p base = (cchar_p)(uintptr_t)records_g;
    const int32_t leaf_off = (int32_t)(full_levels_below(leaf_first) - leaf_first);
    uint64_t seeds = __ballot(my_node != 0u && !retired);  // lanes whose seed node has not been a start
    uint32_t id = 1u, steps = 0u;
    int32_t off = -1;
    if (seeds != 0ull) {
        id = (uint32_t)__builtin_amdgcn_readlane((int)my_node, (int)__builtin_ctzll(seeds));
        off = leaf_off;
        seeds &= ~__ballot(my_node == id);
    }
    uint32_t top = id;
    int32_t top_off = off;
    uint32_t skip = 8u;
    uint64_t pend = 0ull;
    // starts searched before the last one: parent id and the hit-mask that removes them there
    uint32_t dpar0 = 0u, dpar1 = 0u, dmask0 = ~0u, dmask1 = ~0u, ndone = 0u;
    bool climbing = false;
    float ownv = 0.0f;
    const uint64_t full_exec = __builtin_amdgcn_read_exec();
    for (;;) {
        ++steps;
        id = __builtin_amdgcn_readfirstlane(id);
        off = __builtin_amdgcn_readfirstlane(off);
        const uint32_t byte_off = (id + (uint32_t)off) << 8;
        const cf16_p rec = (cf16_p)(base + byte_off);
        const f16v r0 = rec[0], r1 = rec[1], r2 = rec[2];
        if (id == top) {  // own box + flag of top (lanes 0..7), parent's lines warmed by the others
            const uint32_t poff = (id > 1u) ? (((id >> 3) + (uint32_t)(off >> 3)) << 8) : byte_off;
            const uint32_t ln = (uint32_t)lane_id();
            const uint32_t boff = (ln < 8u) ? (byte_off + 192u + 4u * ln) : (poff + 64u * (ln & 3u));
            ownv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(records_g) + boff);
        }
        __builtin_amdgcn_sched_barrier(0);
        float w[48];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            w[e] = r0[e];
            w[16 + e] = r1[e];
            w[32 + e] = r2[e];
        }
        uint32_t vm = 0u;
#pragma unroll
        for (int p = 3; p >= 0; --p) {
            const float* b = w + p * kPairStride;
            pair_hits(vm, full_exec, cube, b[0], b[2], b[4], b[6], b[8], b[10], b[1], b[3], b[5], b[7], b[9],
                      b[11]);
        }
        uint32_t hit = wave_or_mask(vm);
        if (id == top && skip < 8u) hit &= ~(1u << skip);
        if (climbing) {  // starts already searched, as children of this node (the last one is `skip`)
            if (dpar0 == id) hit &= dmask0;
            if (dpar1 == id) hit &= dmask1;
        }
        if (id < leaf_first) {
            if (hit) {  // (in child order: the walk starts at the seed's node, where the lanes' answers mostly are --
                        // most_wanted_child here measured nothing on the noisy passes and the transient)
                pend = (pend << 8) | (uint64_t)(hit & (hit - 1u));
                id = id * 8u + (uint32_t)__builtin_ctz(hit);
                off = off * 8 + 1;
                continue;
            }
        } else if (hit) {
            leaf_rec((id - leaf_first) * 8u, vm, hit);
        }
        if (pend != 0ull) {
            const uint32_t z = (uint32_t)__builtin_ctzll(pend);
            const uint32_t j3 = (z >> 3) * 3u;
            pend >>= (z & 56u);
            id = ((id >> j3) & ~7u) | (z & 7u);
            off >>= j3;
            const uint32_t lo = (uint32_t)pend;
            pend = (pend & 0xffffffff00000000ull) | (uint64_t)(lo & (lo - 1u));
            continue;
        }
        // ---- the subtree of `top` is complete
        top = __builtin_amdgcn_readfirstlane(top);
        top_off = __builtin_amdgcn_readfirstlane(top_off);
        if (top == 1u) break;
        if (__builtin_amdgcn_readlane(__float_as_int(ownv), 6) != 0) {
            const float mnx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 0));
            const float mny = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 1));
            const float mnz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 2));
            const float mxx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 3));
            const float mxy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 4));
            const float mxz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ownv), 5));
            const bool inside = mnx <= cube.lox && mny <= cube.loy && mnz <= cube.loz && mxx >= cube.hix &&
                                mxy >= cube.hiy && mxz >= cube.hiz;
            if (inside) {  // (an empty cube is inside everything)
                retired = true;
                cube.lox = cube.loy = cube.loz = INFINITY;
                cube.hix = cube.hiy = cube.hiz = -INFINITY;
            }
        }
        const uint64_t active = __ballot(!retired);
        if (active == 0ull) break;
        seeds &= active;
        if (top >= leaf_first && seeds != 0ull && ndone + 1u < kSeedStarts) {
            // a start, and a lane still waits for its own: remember this one, take that one next
            if (ndone == 0u) {
                dpar0 = top >> 3;
                dmask0 = ~(1u << (top & 7u));
            } else {
                dpar1 = top >> 3;
                dmask1 = ~(1u << (top & 7u));
            }
            ++ndone;
            top = (uint32_t)__builtin_amdgcn_readlane((int)my_node, (int)__builtin_ctzll(seeds));
            seeds &= ~__ballot(my_node == top);
            top_off = leaf_off;
            skip = 8u;
            id = top;
            off = top_off;
            continue;
        }
        climbing = true;
        skip = top & 7u;
        top >>= 3;
        top_off >>= 3;
        id = top;
        off = top_off;
    }
    return steps;
}

// THE PER-LANE WALK (for queries that do not belong to their packet: knn_normals.h knn_walks_alone, nn_search.h
// kCapped).  Every lane with `solo` set searches the whole tree for itself: its own node id and stack of
// pending siblings (the wave-uniform walk's scheme, in vector registers), records and leaves fetched by the lane,
// a box entered when its L2 distance from the query -- formed with the same rounding as the points' distances, so
// never larger than any of them -- is below the lane's bound, the nearest hit child first.  offer(L): the lane's
// leaf L to its list (called in divergent code: per-lane work only).
template <class BoundFn, class OfferFn>
__device__ __forceinline__ void solo_walk(const float* __restrict__ records_g, uint32_t leaf_first, bool solo,
                                              float qx, float qy, float qz, BoundFn&& bound2, OfferFn&& offer) {
    uint32_t id = 1u;
    int32_t off = -1;
    uint64_t pend = 0ull;
    // (a query with a NaN coordinate has no neighbours and must not walk: fmaxf drops the NaN, every gap below comes out
    // 0, every slot -- the empty ones with their inverted boxes too -- counts as hit, and the lane descends into records
    // that do not exist: a memory fault until round 6, found by a source with NaN points under a radius beyond the cap)
    bool on = solo && qx == qx && qy == qy && qz == qz;
    while (__ballot(on) != 0ull) {
        uint32_t hit = 0u, nearest = 0u;
        float dnear = INFINITY;
        if (on) {
            const float4* rec = reinterpret_cast<const float4*>(records_g + ((size_t)(id + (uint32_t)off) << 6));
            const float w2 = bound2();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 a = rec[3 * p], b = rec[3 * p + 1], c = rec[3 * p + 2];
                // {Amin.x,Bmin.x,Amin.y,Bmin.y} {Amin.z,Bmin.z,Amax.x,Bmax.x} {Amax.y,Bmax.y,Amax.z,Bmax.z}
                const float gax = fmaxf(fmaxf(a.x - qx, qx - b.z), 0.0f), gay = fmaxf(fmaxf(a.z - qy, qy - c.x), 0.0f);
                const float gaz = fmaxf(fmaxf(b.x - qz, qz - c.z), 0.0f);
                const float gbx = fmaxf(fmaxf(a.y - qx, qx - b.w), 0.0f), gby = fmaxf(fmaxf(a.w - qy, qy - c.y), 0.0f);
                const float gbz = fmaxf(fmaxf(b.y - qz, qz - c.w), 0.0f);
                const float da = sq3(gax, gay, gaz), db = sq3(gbx, gby, gbz);  // (an empty slot's inverted box: +inf)
                if (da < w2) {
                    hit |= 1u << (2 * p);
                    if (da < dnear) {
                        dnear = da;
                        nearest = 2u * p;
                    }
                }
                if (db < w2) {
                    hit |= 2u << (2 * p);
                    if (db < dnear) {
                        dnear = db;
                        nearest = 2u * p + 1u;
                    }
                }
            }
        }
        const bool inner = on && id < leaf_first;
        if (on && !inner) {  // a leaf-level record: its hit leaves, one after the other (the lanes that are at one)
            const uint32_t lbase = (id - leaf_first) * 8u;
            while (hit != 0u) {
                const uint32_t c = (uint32_t)__builtin_ctz(hit);
                hit &= hit - 1u;
                offer(lbase + c);
            }
        }
        if (inner && hit != 0u) {
            pend = (pend << 8) | (uint64_t)(hit & ~(1u << nearest));
            id = id * 8u + nearest;
            off = off * 8 + 1;
        } else if (on) {
            if (pend == 0ull) {
                on = false;
            } else {
                const uint32_t z = (uint32_t)__builtin_ctzll(pend);
                const uint32_t j3 = (z >> 3) * 3u;
                pend >>= (z & 56u);
                id = ((id >> j3) & ~7u) | (z & 7u);
                off >>= j3;
                const uint32_t lo = (uint32_t)pend;
                pend = (pend & 0xffffffff00000000ull) | (uint64_t)(lo & (lo - 1u));
            }
        }
    }
}


template <class LeafRecFn>
__device__ __forceinline__ uint32_t traverse_records(const float* records_g, uint32_t leaf_first,
                                                     const Cube& cube, LeafRecFn&& leaf_rec) {
    return traverse_from(records_g, leaf_first, 1u, cube, leaf_rec);
}

// leaf(L): processes leaf L for every lane and may shrink the lane's cube (the leaves of a
// record were all tested before the first of them shrank the bounds: at worst a wasted leaf).
template <class LeafFn>
__device__ __forceinline__ uint32_t traverse_wide(const float* records_g, uint32_t leaf_first,
                                                  const Cube& cube, LeafFn&& leaf) {
    return traverse_records(records_g, leaf_first, cube, [&](uint32_t lbase, uint32_t, uint32_t hit) {
        while (hit) {
            const uint32_t c = (uint32_t)__builtin_ctz(hit);
            hit &= hit - 1u;
            leaf(lbase + c);
        }
    });
}

}  // namespace mi
