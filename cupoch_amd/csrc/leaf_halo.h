// leaf_halo.h -- every leaf's HALO: the nearest points of other leaves around its region, grouped
// by the faces of the region they lie beyond.
//
// Why.  A seeded query (nn_search.h) whose search cube lies inside its previous match's leaf
// region is finished without touching the tree.  With measurement noise the cube -- half-width =
// the distance to the current match -- pokes out of that region for every query that sits closer
// to a face than to its match (a third to a half of them at sigma = 0.15 spacings).  What such a
// query can still find are the few points just beyond the faces it pokes through.  Round 2 kept,
// per leaf, a distance-ordered list of the 32 nearest neighbour LEAVES with direction masks: a
// lane scanned it in lock-step with the rest of its wave (~12 wave-instructions per entry, 12-28
// entries for the slowest lane, most of them excluded by their mask) and then evaluated the 3-5
// leaves that passed: 0.12 / 0.18 / 0.25 ms per 6M-query pass at sigma = 0.05 / 0.15 / 0.3 against
// 0.055 on exact data.  Here the POINTS are kept instead, in the leaf lines' own format (128 bytes:
// x[8] y[8] z[8] slot[8]), 18 primary lines per leaf with region R:
//   * face line f (0: +x, 1: -x, 2: +y, 3: -y, 4: +z, 5: -z -- the search's `faces` mask): the 7
//     points of other leaves nearest to R (L-infinity distance to the box) among those that lie on
//     or beyond face f and NO other face, i.e. inside R's extent along the other two axes;
//   * edge line (f, g), faces of different axes (halo_edge_line): the 7 nearest among those on or
//     beyond both f and g -- including the points beyond a third face (they are in three edge lines).
// Points ascend in slot inside a line; the eighth x holds the line's REACH: every member point
// nearer to R than that is in the line (y[7] = z[7] = +inf: as a point it is infinitely far away, so
// a halo line is evaluated by the same code as a leaf line).  A primary line with more than 7 members
// within the bound gets one of the leaf's 8 EXTENSION lines (the next 7 members and the reach of the
// two together; the primary's slot[7] names it, -1: none): on uniform data 2.6 of a leaf's 18 lines
// have an eighth member within the bound, a fifteenth practically never.
// A point of another leaf lies on or beyond at least one face of R (the region's defining
// property), and if it lies inside a query's cube the cube pokes through every face the point is
// beyond.  So a query whose cube pokes through the faces F reads the face lines of F and the edge
// lines of the pairs in F -- 1 line for one face, 3 for two, 6 for three; no scan, no filter -- and
// has then seen every point its cube can hold that is nearer to R than those lines' reaches; its
// distance to R is at most the cube's overhang.  Splitting by the exact set of faces is what makes
// the reach long: a face line's members fill a slab of R's own cross-section (~4 points per spacing
// of depth on uniform data), an edge line's a quarter-pipe, so a line of 7 normally ends beyond the
// bound the candidates were collected to, about one spacing -- every lane up to that overhang is
// served.  (First attempt: per face the 7 + 7 nearest beyond that face whatever else they are
// beyond -- the slab then includes the rim around R's cross-section, the 8th / 15th point came at
// 0.64 / 0.89 spacings in the median and 0.26 / 0.52 for 1 % of the lines, and ONE lane of a packet
// beyond its lines' reach sends the whole wave up the tree: 1.2 / 12.8 records per packet at
// sigma = 0.15 / 0.3.)
//
// Built in two launches.  leaf_halo_collect: one wave per 64 consecutive leaves with the packet
// walk of traverse.h -- lane = leaf, search cube = R grown by the lane's bound (lreg[7] = a quarter
// of the leaf-level node's extent, about one point spacing on volumetric data), the walk starts
// at the node that holds the 64 leaves and climbs until every lane's cube is inside a completed
// subtree's region; every leaf box a lane's cube overlaps is appended to the lane's row of a
// scratch tile (no LDS, so the walk -- a chain of dependent record fetches -- runs at full
// occupancy).  A row of 64 candidates that fills up stops accepting and its bound drops to the
// nearest box it turned away (rare).  leaf_halo_build: a wave per leaf gathers the candidates'
// points (8 leaves per round, one point per lane), keeps those nearer than the bound, sorts them by
// distance (register bitonic network, 128 keys) and deals the first 7 members of every line to it.
#pragma once
#include "device_utils.h"
#include "nn_search.h"
#include "traverse.h"

namespace mi {

constexpr uint32_t kLinkIdMask = 0x3ffffffu;  // leaf ids fit 26 bits
constexpr int kLinkCand = 64;                 // candidate leaves per leaf (scratch)
static_assert(kHaloFaces == 6 && kHaloPrimary == 18 && kHaloExt == 8 && kHaloNear == 3 && kHaloLineFloats == kLeafFloats, "nn_search.h evaluates halo lines as leaf lines");

// Scratch: tiles of 64 consecutive leaves, candidate t of leaf L at link_temp_index (slot-major inside the
// tile: the build's loads are coalesced).
__host__ __device__ __forceinline__ size_t link_temp_index(uint32_t L, int t) {  // in uint2
    return ((size_t)(L >> 6) * kLinkCand + (size_t)t) * 64u + (L & 63u);
}

// lreg[L][3] <- bound, lreg[L][7] <- number of candidates (as an integer's bits), cand[L][0..count) unsorted;
// candidate = {leaf id | direction mask << 26, distance bits}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void leaf_halo_collect(
        const float* __restrict__ records_g, uint32_t leaf_first, int nleaf, uint32_t nblocks,
        float* __restrict__ lreg, uint2* __restrict__ cand) {
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const int lane = lane_id();
    const uint32_t L = logical * 64u + (uint32_t)lane;
    const bool valid = L < (uint32_t)nleaf;
    float4 g0 = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), g1 = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    if (valid) {
        const float4* rg = reinterpret_cast<const float4*>(lreg + (size_t)L * kLeafRegFloats);
        g0 = rg[0];
        g1 = rg[1];
    }
    const float delta0 = g1.w;
    const bool usable = valid && g0.x <= g1.x && g0.y <= g1.y && g0.z <= g1.z && delta0 > 0.0f && delta0 < INFINITY;
    float bound = usable ? delta0 : -1.0f;  // strict acceptance limit
    int count = 0;
    if (__ballot(usable) != 0ull) {  // (a packet of padding leaves -- the tail of every group -- has nothing to do)
        Cube cube;
        auto grow = [&](float reach) {  // R grown by `reach` on every side (infinite faces stay infinite)
            const float r = reach * 1.000001f;
            cube.lox = widen_down(g0.x - r);
            cube.loy = widen_down(g0.y - r);
            cube.loz = widen_down(g0.z - r);
            cube.hix = widen_up(g1.x + r);
            cube.hiy = widen_up(g1.y + r);
            cube.hiz = widen_up(g1.z + r);
        };
        if (usable) {
            grow(delta0);
        } else {
            cube.lox = cube.loy = cube.loz = INFINITY;
            cube.hix = cube.hiy = cube.hiz = -INFINITY;
        }
        // the node that holds this packet's 64 leaves: 8 leaf-level nodes = one node of the level above
        const uint32_t start = (leaf_first >= 8u) ? ((leaf_first >> 3) + logical) : 1u;
        traverse_from(records_g, leaf_first, start, cube,
                      [&](uint32_t lbase, uint32_t vm, uint32_t hit, const float(&w)[48]) {
#pragma unroll
                          for (int c = 0; c < 8; ++c) {
                              if (!((hit >> c) & 1u)) continue;  // wave-uniform
                              const float* b = w + (c >> 1) * kPairStride + (c & 1);
                              const uint32_t leaf = lbase + (uint32_t)c;
                              // gaps between the box [b0,b2,b4 .. b6,b8,b10] and R along each axis and side
                              const float ux = b[0] - g1.x, uy = b[2] - g1.y, uz = b[4] - g1.z;   // beyond the upper faces
                              const float lx = g0.x - b[6], ly = g0.y - b[8], lz = g0.z - b[10];  // beyond the lower faces
                              const float dist = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(fmaxf(uz, lz), 0.0f));
                              if (((vm >> c) & 1u) != 0u && leaf != L && dist < bound) {
                                  if (count < kLinkCand) {
                                      cand[link_temp_index(L, count)] = make_uint2(leaf, __float_as_uint(dist));
                                      ++count;
                                  } else {  // full: turned away, and nothing this far is promised any more
                                      bound = dist;
                                      grow(bound);
                                  }
                              }
                          }
                      });
    }
    if (!valid) return;
    lreg[(size_t)L * kLeafRegFloats + 3] = usable ? bound : 0.0f;
    lreg[(size_t)L * kLeafRegFloats + 7] = __int_as_float(count);
}

// Wave-wide bitonic sort of 128 keys, ascending: element e = r * 64 + lane.
__device__ __forceinline__ void halo_sort128(uint32_t (&key)[2], int lane) {
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) {  // k = 128: the partner is the other register, ascending throughout
                const uint32_t mn = min(key[0], key[1]), mx = max(key[0], key[1]);
                key[0] = mn;
                key[1] = mx;
            } else {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    // direction of the merge this element takes part in: bit k of e (k = 64: r, k = 128: 0)
                    const bool asc = (k == 128) ? true : ((k == 64) ? (r == 0) : ((lane & k) == 0));
                    const uint32_t o = (uint32_t)__shfl_xor((int)key[r], j, 64);
                    key[r] = (lower == asc) ? min(key[r], o) : max(key[r], o);
                }
            }
        }
    }
}

constexpr int kHaloWaves = 4;                  // waves per workgroup = per tile of 64 leaves
constexpr uint32_t kHaloDistMask = 0xffff8000u;  // key: 17 bits of the distance | 6 of the face mask | 9 of the point's index
constexpr float kHaloShrink = 0.999999f;       // the reach is reported a little short, the overhang a little long

// One workgroup per tile of 64 leaves, a wave per leaf (16 leaves each).  halo: [nleaf][18 + 8 + 3] lines of 32 floats;
// lreg[L][7] <- the smallest reach of the leaf's face / edge lines (0: no halo), lreg[L][3] <- the near lines' reaches
// as fractions of it (what the search tests the overhang against).
__global__ __launch_bounds__(64 * kHaloWaves) void leaf_halo_build(float* __restrict__ lreg, int nleaf,
                                                                   const uint2* __restrict__ cand,
                                                                   const float* __restrict__ tblk,
                                                                   float* __restrict__ halo) {
    __shared__ uint32_t s_ids[64 * 65];           // [leaf of the tile][candidate], row stride 65: conflict-free both ways
    __shared__ uint32_t s_keys[kHaloWaves][128];  // a wave's in-bound keys, compacted
    __shared__ uint32_t s_sel[kHaloWaves][kHaloPrimary * 16 + 32];  // [primary line][rank]: its first 15 members in distance order | the 22 nearest of all
    __shared__ int s_ext[kHaloWaves][kHaloPrimary + kHaloExt];   // extension line of a primary line (-1: none) | primary line of an extension
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t tile = blockIdx.x;
    // the tile's candidate ids: coalesced rows of the scratch tile -> LDS, transposed
    int cnt_mine = 0;
    {
        const uint32_t Lm = tile * 64u + (uint32_t)lane;
        if (Lm < (uint32_t)nleaf) cnt_mine = __float_as_int(lreg[(size_t)Lm * kLeafRegFloats + 7]);
        int cmax = cnt_mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
        for (int t = wid; t < cmax; t += kHaloWaves)
            s_ids[lane * 65 + t] = cand[((size_t)tile * kLinkCand + (size_t)t) * 64u + (size_t)lane].x;
    }
    __syncthreads();
    uint32_t* keys = s_keys[wid];
    uint32_t* sel = s_sel[wid];
    int* ext_of = s_ext[wid];
    int* ext_src = s_ext[wid] + kHaloPrimary;
    for (int li = wid; li < 64; li += kHaloWaves) {
        const uint32_t L = tile * 64u + (uint32_t)li;  // wave-uniform
        if (L >= (uint32_t)nleaf) break;
        const int count = __shfl(cnt_mine, li, 64);
        float* lines = halo + (size_t)L * (kHaloLines * kHaloLineFloats);
        const float4* rg = reinterpret_cast<const float4*>(lreg + (size_t)L * kLeafRegFloats);
        const float4 g0 = rg[0], g1 = rg[1];
        float bound = g0.w;  // (collect's final bound; 0: not usable)
        if (count <= 0 || !(bound > 0.0f)) {
            // no halo: the search never reads the lines (reaches 0)
            if (lane == 0) {
                lreg[(size_t)L * kLeafRegFloats + 3] = 0.0f;
                lreg[(size_t)L * kLeafRegFloats + 7] = 0.0f;
            }
            continue;
        }
        // ---- gather: round r covers candidates 8r .. 8r+7, lane = (candidate, slot)
        const int rounds = (count + 7) >> 3;
        uint32_t key[8];
        {
            float px[8], py[8], pz[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                px[r] = py[r] = pz[r] = INFINITY;
                if (r < rounds) {
                    const int c = r * 8 + (lane >> 3);
                    if (c < count) {
                        const uint32_t id = s_ids[li * 65 + c] & kLinkIdMask;
                        const float* ln = tblk + (size_t)id * kLeafFloats + (lane & 7);
                        px[r] = ln[0];
                        py[r] = ln[8];
                        pz[r] = ln[16];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                key[r] = 0xffffffffu;
                if (r < rounds) {
                    // gaps to R's faces: >= 0 means on or beyond that face
                    const float ux = px[r] - g1.x, lx = g0.x - px[r], uy = py[r] - g1.y, ly = g0.y - py[r];
                    const float uz = pz[r] - g1.z, lz = g0.z - pz[r];
                    const float dist = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(fmaxf(uz, lz), 0.0f));
                    const uint32_t mask = (ux >= 0.0f ? 1u : 0u) | (lx >= 0.0f ? 2u : 0u) | (uy >= 0.0f ? 4u : 0u) |
                                          (ly >= 0.0f ? 8u : 0u) | (uz >= 0.0f ? 16u : 0u) | (lz >= 0.0f ? 32u : 0u);
                    // (padding slots are +inf: dist = +inf; NaN coordinates: dist compares false below)
                    if (dist < INFINITY && mask != 0u)
                        key[r] = (__float_as_uint(dist) & kHaloDistMask) | (mask << 9) | (uint32_t)(r * 64 + lane);
                }
            }
        }
        // ---- the points nearer than the bound, at most 128 of them (halve the bound until they are)
        uint32_t n_in;
        for (;;) {
            const uint32_t bb = __float_as_uint(bound);
            n_in = 0u;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < rounds) n_in += (uint32_t)__popcll(__ballot(key[r] != 0xffffffffu && (key[r] & kHaloDistMask) < bb));
            if (n_in <= 128u) break;
            bound *= 0.5f;  // (reaches 0 eventually: nothing is nearer than that)
        }
        {
            const uint32_t bb = __float_as_uint(bound);
            uint32_t base = 0u;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < rounds) {
                    const bool in = key[r] != 0xffffffffu && (key[r] & kHaloDistMask) < bb;
                    const uint64_t m = __ballot(in);
                    if (in) keys[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = key[r];
                    base += (uint32_t)__popcll(m);
                }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t sk[2];
        sk[0] = ((uint32_t)lane < n_in) ? keys[lane] : 0xffffffffu;
        sk[1] = ((uint32_t)lane + 64u < n_in) ? keys[64 + lane] : 0xffffffffu;
        halo_sort128(sk, lane);
        // ---- per primary line, in distance order: the first 7 members go to it, the next 7 to its extension
        // line if it gets one, the first member left out gives the reach (none left out: the bound).  Members
        // of face line f: the points beyond that face only; of edge line (f, g): beyond both (a point beyond
        // three faces is a member of three edge lines).
        for (int t = lane; t < kHaloPrimary * 16; t += 64) sel[t] = 0xffffffffu;
        if (lane < 32) sel[kHaloPrimary * 16 + lane] = sk[0];  // the nearest of all, in distance order (sorted element e = r * 64 + lane)
        __builtin_amdgcn_wave_barrier();
        auto deal = [&](int line, uint32_t want, uint32_t care) {
            uint32_t base = 0u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool has = sk[r] != 0xffffffffu && (((sk[r] >> 9) & care) == want);
                const uint64_t m = __ballot(has);
                if (m != 0ull && base < 15u) {
                    const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (has && rank < 15u) sel[line * 16 + (int)rank] = sk[r];
                }
                base += (uint32_t)__popcll(m);
            }
        };
#pragma unroll
        for (int f = 0; f < kHaloFaces; ++f) deal(f, 1u << f, 63u);
#pragma unroll
        for (int f = 0; f < kHaloFaces; ++f)
#pragma unroll
            for (int g = (f | 1) + 1; g < kHaloFaces; ++g) deal(halo_edge_line(f, g), (1u << f) | (1u << g), (1u << f) | (1u << g));
        __builtin_amdgcn_wave_barrier();
        // extension lines go to the primary lines with an eighth member, in line order, while there are any
        {
            const bool need = lane < kHaloPrimary && sel[lane * 16 + 7] != 0xffffffffu;
            const uint64_t m = __ballot(need);
            const uint32_t nth = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const bool gets = need && nth < (uint32_t)kHaloExt;
            if (lane < kHaloPrimary) ext_of[lane] = gets ? (int)(kHaloPrimary + nth) : -1;
            if (lane < kHaloExt) ext_src[lane] = -1;
            __builtin_amdgcn_wave_barrier();
            if (gets) ext_src[nth] = lane;
        }
        __builtin_amdgcn_wave_barrier();
        float reach_min = INFINITY, reach_mine = INFINITY;
        // one pass writes 6 lines: lanes 0..47 = (line, rank): the chosen points by ascending slot, rank 7 = the
        // reach and, in the fourth row, the line's extension (-1: none)
        auto write_lines = [&](int line, int sel_at, bool primary, int src, bool active) {
            const int j = lane & 7;
            const bool is_reach = j == 7;
            uint32_t e = 0xffffffffu;
            if (active) e = sel[sel_at + j];  // (the eighth: the first member this line leaves out)
            uint32_t slot = 0xffffffffu;
            if (active && !is_reach && e != 0xffffffffu) {
                const uint32_t idx = e & 511u;  // r * 64 + lane of the gather: candidate 8r + lane/8, slot lane%8
                const uint32_t c = (idx >> 6) * 8u + ((idx >> 3) & 7u);
                slot = (s_ids[li * 65 + (int)c] & kLinkIdMask) * (uint32_t)kLeaf + (idx & 7u);
            }
            // sort the 8 lanes of a line by slot (equal distances must resolve to the lowest slot: nn_search.h takes
            // the first of equals inside a line); rank 7 holds 0xffffffff and stays last
#pragma unroll
            for (int k = 2; k <= 8; k <<= 1)
#pragma unroll
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)slot, jj, 64);
                    const bool asc = (k == 8) ? true : ((lane & k) == 0);
                    slot = (((lane & jj) == 0) == asc) ? min(slot, o) : max(slot, o);
                }
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (slot != 0xffffffffu) {
                const float* ln = tblk + (size_t)(slot >> 3) * kLeafFloats + (slot & 7u);
                x = ln[0];
                y = ln[8];
                z = ln[16];
            }
            int fourth = (int)slot;
            if (is_reach) {
                x = ((e != 0xffffffffu) ? __uint_as_float(e & kHaloDistMask) : bound) * kHaloShrink;
                const int ext = (active && primary) ? ext_of[src] : -1;
                fourth = ext;
                // the line's reach together with its extension is the extension's (written by its own pass)
                if (active && ext < 0) reach_mine = x;
            }
            if (active) {
                float* ln = lines + line * kHaloLineFloats + j;
                ln[0] = x;
                ln[8] = y;
                ln[16] = z;
                ln[24] = __int_as_float(fourth);
            }
        };
#pragma unroll
        for (int p = 0; p < kHaloPrimary / 6; ++p) {
            reach_mine = INFINITY;
            write_lines(p * 6 + (lane >> 3), (p * 6 + (lane >> 3)) * 16, true, p * 6 + (lane >> 3), lane < 48);
            reach_min = fminf(reach_min, reach_mine);
        }
        {
            // extension lines (8 of them: all 64 lanes); unused ones are never named by a primary line
            const int src = ext_src[lane >> 3];
            reach_mine = INFINITY;
            write_lines(kHaloPrimary + (lane >> 3), (src < 0 ? 0 : src) * 16 + 7, false, 0, src >= 0);
            reach_min = fminf(reach_min, reach_mine);
        }
        // near lines: the 7 / 14 / 21 nearest of all; each one's reach covers the lines before it as well
        reach_mine = INFINITY;
        write_lines(kHaloNearFirst + (lane >> 3), kHaloPrimary * 16 + 7 * (lane >> 3), false, 0, lane < 8 * kHaloNear);
        // What the search compares a cube's overhang with before it reads any line: float 7 = the smallest reach
        // of the face / edge lines, float 3 = the three near reaches as 10-bit fractions of it, rounded down (with
        // the search's own arithmetic).
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) reach_min = fminf(reach_min, __shfl_xor(reach_min, o, 64));
        uint32_t q = 0u;
        if (lane < 8 * kHaloNear && (lane & 7) == 7 && reach_min > 0.0f && reach_min < INFINITY) {
            const float unit = reach_min * 0.0009765625f;
            q = (uint32_t)fminf(reach_mine / unit, 1023.0f);
            while (q > 0u && !(unit * (float)q <= reach_mine)) --q;
        }
        const uint32_t q0 = (uint32_t)__shfl((int)q, 7, 64), q1 = (uint32_t)__shfl((int)q, 15, 64), q2 = (uint32_t)__shfl((int)q, 23, 64);
        if (lane == 0) {
            lreg[(size_t)L * kLeafRegFloats + 3] = __uint_as_float(q0 | (q1 << 10) | (q2 << 20));
            lreg[(size_t)L * kLeafRegFloats + 7] = (reach_min < INFINITY) ? reach_min : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();  // (sel / keys are rewritten by the next leaf)
    }
}

}  // namespace mi
