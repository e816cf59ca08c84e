// leaf_halo.h -- every leaf's HALO: the nearest points of other leaves around its region, in rings.
//
// Why.  A seeded query (nn_search.h) whose search cube lies inside its previous match's leaf
// region is finished without touching the tree.  With measurement noise the cube -- half-width =
// the distance to the current match -- pokes out of that region for every query that sits closer
// to a face than to its match (a third to a half of them at sigma = 0.15 spacings).  What such a
// query can still find are the few points just outside the region.  Round 2 kept, per leaf, a
// distance-ordered list of the 32 nearest neighbour LEAVES with direction masks: a lane scanned it
// in lock-step with the rest of its wave (~12 wave-instructions per entry, 12-28 entries for the
// slowest lane, most of them excluded by their mask) and then evaluated the 3-5 leaves that
// passed: 0.12 / 0.18 / 0.25 ms per 6M-query pass at sigma = 0.05 / 0.15 / 0.3 against 0.055 on
// exact data.  Here the POINTS are kept instead, in the leaf lines' own format (128 bytes:
// x[8] y[8] z[8] slot[8]): halo line k of leaf L with region R holds the points of other leaves
// that are 8k+1-th .. 8k+8-th nearest to R (L-infinity distance to the box), ascending in slot
// inside the line, 8 lines = 64 points per leaf; REACH k: every point of another leaf nearer to R
// than that is in the lines 0 .. k (the distance of the first point behind them, or the bound the
// candidates were collected to, about one spacing).  The eight reaches travel with the region, in
// the two spare words of its record (halo_pack_reaches below): a query whose cube pokes out of R
// by `over` -- every point of another leaf inside the cube is then within `over` of R -- knows
// from the record it has anyway how many lines to read, reads them (no scan, no filter; the lanes
// of a leaf read the same lines) and is finished.  Mean reaches on uniform data: 0.2 / 0.37 / 0.51
// ... spacings for 1 / 2 / 3 lines, the bound for all eight.
// Tried on the way (all exact, all measured on the 6M-query noisy pass): lines per FACE of the
// region (the 7 + 7 nearest beyond each face: one or two lines per poked face, but the 8th / 15th
// point came at 0.64 / 0.89 spacings in the median and 0.26 / 0.52 for 1 % of the lines, and ONE
// lane of a packet beyond its lines' reach sends the whole wave up the tree: 1.2 / 12.8 records
// per packet at sigma = 0.15 / 0.3); lines per face and per EDGE of the region with extension
// lines (18 + 8 lines: reach = the bound almost everywhere, 0.101 / 0.144 / 0.234 ms with three
// ring lines in front) -- but 3.7 KB per leaf, 3.8 GB for a 10M-point target, whose build took
// 5.6 ms, and nearly every line a lane read was its own 128 bytes from HBM.
//
// Built in two launches.  leaf_halo_collect: one wave per 64 consecutive leaves with the packet
// walk of traverse.h -- lane = leaf, search cube = R grown by the lane's bound (lreg[7] = a quarter
// of the leaf-level node's size, about one point spacing on volumetric data), the walk starts at
// the node that holds the 64 leaves and climbs until every lane's cube is inside a completed
// subtree's region; every leaf box a lane's cube overlaps is appended to the lane's row of a
// scratch tile (no LDS, so the walk -- a chain of dependent record fetches -- runs at full
// occupancy).  A row of 64 candidates that fills up stops accepting and its bound drops to the
// nearest box it turned away (rare).  leaf_halo_build: a wave per leaf gathers the candidates'
// points (8 leaves per round, one point per lane), keeps those nearer than the bound, sorts them by
// distance (register bitonic network, 128 keys): lane l then holds the l-th nearest point, i.e.
// entry l % 8 of line l / 8.
#pragma once
#include "device_utils.h"
#include "halo_format.h"
#include "traverse.h"

namespace mi {

constexpr uint32_t kLinkIdMask = 0x3ffffffu;  // leaf ids fit 26 bits
constexpr int kLinkCand = 64;                 // candidate leaves per leaf (scratch)
static_assert(kHaloLines == 8 && kHaloLineFloats == kLeafFloats, "nn_search.h evaluates halo lines as leaf lines");

// Scratch: tiles of 64 consecutive leaves, candidate t of leaf L at link_temp_index (slot-major inside the
// tile: the build's loads are coalesced).
__host__ __device__ __forceinline__ size_t link_temp_index(uint32_t L, int t) {  // in uint2
    return ((size_t)(L >> 6) * kLinkCand + (size_t)t) * 64u + (L & 63u);
}

// lreg[L][3] <- bound, lreg[L][7] <- number of candidates (as an integer's bits), cand[L][0..count) unsorted;
// candidate = {leaf id | direction mask << 26, distance bits}
static __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void leaf_halo_collect(
        const float* __restrict__ records_g, uint32_t leaf_first, int nleaf, uint32_t nblocks,
        float* __restrict__ lreg, uint2* __restrict__ cand) {
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const int lane = lane_id();
    const uint32_t L = logical * 64u + (uint32_t)lane;
    const bool valid = L < (uint32_t)nleaf;
    float4 g0 = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), g1 = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    if (valid) {
        const float4* rg = reinterpret_cast<const float4*>(lreg + (size_t)L * kLeafRegStride);
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
    lreg[(size_t)L * kLeafRegStride + 3] = usable ? bound : 0.0f;
    lreg[(size_t)L * kLeafRegStride + 7] = __int_as_float(count);
}

// inclusive prefix sum over the wave on the DPP network (row_shr 1, 2, 4, 8 scan each row of 16 lanes,
// row_bcast:15 / row_bcast:31 carry the row totals across)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// lane ^ J's value of v for a compile-time J < 64: DPP inside a quad, ds_swizzle (no address register, no memory)
// inside 32 lanes, ds_bpermute across the halves
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    else if constexpr (J < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1f);   // bit mode: and 0x1f, xor J
    else return (uint32_t)__shfl_xor((int)v, J, 64);
}

template <int K, int J>
__device__ __forceinline__ void halo_sort_stage(uint32_t (&key)[2], int lane) {
    if constexpr (J == 64) {  // K = 128: the partner is the other register, ascending throughout
        const uint32_t mn = min(key[0], key[1]), mx = max(key[0], key[1]);
        key[0] = mn;
        key[1] = mx;
    } else {
        const bool lower = (lane & J) == 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // direction of the merge this element takes part in: bit K of e = r * 64 + lane (K = 64: r, K = 128: 0)
            const bool asc = (K == 128) ? true : ((K == 64) ? (r == 0) : ((lane & K) == 0));
            const uint32_t o = lane_xor<J>(key[r]);
            key[r] = (lower == asc) ? min(key[r], o) : max(key[r], o);
        }
    }
    if constexpr (J > 1) halo_sort_stage<K, J / 2>(key, lane);
}
template <int K>
__device__ __forceinline__ void halo_sort_merges(uint32_t (&key)[2], int lane) {
    if constexpr (K > 2) halo_sort_merges<K / 2>(key, lane);
    halo_sort_stage<K, K / 2>(key, lane);
}
// Wave-wide bitonic sort of 128 keys, ascending: element e = r * 64 + lane.
__device__ __forceinline__ void halo_sort128(uint32_t (&key)[2], int lane) { halo_sort_merges<128>(key, lane); }

// ... of 64 keys, one per lane
template <int K, int J>
__device__ __forceinline__ void halo_sort64_stage(uint32_t& key, int lane) {
    const bool asc = (K == 64) ? true : ((lane & K) == 0);
    const uint32_t o = lane_xor<J>(key);
    key = (((lane & J) == 0) == asc) ? min(key, o) : max(key, o);
    if constexpr (J > 1) halo_sort64_stage<K, J / 2>(key, lane);
}
template <int K>
__device__ __forceinline__ void halo_sort64_merges(uint32_t& key, int lane) {
    if constexpr (K > 2) halo_sort64_merges<K / 2>(key, lane);
    halo_sort64_stage<K, K / 2>(key, lane);
}
__device__ __forceinline__ void halo_sort64(uint32_t& key, int lane) { halo_sort64_merges<64>(key, lane); }

constexpr int kHaloTile = 16;                    // leaves per wave = per workgroup
constexpr uint32_t kHaloDistMask = 0xfffffe00u;  // key: 23 bits of the distance | 9 of the point's index
constexpr float kHaloShrink = 0.999999f;         // reaches are reported a little short, the overhang a little long

// One wave per 16 consecutive leaves (a quarter of a scratch tile), a leaf at a time; no workgroup barrier and
// 4.7 KB of LDS per wave, so the kernel's occupancy is what its registers allow (with four waves sharing a
// tile's 17 KB of candidate ids it held 13.6 waves per CU).  halo: [nleaf][8] lines of 32 floats; lreg[L][3], [7] <-
// the rings' reaches, packed (halo_pack_reaches; both 0: no halo).
static __global__ __launch_bounds__(64) void leaf_halo_build(float* __restrict__ lreg, int nleaf,
                                                      const uint2* __restrict__ cand,
                                                      const float* __restrict__ tblk,
                                                      float* __restrict__ halo) {
    __shared__ uint32_t s_ids[kHaloTile * 65];  // [leaf of the wave's 16][candidate], row stride 65: conflict-free both ways
    __shared__ uint32_t s_keys[128];            // the in-bound keys, compacted
    const int lane = (int)threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t)kHaloTile;  // first leaf
    const uint32_t tile = base >> 6, sub = base & 63u;        // scratch tile, first leaf inside it
    // the candidate ids of the 16 leaves: rows of the scratch tile -> LDS, transposed (lane = (row of 4, leaf))
    int cnt_mine = 0;
    {
        const uint32_t Lm = base + (uint32_t)(lane & 15);
        if (Lm < (uint32_t)nleaf) cnt_mine = __float_as_int(lreg[(size_t)Lm * kLeafRegStride + 7]);
        int cmax = cnt_mine;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
        cmax = __builtin_amdgcn_readfirstlane(cmax);
        for (int t0 = 0; t0 < cmax; t0 += 4) {
            const int t = t0 + (lane >> 4);
            if (t < cmax)
                s_ids[(lane & 15) * 65 + t] = cand[((size_t)tile * kLinkCand + (size_t)t) * 64u + sub + (uint32_t)(lane & 15)].x;
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t* keys = s_keys;
    for (int li = 0; li < kHaloTile; ++li) {
        const uint32_t L = base + (uint32_t)li;  // wave-uniform
        if (L >= (uint32_t)nleaf) break;
        const int count = __shfl(cnt_mine, li, 64);
        float* lines = halo + (size_t)L * (kHaloStored * kHaloLineFloats);
        const float4* rg = reinterpret_cast<const float4*>(lreg + (size_t)L * kLeafRegStride);
        const float4 g0 = rg[0], g1 = rg[1];
        float bound = g0.w;  // (collect's final bound; 0: not usable)
        if (count <= 0 || !(bound > 0.0f)) {
            // no halo: the search never reads the lines (reaches 0)
            if (lane == 0) {
                lreg[(size_t)L * kLeafRegStride + 3] = 0.0f;
                lreg[(size_t)L * kLeafRegStride + 7] = 0.0f;
            }
            continue;
        }
        // ---- gather: round r covers candidates 32r .. 32r+31, lane = (candidate, half of its line): three 16-byte
        // loads bring four points (12 one-float loads per leaf cost the texture path more than everything else here);
        // key = distance to the region (L-infinity distance to the box; 23 bits, rounded down) | index of the point
        // = candidate * 8 + slot
        const int rounds = (count + 31) >> 5;
        uint32_t key[8];
        {
            float4 px[2], py[2], pz[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                px[r] = py[r] = pz[r] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                if (r < rounds) {
                    const int c = r * 32 + (lane >> 1);
                    if (c < count) {
                        const uint32_t id = s_ids[li * 65 + c] & kLinkIdMask;
                        const float4* ln = reinterpret_cast<const float4*>(tblk + (size_t)id * kLeafFloats) + (lane & 1);
                        px[r] = ln[0];
                        py[r] = ln[2];
                        pz[r] = ln[4];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float xs[4] = {px[r].x, px[r].y, px[r].z, px[r].w}, ys[4] = {py[r].x, py[r].y, py[r].z, py[r].w};
                const float zs[4] = {pz[r].x, pz[r].y, pz[r].z, pz[r].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    key[r * 4 + j] = 0xffffffffu;
                    if (r < rounds) {
                        const float ux = xs[j] - g1.x, lx = g0.x - xs[j], uy = ys[j] - g1.y, ly = g0.y - ys[j];
                        const float uz = zs[j] - g1.z, lz = g0.z - zs[j];
                        const float dist = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(fmaxf(uz, lz), 0.0f));
                        // (padding slots are +inf: dist = +inf; NaN coordinates compare false)
                        if (dist < INFINITY)
                            key[r * 4 + j] = (__float_as_uint(dist) & kHaloDistMask) | (uint32_t)((r * 32 + (lane >> 1)) * 8 + (lane & 1) * 4 + j);
                    }
                }
            }
        }
        // ---- the points nearer than the bound, at most 128 of them (halve the bound until they are)
        const int nk = rounds * 4;  // keys in use
        // (a lane's keys go to consecutive places: its count of them, a wave-wide scan of the counts, then one LDS
        // store per key -- a ballot and two bit counts per key register were a tenth of this kernel's instructions)
        uint32_t n_in, mine, incl;
        for (;;) {
            const uint32_t bb = __float_as_uint(bound);
            mine = 0u;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < nk) {
                    if ((key[t] & kHaloDistMask) >= bb) key[t] = 0xffffffffu;  // (also the invalid ones: all bits set)
                    mine += (key[t] != 0xffffffffu) ? 1u : 0u;
                }
            incl = wave_inclusive_sum(mine);
            n_in = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (n_in <= 128u) break;
            bound *= 0.5f;  // (reaches 0 eventually: nothing is nearer than that)
        }
        {
            uint32_t pos = incl - mine;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < nk) {
                    if (key[t] != 0xffffffffu) keys[pos++] = key[t];
                }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t sk[2];
        sk[0] = ((uint32_t)lane < n_in) ? keys[lane] : 0xffffffffu;
        sk[1] = 0xffffffffu;
        if (n_in <= 64u) {  // (about half of the leaves on uniform data: a third of the larger network's exchanges)
            halo_sort64(sk[0], lane);
        } else {
            sk[1] = ((uint32_t)lane + 64u < n_in) ? keys[64 + lane] : 0xffffffffu;
            halo_sort128(sk, lane);
        }
        // ---- the rings: sorted elements 8k .. 8k+7 are line k, i.e. lane = (line, entry) holds its point already
        uint32_t slot = 0xffffffffu;
        if (lane < 8 * kHaloStored && sk[0] != 0xffffffffu) {
            const uint32_t idx = sk[0] & 511u;  // candidate * 8 + slot
            slot = (s_ids[li * 65 + (int)(idx >> 3)] & kLinkIdMask) * (uint32_t)kLeaf + (idx & 7u);
        }
        // a line's entries ascend in slot (equal distances must resolve to the lowest slot: nn_search.h takes the
        // first of equals inside a line)
        {
            uint32_t o;
            o = lane_xor<1>(slot); slot = (((lane & 1) == 0) == ((lane & 2) == 0)) ? min(slot, o) : max(slot, o);
            o = lane_xor<2>(slot); slot = (((lane & 2) == 0) == ((lane & 4) == 0)) ? min(slot, o) : max(slot, o);
            o = lane_xor<1>(slot); slot = (((lane & 1) == 0) == ((lane & 4) == 0)) ? min(slot, o) : max(slot, o);
            o = lane_xor<4>(slot); slot = ((lane & 4) == 0) ? min(slot, o) : max(slot, o);
            o = lane_xor<2>(slot); slot = ((lane & 2) == 0) ? min(slot, o) : max(slot, o);
            o = lane_xor<1>(slot); slot = ((lane & 1) == 0) ? min(slot, o) : max(slot, o);
        }
        if (lane < 8 * kHaloStored) {
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (slot != 0xffffffffu) {
                const float* ln = tblk + (size_t)(slot >> 3) * kLeafFloats + (slot & 7u);
                x = ln[0];
                y = ln[8];
                z = ln[16];
            }
            float* ln = lines + (lane >> 3) * kHaloLineFloats + (lane & 7);
            ln[0] = x;
            ln[8] = y;
            ln[16] = z;
            ln[24] = __int_as_float((int)slot);
        }
        // ---- reaches: ring k ends where the first point behind it begins (none: the bound)
        uint32_t nx = (uint32_t)__shfl((int)sk[0], (lane < 7) ? 8 * (lane + 1) : 0, 64);  // lanes 0..6: sorted element 8 (k + 1)
        const uint32_t nx7 = (uint32_t)__shfl((int)sk[1], 0, 64);                         // sorted element 64
        if (lane == 7) nx = nx7;
        const uint32_t bbits = __float_as_uint(bound * kHaloShrink) & 0xffff0000u;  // the bound, 16 bits, rounded down
        const float unit = __uint_as_float(bbits) * kHaloUnit;
        uint32_t pa = 0u, pb = 0u;  // this lane's bits of the two words (halo_reach_fraction)
        if (lane < kHaloLines && unit > 0.0f) {
            const float reach = fminf((nx != 0xffffffffu) ? __uint_as_float(nx & kHaloDistMask) * kHaloShrink : INFINITY,
                                      __uint_as_float(bbits));
            uint32_t q = (uint32_t)fminf(reach * __builtin_amdgcn_rcpf(unit) + 1.0f, 63.0f);
            while (q > 0u && !(unit * (float)q <= reach)) --q;  // (the search's own arithmetic)
            if (kHaloStored < kHaloLines) q = (uint32_t)__shfl((int)q, min(lane, kHaloStored - 1), 64);  // (lines not stored: nothing beyond the last stored one's reach)
            if (lane < 5) pa = q << (6 * lane);
            else if (lane < 7) pb = q << (6 * (lane - 5));
            else {
                pa = (q & 3u) << 30;
                pb = (q >> 2) << 12;
            }
        }
        // OR over the lanes 0..7 (one row of the DPP network)
        pa |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pa, 0x111, 0xf, 0xf, false);  // row_shr:1
        pb |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pb, 0x111, 0xf, 0xf, false);
        pa |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pa, 0x112, 0xf, 0xf, false);  // row_shr:2
        pb |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pb, 0x112, 0xf, 0xf, false);
        pa |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pa, 0x114, 0xf, 0xf, false);  // row_shr:4
        pb |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pb, 0x114, 0xf, 0xf, false);
        const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)pa, 7);
        const uint32_t wb = bbits | (uint32_t)__builtin_amdgcn_readlane((int)pb, 7);
        if (lane == 0) {
            lreg[(size_t)L * kLeafRegStride + 3] = __uint_as_float(wa);
            lreg[(size_t)L * kLeafRegStride + 7] = (unit > 0.0f) ? __uint_as_float(wb) : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();  // (keys are rewritten by the next leaf)
    }
}

}  // namespace mi
