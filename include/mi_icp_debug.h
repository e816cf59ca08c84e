/*
 * mi_icp_debug.h -- test-only entry points of libmi_icp.so.  Not part of the
 * drop-in boundary (nothing in the reference binds to these): they expose the
 * hand-written device primitives (radix sort, exclusive scan) that replace
 * thrust::sort_by_key / exclusive_scan inside the engine, so that the parity
 * tests can pin them on their own.  Buffers are host memory.
 * (and the built tree, for invariant tests)
 */
#ifndef MI_ICP_DEBUG_H_
#define MI_ICP_DEBUG_H_
#include "mi_icp.h"
#ifdef __cplusplus
extern "C" {
#endif
/* stable LSD radix sort of (key, value) pairs by the low key_bits bits, in place */
MI_ICP_API int mi_icp_debug_sort_pairs(mi_icp_ctx* ctx, uint64_t* keys, uint32_t* vals,
                                       int64_t n, int key_bits);
/* out[i] = sum of in[0..i); *total = sum of all (out may alias in) */
MI_ICP_API int mi_icp_debug_exclusive_scan(mi_icp_ctx* ctx, const uint32_t* in, uint32_t* out,
                                           int64_t n, uint64_t* total);
/* Morton order of a cloud as the engine computes it: order[sorted] = original */
MI_ICP_API int mi_icp_debug_morton_order(mi_icp_ctx* ctx, const float* xyz, int64_t n,
                                         uint32_t* order);
/* traversal census of one nearest-neighbour pass: out4 = {node visits summed over
 * packets, leaf visits, packets, visits (nodes + leaves) of the slowest packet}.
 * use_seed != 0 seeds from the previous pass. */
MI_ICP_API int mi_icp_debug_nn_stats(mi_icp_ctx* ctx, const float* T, float radius, int use_seed,
                                     uint64_t* out4);
/* the same pass with four more counters: out8[4] = halo lines evaluated (leaf_halo.h), [5] = packets
 * that evaluated any, [6] = packets that walked the tree, [7] = lanes unfinished when their packet's walk
 * started. */
MI_ICP_API int mi_icp_debug_nn_stats8(mi_icp_ctx* ctx, const float* T, float radius, int use_seed,
                                      uint64_t* out8);
/* Forget the previous search result: the next nearest-neighbour pass starts top-down instead of
 * from its predecessor's matches (tests compare the two; the results must be identical).  The
 * context has no correspondence set until that pass has run. */
MI_ICP_API int mi_icp_debug_drop_seeds(mi_icp_ctx* ctx);
/* How the last nearest-neighbour pass started: 0 = from the root (no previous matches), 1 = from the
 * previous pass's matches, 2 = from seeds it made itself (a greedy descent per query; what a registration
 * loop's first pass does when the target's neighbour lists were built ahead, mi_icp_set_target on a
 * context that has registered before), -1 = no pass yet. */
MI_ICP_API int mi_icp_debug_last_search_kind(const mi_icp_ctx* ctx);
/* The loop step's two forms of utility::SolveJacobianSystemAndObtainExtrinsicMatrix side by side, on the
 * device: n systems of 32 doubles each (host memory; the reduction's layout: 21 upper-triangle sums of
 * JtJ, 6 of Jtr, ...) are solved by one thread with the serial routines and by a wave with a matrix row
 * per lane.  out_serial / out_wave: n * 16 floats (column-major 4x4); ok_serial / ok_wave: n flags
 * (0 = the determinant check failed, result identity).  The two must agree bit for bit. */
MI_ICP_API int mi_icp_debug_solve_both(int device, const double* systems, int n, float det_thresh,
                                       float* out_serial, float* out_wave, int32_t* ok_serial, int32_t* ok_wave);
/* The target's tree as built by mi_icp_set_target, for invariant tests.  info5 = {slots
 * (padded sorted positions), leaves, leaf_first (id of the first leaf-level node), records,
 * points}.  records_out (records * 64 floats: 8 child boxes as 4 sibling pairs of 12,
 * floats 48..53 the node's region, 54 its validity flag) and leaf_lines_out (leaves * 32
 * floats: x[8] y[8] z[8] original index[8], padding = +inf / -1) may be NULL to query the
 * sizes only. */
MI_ICP_API int mi_icp_debug_get_tree(mi_icp_ctx* ctx, int64_t* info5, float* records_out,
                                     float* leaf_lines_out);
/* Per leaf (mi_icp_debug_get_tree's info5[1] leaves) 8 floats: region lo.xyz, the reaches of the leaf's
 * three near halo lines as 10-bit fractions (bits 0-9, 10-19, 20-29, in 1/1024) of float 7, region hi.xyz,
 * the smallest reach of the leaf's face / edge halo lines (0: no halo).  The region is free of points of any
 * other leaf (an invalid one is +inf / -inf: nothing is inside). */
MI_ICP_API int mi_icp_debug_get_leaf_regions(mi_icp_ctx* ctx, float* regions_out);
/* Per leaf 29 halo lines of 32 floats (builds them if no seeded search has yet): x[8] y[8] z[8] slot[8].
 * Line f < 6, face f of the leaf's region (+x, -x, +y, -y, +z, -z): the up to 7 points of OTHER leaves
 * nearest to the region (L-infinity distance to the box) among those on or beyond face f and no other
 * face.  Line 6 + ((a + b - 1) * 4 + 2 * side_a + side_b), the edge between the faces 2a + side_a and
 * 2b + side_b of the axes a < b: the up to 7 nearest among those on or beyond both faces.  Points ascend in
 * slot (sorted target position, as an integer's bits; unused entries +inf / -1); x[7] is the line's reach:
 * every member nearer to the region than that is in the line (y[7] = z[7] = +inf).  slot[7] of these 18
 * primary lines: -1, or the one of the lines 18..25 that holds the line's next 7 members and the reach of the
 * two together (lines 18..25 that no primary line names are unwritten).  Lines 26..28: the 7 / 14 / 21
 * points of other leaves nearest to the region whatever faces they lie beyond; x[7] of line 26 + k: every
 * point of another leaf nearer than that is in the lines 26 .. 26 + k.  Call mi_icp_debug_get_leaf_regions
 * afterwards for the regions. */
MI_ICP_API int mi_icp_debug_get_leaf_halos(mi_icp_ctx* ctx, float* halos_out);
#ifdef __cplusplus
}
#endif
#endif
