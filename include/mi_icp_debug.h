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
 * loop's first pass does when the target's halos exist already), -1 = no pass yet. */
MI_ICP_API int mi_icp_debug_last_search_kind(const mi_icp_ctx* ctx);
/* Which path the last mi_icp_voxel_downsample call took: 1 = the dense-grid path (csrc/voxel_dense.h: one partition,
 * a workgroup per bucket), 0 = the general path (radix passes with the payload), -1 = no call yet (or one that returned
 * early: an empty cloud, a voxel size <= 0, a grid beyond int32).  Both are exact; tests run each on purpose
 * (the switch MI_ICP_NO_DENSE_VOXEL is read at every call). */
MI_ICP_API int mi_icp_debug_last_voxel_path(const mi_icp_ctx* ctx);
/* Resident workgroups per CU the runtime grants a kernel at its launch shape (hipOccupancyMaxActiveBlocksPerMultiprocessor):
 * which = 0 kd_build_groups, 1 nn_packet_kernel<seeded>, 2 nn_packet_kernel<from the root>, 3 reduce_pt2pl_kernel<4,1>,
 * 4 leaf_halo_build, 5 rs_scatter_pay<8>, 6 voxel_means_wave, 7 vx_scatter<1>, 8 vx_finish<points only>.  Returns the
 * count, < 0 on error. */
MI_ICP_API int mi_icp_debug_occupancy(int which);
/* Where an iteration's time goes (csrc/loop.h): with stamps enabled the NEXT registration loop on the context runs the
 * same search / point-to-plane reduction kernels instantiated with device-clock stamps (s_memrealtime, one clock for the
 * whole GPU): the search's first wave start / last wave end, the reduction's first block start, the last block's
 * ticket, rows totalled, ranks' exchange done, solve done, state written.  get: the 32 stamp words -- [0..7] the
 * current iteration's stamps (re-armed), [16..23] the SUMS of the eight spans over the iterations counted in [24]:
 * step-end -> next search start, search, search end -> reduction start, reduction's streaming phase, row total,
 * exchange, solve, state write -- and the clock's ticks per microsecond. */
/* Counters of the present registration loop as of its last host-side look: out4 = {iterations (updates applied),
 * passes (evaluations), re-locations (steps that moved the source by more than about a leaf's width, ~1.9 point spacings, and had the next
 * search's seeds replaced by the leaves the moved queries fall into: csrc/loop.h, nn_search.h locate_by_planes),
 * 1 if the next chunk of iterations would still carry the gated re-location launches}. */
MI_ICP_API int mi_icp_debug_loop_counters(mi_icp_ctx* ctx, int32_t* out4);
/* The leaf every staged source point FALLS INTO under T (column-major 4x4 or NULL) by the binary descent through the
 * cell planes and its group's planes (nn_search.h locate_by_planes): leaf_out[original source index] = leaf (host memory,
 * one per source point).  The located leaves are left behind as the seeds of the next seeded pass.  Fails on a tree
 * without planes (MI_ICP_NO_CELLS, a target below one group). */
MI_ICP_API int mi_icp_debug_locate(mi_icp_ctx* ctx, const float* T, int32_t* leaf_out);
MI_ICP_API int mi_icp_debug_set_step_stamps(mi_icp_ctx* ctx, int enable);
MI_ICP_API int mi_icp_debug_get_step_stamps(mi_icp_ctx* ctx, uint64_t* out32, double* ticks_per_us);
/* The loop step's two forms of utility::SolveJacobianSystemAndObtainExtrinsicMatrix side by side, on the
 * device: n systems of 32 doubles each (host memory; the reduction's layout: 21 upper-triangle sums of
 * JtJ, 6 of Jtr, ...) are solved by one thread with the serial routines and by a wave with a matrix row
 * per lane.  out_serial / out_wave: n * 16 floats (column-major 4x4); ok_serial / ok_wave: n flags
 * (0 = the determinant check failed, result identity).  The two must agree bit for bit. */
MI_ICP_API int mi_icp_debug_solve_both(int device, const double* systems, int n, float det_thresh,
                                       float* out_serial, float* out_wave, int32_t* ok_serial, int32_t* ok_wave);
/* csrc/eigen3.h one matrix at a time, so that an outside implementation (LAPACK through numpy, in fp64) can be held
 * against it: n symmetric 3x3 matrices A (row-major, host memory) -> eval (n * 3), evec (n * 9 row-major, COLUMN k =
 * eigenvector k), S (n * 9: what the GICP reduction adds for W = SqrtMatrix3x3(A), S = W W) as FastEigen3x3 /
 * gicp_weight compute them -- on the host (device < 0: the __host__ half of the same functions, no GPU needed) or in
 * a kernel on that device (the GPU's own acosf / cosf / sqrtf / divisions).  Any output may be NULL. */
MI_ICP_API int mi_icp_debug_eigen3(int device, const float* A, int64_t n, float* eval, float* evec, float* S);
/* The target's tree as built by mi_icp_set_target, for invariant tests.  info5 = {slots
 * (padded sorted positions), leaves, leaf_first (id of the first leaf-level node), records,
 * points}.  records_out (records * 64 floats: 8 child boxes as 4 sibling pairs of 12,
 * floats 48..53 the node's region, 54 its validity flag) and leaf_lines_out (leaves * 32
 * floats: x[8] y[8] z[8] original index[8], padding = +inf / -1; on the device a line's fourth
 * row holds the leaf's region record and the indices are an array of their own -- the export
 * puts the indices there) may be NULL to query the sizes only. */
MI_ICP_API int mi_icp_debug_get_tree(mi_icp_ctx* ctx, int64_t* info5, float* records_out,
                                     float* leaf_lines_out);
/* Per leaf (mi_icp_debug_get_tree's info5[1] leaves) 8 floats: region lo.xyz, word A, region hi.xyz,
 * word B.  The region is free of points of any other leaf (an invalid one is +inf / -inf: nothing is
 * inside).  Words A and B pack the reaches of the leaf's eight halo lines as 6-bit fractions of the bound:
 * A = q0 .. q4 (6 bits each from bit 0), bits 30-31 the low two bits of q7; B = q5 (bits 0-5), q6 (6-11),
 * the high four bits of q7 (12-15), the bound (the upper 16 bits of an fp32) on top; reach k = bound / 64 *
 * q_k.  B = 0: no halo. */
MI_ICP_API int mi_icp_debug_get_leaf_regions(mi_icp_ctx* ctx, float* regions_out);
/* Per leaf 8 halo lines of 32 floats (builds them if no seeded search has yet): x[8] y[8] z[8] slot[8] --
 * line k holds the points of OTHER leaves that are 8k+1-th .. 8k+8-th nearest to the leaf's region
 * (L-infinity distance to the box), ascending in slot (sorted target position, as an integer's bits; unused
 * entries +inf / -1).  Every point of another leaf nearer to the region than reach k is in the lines 0 .. k.
 * Call mi_icp_debug_get_leaf_regions afterwards for the regions and reaches. */
MI_ICP_API int mi_icp_debug_get_leaf_halos(mi_icp_ctx* ctx, float* halos_out);
#ifdef __cplusplus
}
#endif
#endif
