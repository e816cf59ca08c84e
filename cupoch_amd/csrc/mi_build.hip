// mi_build.hip -- the target tree (kd cells -> 4096-slot groups -> 8-ary levels, regions, halos), the staging of the
// source cloud and its match-order re-sort: mi_icp_set_target / mi_icp_set_source / mi_icp_spatial_order
// (one translation unit of libmi_icp.so; csrc/ctx.h lists them)
#include "ctx.h"
#include "kd_build.h"
#include "kd_cells.h"
#include "kd_planes.h"
#include "kd_refine.h"
#include "lbvh.h"
#include "leaf_halo.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

namespace mi {
namespace eng {

// Morton grid: 2^bits cells per axis, ~4 per mean point spacing -- fine enough that almost every
// point has a cell of its own (ties keep the input order).  Every 8 key bits are a radix pass, so the
// grid is coarsened to the pass boundary below as long as that leaves >= 1 cell per mean spacing:
// packets of 64 consecutive points stay as compact (10M points: 24-bit keys, 3 passes instead of 4).
int morton_bits_for(int64_t n) {
    int lg = 0;
    while ((1ll << lg) < n) ++lg;
    const int per_axis = (lg + 2) / 3;
    const int fine = std::min(21, std::max(6, per_axis + 2));
    const int coarse = (((3 * fine + 7) / 8 - 1) * 8) / 3;
    return coarse >= std::max(6, per_axis) ? coarse : fine;
}

// bounds (min/max/extent) of an AoS cloud into c->bounds (8 floats, device)
int compute_bounds(mi_icp_ctx* c, const float* pts, int64_t n, float** bounds_out) {
    float *part, *bnd;
    TRY(ensure(c, c->bounds_part, (size_t)kBoundsBlocks * 6, &part));
    TRY(ensure(c, c->bounds, 8, &bnd));
    const int nb = std::min<int64_t>(kBoundsBlocks, blocks_for(n));
    bounds_partial<<<nb, 256, 0, c->stream>>>(pts, (int)n, part);
    KCHK(c);
    bounds_final<<<1, 64, 0, c->stream>>>(part, nb, bnd);
    KCHK(c);
    *bounds_out = bnd;
    return MI_ICP_OK;
}

int sort_buffers(mi_icp_ctx* c, int64_t n, SortBuffers* sb) {
    // the buffers also serve sorts of FEWER elements (samples), which may use smaller tiles
    int nseg = sort_num_segments(n);
    nseg = std::max(nseg, sort_num_segments(std::min<int64_t>(n, (1 << 21) - 1)));
    nseg = std::max(nseg, sort_num_segments(std::min<int64_t>(n, (1 << 18) - 1)));
    // (the payload-carrying sort of VoxelDownSample works on tiles of at most 4096 elements)
    nseg = std::max(nseg, sort_pay_num_segments(n));
    nseg = std::max(nseg, sort_pay_num_segments(std::min<int64_t>(n, (1 << 20) - 1)));
    nseg = std::max(nseg, sort_pay_num_segments(std::min<int64_t>(n, (1 << 18) - 1)));
    TRY(ensure(c, c->keys0, (size_t)n, &sb->keys[0]));
    TRY(ensure(c, c->keys1, (size_t)n, &sb->keys[1]));
    TRY(ensure(c, c->vals0, (size_t)n, &sb->vals[0]));
    TRY(ensure(c, c->vals1, (size_t)n, &sb->vals[1]));
    TRY(ensure(c, c->hist, (size_t)256 * nseg, &sb->hist));
    TRY(ensure(c, c->scan_tmp, (size_t)std::max(scan_num_tiles((int64_t)256 * nseg), scan_num_tiles(n)) + 2,
               &sb->scan_tmp));
    return MI_ICP_OK;
}

// Morton order of an AoS cloud: returns the device array order[sorted] = original.
// grid_bounds/grid_bits: quantise on another cloud's grid instead of the cloud's own.
// kd_refine: also split every group of 4096 Morton-consecutive points into kd cells (kd_refine.h)
int morton_order(mi_icp_ctx* c, const float* pts, int64_t n, const uint32_t** order, bool kd_refine,
                 const float* grid_bounds, int grid_bits, float** own_bounds) {
    float* bnd = nullptr;
    if (!grid_bounds || own_bounds) TRY(compute_bounds(c, pts, n, &bnd));
    if (own_bounds) *own_bounds = bnd;
    if (grid_bounds) bnd = const_cast<float*>(grid_bounds);
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    const int bits = grid_bounds ? grid_bits : morton_bits_for(n);
    int cur;
    if (3 * bits <= 32) {  // narrow keys: a third less traffic per pass
        morton_keys<uint32_t><<<blocks_for(n), 256, 0, c->stream>>>(pts, (int)n, bnd, bits, (uint32_t*)sb.keys[0], sb.vals[0]);
        KCHK(c);
        cur = radix_sort_pairs32(c->stream, sb, n, 3 * bits);
    } else {
        morton_keys<uint64_t><<<blocks_for(n), 256, 0, c->stream>>>(pts, (int)n, bnd, bits, sb.keys[0], sb.vals[0]);
        KCHK(c);
        cur = radix_sort_pairs(c->stream, sb, n, 3 * bits);
    }
    KCHK(c);
    if (!kd_refine) {
        *order = sb.vals[cur];
        return MI_ICP_OK;
    }
    // Morton runs -> kd cells inside every group of 4096 points (kd_refine.h)
    const int ngroups = (int)((n + kKdGroup - 1) / kKdGroup);
    kd_refine_groups<<<ngroups, kKdThreads, 0, c->stream>>>(pts, sb.vals[cur], sb.vals[cur ^ 1], n);
    KCHK(c);
    *order = sb.vals[cur ^ 1];
    return MI_ICP_OK;
}

// kd-cell layout of the target (kd_cells.h): point indices sorted by cell, the cells'
// first positions and first groups.  One host synchronisation (the number of groups sizes
// the tree).
struct CellLayout {
    const uint32_t* vals;
    const uint32_t* cstart;
    const uint32_t* gstart;
    int ncells;
    int64_t ngroups;
    const float2* planes;  // split planes, heap order
    int levels;            // depth of the plane tree + layout flag (kd_descend.h)
};

int kd_cell_layout(mi_icp_ctx* c, const float* pts, int64_t n, CellLayout* out) {
    // (MI_ICP_CELL_LAYOUT, A/B switch: "pow2" -- 2^d cells only, still filled to 80 %; "r4" -- 2^d cells at <= 2/3 fill,
    // rounds 2-4's layout, on this round's planes)
    static const int forced = [] { const char* e = std::getenv("MI_ICP_CELL_LAYOUT"); return !e ? 0 : (std::strcmp(e, "pow2") == 0 ? 1 : (std::strcmp(e, "r4") == 0 ? 2 : 0)); }();
    const int lv = forced == 0 ? cell_layout_for(n) : cell_layout_pow2(n, forced == 2 ? 2731 : kCellTargetFill);
    const int d = cell_depth(lv);
    const int ncells = (int)cell_count(lv);
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    float2* planes;
    TRY(ensure(c, c->cell_planes, (size_t)2 << d, &planes));
    if (d > 0) {
        // the planes (kd_planes.h): histogram levels over a quarter of the sample, then the last three levels from all
        // of a node's samples in LDS
        const int64_t S = std::min<int64_t>(n, (int64_t)kPlaneSamples * ncells);
        float* samp;
        TRY(ensure(c, c->cell_samples, (size_t)S * 3, &samp));
        cells_sample_gather<<<blocks_for(S), 256, 0, c->stream>>>(pts, n, S, samp);
        KCHK(c);
        // histogram levels (a TRI layout's first two -- the 1/3 cut, the dummy -- are no medians: always histogram levels)
        const int dh = std::max(cell_tri(lv) ? 2 : 0, d - kPlaneLdsLevels);
        if (dh > 0) {
            uint32_t *boxmin, *boxmax, *hist;
            const size_t nbox = (size_t)4 << dh;  // [node < 2^dh][4]
            const size_t nhist = std::max<size_t>((size_t)kPlaneBinBudget, ((size_t)1 << (dh - 1)) * kPlaneMinBins);
            TRY(ensure(c, c->cell_boxes, nbox * 2, &boxmin));
            boxmax = boxmin + nbox;
            TRY(ensure(c, c->cell_hist, nhist, &hist));
            HIPCHK(c, hipMemsetAsync(boxmin, 0xff, nbox * sizeof(uint32_t), c->stream));
            HIPCHK(c, hipMemsetAsync(boxmax, 0, nbox * sizeof(uint32_t), c->stream));
            HIPCHK(c, hipMemsetAsync(hist, 0, nhist * sizeof(uint32_t), c->stream));
            uint32_t* snode = reinterpret_cast<uint32_t*>(sb.keys[1]);  // (free until the samples / the points' cell ids are sorted, below)
            const int stride = (S >= (int64_t)kPlaneStride * 65536) ? kPlaneStride : 1;  // (small clouds: every sample)
            const int64_t Sh = (S + stride - 1) / stride;
            const int sgrid = (int)std::min<int64_t>(blocks_for(Sh), 2048);
            const int tri = cell_tri(lv) ? 1 : 0;
            for (int l = 0; l < dh; ++l) {
                const int bins = plane_bins(l);
                if (l < kPlaneExactBoxLevels) {
                    hp_assign_bbox<<<std::min(sgrid, 256), 256, 0, c->stream>>>(samp, Sh, stride, planes, snode, l, boxmin, boxmax);
                    KCHK(c);
                }
                hp_hist<<<sgrid, 256, 0, c->stream>>>(samp, Sh, stride, planes, snode, l, l >= kPlaneExactBoxLevels ? 1 : 0, boxmin, boxmax,
                                                      hist, bins);
                KCHK(c);
                hp_select<<<1 << l, 64, 0, c->stream>>>(planes, l, tri, boxmin, boxmax, hist, bins,
                                                        (l + 1 >= kPlaneExactBoxLevels && l + 1 < dh) ? 1 : 0);
                KCHK(c);
            }
        }
        // the levels dh .. d - 1: every node of depth dh sorts its samples in LDS
        int cur = 0;
        if (dh > 0) {  // samples grouped by their depth-dh node (plain heap numbering: no layout flag; narrow keys)
            cells_assign<uint32_t><<<blocks_for(S), 256, 0, c->stream>>>(samp, S, planes, dh, (uint32_t*)sb.keys[0], sb.vals[0]);
            KCHK(c);
            cur = radix_sort_pairs32(c->stream, sb, S, dh);
            KCHK(c);
        }
        if (d > dh) {
            hp_last_levels<uint32_t><<<1 << dh, 256, 0, c->stream>>>(samp, S, (const uint32_t*)sb.keys[cur], sb.vals[cur], dh, d - dh, planes);
            KCHK(c);
        }
    }
    uint32_t *cstart, *gstart;
    TRY(ensure(c, c->cell_cstart, (size_t)ncells + 2, &cstart));
    TRY(ensure(c, c->cell_gstart, (size_t)ncells, &gstart));
    cells_assign<uint32_t><<<blocks_for(n), 256, 0, c->stream>>>(pts, n, planes, lv, (uint32_t*)sb.keys[0], sb.vals[0]);
    KCHK(c);
    const int cur = radix_sort_pairs32(c->stream, sb, n, d);  // (cell ids: narrow keys)
    KCHK(c);
    cells_starts<uint32_t><<<blocks_for(n), 256, 0, c->stream>>>((const uint32_t*)sb.keys[cur], n, ncells, cstart);
    KCHK(c);
    cells_layout<<<1, 1024, 0, c->stream>>>(cstart, ncells, gstart, cstart + ncells + 1);
    KCHK(c);
    if (!c->cell_total_host) HIPCHK(c, hipHostMalloc((void**)&c->cell_total_host, 64, hipHostMallocDefault));
    HIPCHK(c, hipMemcpyAsync(c->cell_total_host, cstart + ncells + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t ngroups = (int64_t)c->cell_total_host[0];
    if (ngroups <= 0 || ngroups > (int64_t)ncells + n / kKdGroup + 1)
        return fail(c, MI_ICP_ERR_HIP, "kd cell layout: implausible group count %lld", (long long)ngroups);
    out->vals = sb.vals[cur];
    out->cstart = cstart;
    out->gstart = gstart;
    out->ncells = ncells;
    out->ngroups = ngroups;
    out->planes = planes;
    out->levels = lv;
    return MI_ICP_OK;
}

// Every leaf's halo (leaf_halo.h): what lets a seeded query whose cube pokes out of its leaf's region
// finish without a tree walk.  Built once per target: right behind the tree on a context that has
// registered before (mi_icp_set_target), otherwise by the first registration loop / seeded search
// (one-shot searches, k-NN and normal estimation on a fresh context never pay for it).
int build_links(mi_icp_ctx* c, hipStream_t st) {
    static const bool no_links = std::getenv("MI_ICP_NO_LINKS") != nullptr;  // A/B switch
    if (!c->links_allowed || no_links) return MI_ICP_OK;  // (every leaf's largest reach is 0 as built: no query asks for a line)
    float* halo;
    const size_t ntiles = ((size_t)c->nleaf + 63) / 64;
    TRY(ensure(c, c->thalo, ntiles * 64 * kHaloStored * kHaloLineFloats, &halo));
    uint2* cand;  // scratch: up to 64 candidate leaves per leaf
    TRY(ensure(c, c->tlinks_tmp, ntiles * 64 * kLinkCand, &cand));
    const uint32_t lblocks = (uint32_t)ntiles;
    leaf_halo_collect<<<((lblocks + 7u) / 8u) * 8u, 64, 0, st>>>(
            (const float*)c->nodes.p, c->leaf_first, c->nleaf, lblocks, lreg_of(c), cand);
    KCHK(c);
    leaf_halo_build<<<(unsigned)(((size_t)c->nleaf + kHaloTile - 1) / kHaloTile), 64, 0, st>>>(lreg_of(c), c->nleaf, cand, (const float*)c->tblk.p, halo);
    KCHK(c);
    return MI_ICP_OK;
}

// The halos must be complete before the next kernel on the context's stream reads them (one-shot searches,
// tests, the debug export: builds them on the spot if nobody has yet).
int ensure_links(mi_icp_ctx* c) {
    if (c->nt <= 0) return MI_ICP_OK;
    if (c->links_inflight) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_links, 0));
        c->links_inflight = false;
        c->links_ready = true;
    }
    if (c->links_ready) return MI_ICP_OK;
    TRY(build_links(c, c->stream));
    c->links_ready = true;
    return MI_ICP_OK;
}

// The build's candidate scratch (512 B per leaf: 0.9 GB for a 10M-point target) is dead once the halos are complete.
// Called where the device is idle anyway (the end of a registration call): hipFree synchronises.  Small targets keep
// theirs -- frame-to-frame callers would pay an allocation per frame.
void release_links_scratch(mi_icp_ctx* c) {
    constexpr size_t kKeepBelow = (size_t)64 << 20;
    if (c->links_ready && !c->links_inflight && c->tlinks_tmp.p && c->tlinks_tmp.bytes >= kKeepBelow) release(c->tlinks_tmp);
}

// Are they there?  Never waits: a build in flight counts once its event has completed.
bool halo_poll(mi_icp_ctx* c) {
    if (c->links_inflight && hipEventQuery(c->ev_links) == hipSuccess) {
        c->links_inflight = false;
        c->links_ready = true;
    }
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    static const bool no_links = std::getenv("MI_ICP_NO_LINKS") != nullptr;
    return c->links_ready && c->links_allowed && !no_links && c->thalo.p != nullptr;
}

// Start the build on the private stream (behind everything enqueued on the context's stream so far); the
// registration loop goes on meanwhile and uses the halos from the first chunk of iterations that finds them done.
int start_links_async(mi_icp_ctx* c) {
    if (c->nt <= 0 || c->links_ready || c->links_inflight || !c->links_allowed) return MI_ICP_OK;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    TRY(build_links(c, c->side));
    HIPCHK(c, hipEventRecord(c->ev_links, c->side));
    c->links_inflight = true;
    return MI_ICP_OK;
}

// A new target: nothing of the old one may still be read or written by the private stream.
int drain_links(mi_icp_ctx* c) {
    if (c->links_inflight) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_links, 0));
        HIPCHK(c, hipStreamSynchronize(c->side));
        c->links_inflight = false;
    }
    c->links_ready = false;
    return MI_ICP_OK;
}

// ---- device-resident registration loop (loop.h) -------------------------------------------
// Re-order the staged source by its current matches (lbvh.h: match_order_keys).
// Enqueue-only; the second set of source arrays becomes the live one.
int resort_source_by_match(mi_icp_ctx* c) {
    const int64_t n = c->ns;
    if (n <= 0 || c->nt <= 0 || !c->nn_valid) return MI_ICP_OK;
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    // the key is the matched LEAF (the order inside a leaf does not matter to a packet, and
    // three bits less can save a radix pass); leaves 0..nleaf-1, nleaf = unmatched
    int bits = 1;
    while (bits < 32 && (1ull << bits) <= (uint64_t)c->nleaf) ++bits;
    match_order_keys<<<blocks_for(n), 256, 0, c->stream>>>((const int32_t*)c->nn_idx.p, (int)n, (uint32_t)c->nleaf,
                                                           (uint32_t*)sb.keys[0], sb.vals[0]);
    KCHK(c);
    const uint32_t* ord = sb.vals[radix_sort_pairs32(c->stream, sb, n, bits)];
    KCHK(c);
    SourceArrays in, out;
    in.sx = (float*)c->sx.p; in.sy = (float*)c->sy.p; in.sz = (float*)c->sz.p;
    in.sperm = (int32_t*)c->sperm.p;
    in.snrm = c->s_has_nrm ? (float4*)c->snrm.p : nullptr;
    in.scov = c->s_has_cov ? (float*)c->scov.p : nullptr;
    in.sint = c->s_has_int ? (float*)c->sint.p : nullptr;
    in.nn_idx = (int32_t*)c->nn_idx.p; in.nn_d2 = (float*)c->nn_d2.p;
    TRY(ensure(c, c->alt[0], (size_t)n, &out.sx));
    TRY(ensure(c, c->alt[1], (size_t)n, &out.sy));
    TRY(ensure(c, c->alt[2], (size_t)n, &out.sz));
    TRY(ensure(c, c->alt[3], (size_t)n, &out.sperm));
    TRY(ensure(c, c->alt[4], (size_t)n, &out.nn_idx));
    TRY(ensure(c, c->alt[5], (size_t)n, &out.nn_d2));
    out.snrm = nullptr;
    out.scov = nullptr;
    out.sint = nullptr;
    if (in.snrm) TRY(ensure(c, c->alt[6], (size_t)n, &out.snrm));
    if (in.scov) TRY(ensure(c, c->alt[7], (size_t)n * 9, &out.scov));
    if (in.sint) TRY(ensure(c, c->alt[8], (size_t)n, &out.sint));
    permute_source<<<blocks_for(n), 256, 0, c->stream>>>(ord, (int)n, in, out);
    KCHK(c);
    std::swap(c->sx, c->alt[0]);
    std::swap(c->sy, c->alt[1]);
    std::swap(c->sz, c->alt[2]);
    std::swap(c->sperm, c->alt[3]);
    std::swap(c->nn_idx, c->alt[4]);
    std::swap(c->nn_d2, c->alt[5]);
    if (in.snrm) std::swap(c->snrm, c->alt[6]);
    if (in.scov) std::swap(c->scov, c->alt[7]);
    if (in.sint) std::swap(c->sint, c->alt[8]);
    c->inv_s_valid = false;
    return MI_ICP_OK;
}

int occupancy_build(int which) {
    int blocks = -1;
    hipError_t e = hipErrorInvalidValue;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kd_build_groups, kKdThreads, 0);
    else if (which == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, leaf_halo_build, 64, 0);
    else return -1;
    return e == hipSuccess ? blocks : -2;
}

}  // namespace eng
}  // namespace mi

constexpr int64_t kHaloAheadMax = 2000000;  // targets below this get their halos right behind the tree on a context that has registered before

extern "C" {

// ---------------------------------------------------------------------------
int mi_icp_set_target(mi_icp_ctx* c, const float* xyz, const float* normals, const float* covs,
                      int64_t n, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0 || n > 0x7fffff00ll || (n > 0 && !xyz)) return fail(c, MI_ICP_ERR_INVALID, "set_target: bad size/pointer");
    TRY(drain_links(c));
    c->nt = 0;
    c->inv_t_valid = false;
    c->nn_valid = false;
    c->n_user_pairs = -1;
    c->loop_active = false;  // a stepping loop (icp_begin / icp_iterate) belongs to the clouds it started on
    c->t_has_nrm = normals != nullptr && n > 0;
    c->t_has_cov = covs != nullptr && n > 0;
    c->t_has_int = c->t_has_grad = false;
    c->t_has_rec = false;
    if (n == 0) return MI_ICP_OK;
    hipEvent_t e0 = c->ev[2], e1 = c->ev[3];
    if (c->profiling) {
        (void)hipStreamSynchronize(c->stream);
        collect_events(c);
        (void)hipEventRecord(e0, c->stream);
    }

    const float *d_pts, *d_nrm, *d_cov;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[0], &d_pts));
    TRY(to_device(c, normals, (size_t)n * 3, mem_kind, c->stage[1], &d_nrm));
    TRY(to_device(c, covs, (size_t)n * 9, mem_kind, c->stage[2], &d_cov));

    static const bool no_cells = std::getenv("MI_ICP_NO_CELLS") != nullptr;  // A/B switch: Morton runs on top
    const uint32_t* order = nullptr;
    CellLayout lay = {};
    int64_t nts = n;
    if (no_cells) {
        TRY(morton_order(c, d_pts, n, &order, true));
    } else {
        TRY(kd_cell_layout(c, d_pts, n, &lay));
        nts = lay.ngroups * kKdGroup;
    }

    const int nleaf = (int)((nts + kLeaf - 1) / kLeaf);
    int levels = 1;  // 8-ary levels of records above the leaves
    uint32_t leaf_first = 1u;
    while ((uint64_t)leaf_first * 8u < (uint64_t)nleaf) {
        leaf_first *= 8u;
        ++levels;
    }
    if (levels > kMaxLevels) return fail(c, MI_ICP_ERR_INVALID, "set_target: cloud too large for the 64-bit traversal stack");
    const uint32_t used_last = (uint32_t)((nleaf + 7) / 8);
    const uint32_t nrecords = full_levels_below(leaf_first) + used_last;
    if ((uint64_t)nrecords * kRecordFloats * sizeof(float) >= (1ull << 32))
        return fail(c, MI_ICP_ERR_INVALID, "set_target: cloud too large for 32-bit record offsets");
    float* tblk;
    float4* tnrm = nullptr;
    float* tcov = nullptr;
    float* nodes;
    TRY(ensure(c, c->tblk, (size_t)nleaf * kLeafFloats, &tblk));
    TRY(ensure(c, c->nodes, (size_t)nrecords * kRecordFloats, &nodes));
    float* lreg = tblk + kLeafRegOffset;  // the region records: fourth row of every leaf line (device_utils.h)
    int32_t* tidx;
    TRY(ensure(c, c->tidx, (size_t)nleaf * kLeaf, &tidx));
    float* trec = nullptr;
    if (d_nrm) TRY(ensure(c, c->tnrm, (size_t)nts, &tnrm));
    if (d_nrm) TRY(ensure(c, c->trec, (size_t)nts * 6, &trec));
    c->t_has_rec = trec != nullptr;
    if (d_cov) TRY(ensure(c, c->tcov, (size_t)nts * 9, &tcov));
    uint32_t first, used;  // the level whose nodes' boxes still have to be formed from their records
    // nodes above the groups are kd subtrees -- disjoint boxes -- when every cell has exactly one group
    const uint32_t upper_flag = (!no_cells && lay.ngroups == (int64_t)lay.ncells) ? 1u : 0u;
    if (no_cells) {
        // own boxes / flags of the leaf-level records stay zero: no early stop on a Morton-run tree
        HIPCHK(c, hipMemsetAsync(nodes, 0, (size_t)nrecords * kRecordFloats * sizeof(float), c->stream));
        fill_invalid_leaf_regions<<<blocks_for(nleaf), 256, 0, c->stream>>>(lreg, nleaf);  // no leaf regions either
        KCHK(c);
        const int nslots = (int)used_last * 8;
        build_leaves<<<blocks_for(nslots), 256, 0, c->stream>>>(order, d_pts, d_nrm, d_cov, nts, nleaf, nslots,
                                                                leaf_first, tblk, tnrm, tcov, nodes, trec, tidx);
        KCHK(c);
        first = leaf_first;
        used = used_last;
    } else {
        GroupBuildArgs ga;
        ga.pts = d_pts;
        ga.nrm = d_nrm;
        ga.cov = d_cov;
        ga.vals = lay.vals;
        ga.cstart = lay.cstart;
        ga.gstart = lay.gstart;
        ga.ncells = lay.ncells;
        ga.planes = lay.planes;
        ga.cell_levels = lay.levels;
        ga.ngroups = (uint32_t)lay.ngroups;
        ga.leaf_first = leaf_first;
        ga.tblk = tblk;
        ga.tnrm = tnrm;
        ga.trec = trec;
        ga.tcov = tcov;
        ga.records = nodes;
        ga.lreg = lreg;
        ga.tidx = tidx;
        TRY(ensure(c, c->gplanes, (size_t)lay.ngroups * 512, &ga.gplanes));
        ga.link_delta = 0.25f;     // the halos' bound: a quarter of the leaf-level node's size (kd_build.h)
        ga.region_margin = 0.5f;   // a leaf's region stays within half that bound of its own box
        kd_build_groups<<<(unsigned)lay.ngroups, kKdThreads, 0, c->stream>>>(ga);
        KCHK(c);
        first = leaf_first >> 9;  // the groups' own boxes sit in the records of this level
        used = ((uint32_t)lay.ngroups + 7u) / 8u;
    }
    int above_groups = 1;  // 8-ary levels between `first` and the groups' level
    for (; first > 1u; first /= 8u, ++above_groups) {
        const uint32_t count = ((used + 7u) / 8u) * 8u;
        // (a TRI layout's nodes are kd subtrees only while they lie inside one of its three parts: 8^above_groups
        // cells <= 2^(depth - 2); the one or two levels above that keep their points' boxes, no early stop there)
        const int region_depth = cell_depth(lay.levels) - 3 * above_groups;
        const uint32_t flag = (upper_flag && (!cell_tri(lay.levels) || region_depth >= 2)) ? 1u : 0u;
        build_level<<<blocks_for(count), 256, 0, c->stream>>>(nodes, first, used, count, flag, lay.planes,
                                                              lay.levels, region_depth);
        KCHK(c);
        used = (used + 7u) / 8u;
    }
    {   // the cap of the wave-uniform walks' cubes, from the leaf-level nodes' sizes (kd_build.h tree_scale)
        float* ts;
        TRY(ensure(c, c->tscale, 4, &ts));
        HIPCHK(c, hipMemsetAsync(ts, 0, 16, c->stream));
        tree_scale<<<std::min(256u, (used_last + 255u) / 256u), 256, 0, c->stream>>>(nodes, leaf_first, used_last, ts);
        KCHK(c);
    }
    c->links_ready = false;  // (the leaves' halos: started below, or by the registration loop / the first seeded search)
    c->halo_iters = c->halo_asked = c->halo_lanes = 0;
    c->links_allowed = !no_cells && (uint32_t)nleaf <= kLinkIdMask;
    c->nt = n;
    c->nts = nts;
    c->nleaf = nleaf;
    c->leaf_first = leaf_first;
    c->nrecords = nrecords;
    c->cell_levels = no_cells ? -1 : lay.levels;
    // A context whose loops have ASKED for halos will run another such loop.  For a small target (frame-to-frame
    // callers: KinFu, odometry) the halos are then started right away, on the private stream, next to the staging of the
    // source: the loop's first seeded iterations find them ready.  For a large one the build would fight the staging
    // for the memory system; there the loop's own searches say whether it is wanted.  (Until round 6 ANY earlier loop on
    // the context was enough: on clean frames the 0.18-ms build per pyramid level cost a KinFu step 2-5 % and a
    // 100k-300k-point call 4 %, same box, for halos nothing read.)
    if (c->ran_loop && c->halo_sticky && c->links_allowed && n < kHaloAheadMax) TRY(start_links_async(c));
    if (c->profiling) {
        (void)hipEventRecord(e1, c->stream);
        (void)hipStreamSynchronize(c->stream);
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->prof[4] = ms;
    }
    return MI_ICP_OK;
}

int mi_icp_set_source(mi_icp_ctx* c, const float* xyz, const float* normals, const float* covs,
                      int64_t n, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0 || n > 0x7fffff00ll || (n > 0 && !xyz)) return fail(c, MI_ICP_ERR_INVALID, "set_source: bad size/pointer");
    c->ns = 0;
    c->inv_s_valid = false;
    c->nn_valid = false;
    c->n_user_pairs = -1;
    c->loop_active = false;
    c->s_has_nrm = normals != nullptr && n > 0;
    c->s_has_cov = covs != nullptr && n > 0;
    c->s_has_int = false;
    if (c->nranks == 1) c->ns_global = 0;
    if (n == 0) return MI_ICP_OK;
    hipEvent_t e0 = c->ev[2], e1 = c->ev[3];
    if (c->profiling) {
        (void)hipStreamSynchronize(c->stream);
        collect_events(c);
        (void)hipEventRecord(e0, c->stream);
    }

    const float *d_pts, *d_nrm, *d_cov;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[3], &d_pts));
    TRY(to_device(c, normals, (size_t)n * 3, mem_kind, c->stage[4], &d_nrm));
    TRY(to_device(c, covs, (size_t)n * 9, mem_kind, c->stage[5], &d_cov));

    // Packets are 64 consecutive points of this order.  It only has to make the packets of the
    // FIRST (unseeded) pass compact: the loop re-sorts the source by match right after it.
    // (Measured: the in-group kd split on top of the Morton order, kd_refine.h, costs more here -- 1.2 ms at
    // 10M -- than it saves in that one pass, 0.15 ms.)
    const uint32_t* order;
    float* own_bounds = nullptr;
    TRY(morton_order(c, d_pts, n, &order, false, nullptr, 0, &own_bounds));
    {   // (c->bounds is every build's scratch: the source's box is kept for the loop, loop.h "re-location")
        float* sb;
        TRY(ensure(c, c->src_bounds, 8, &sb));
        HIPCHK(c, hipMemcpyAsync(sb, own_bounds, 8 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }

    float *sx, *sy, *sz, *scov = nullptr, *d2;
    int32_t *sperm, *idx;
    float4* snrm = nullptr;
    TRY(ensure(c, c->sx, (size_t)n, &sx));
    TRY(ensure(c, c->sy, (size_t)n, &sy));
    TRY(ensure(c, c->sz, (size_t)n, &sz));
    TRY(ensure(c, c->sperm, (size_t)n, &sperm));
    TRY(ensure(c, c->nn_idx, (size_t)n, &idx));
    TRY(ensure(c, c->nn_d2, (size_t)n, &d2));
    if (d_nrm) TRY(ensure(c, c->snrm, (size_t)n, &snrm));
    if (d_cov) TRY(ensure(c, c->scov, (size_t)n * 9, &scov));
    gather_source<<<blocks_for(n), 256, 0, c->stream>>>(order, d_pts, d_nrm, d_cov, (int)n, sx, sy, sz,
                                                        sperm, snrm, scov);
    KCHK(c);
    c->ns = n;
    if (c->profiling) {
        (void)hipEventRecord(e1, c->stream);
        (void)hipStreamSynchronize(c->stream);
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->prof[5] = ms;
    }
    return MI_ICP_OK;
}

int mi_icp_spatial_order(mi_icp_ctx* c, const float* xyz, int64_t n, uint32_t* order_out, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0 || n > 0x7fffff00ll || (n > 0 && (!xyz || !order_out)))
        return fail(c, MI_ICP_ERR_INVALID, "spatial_order: bad arguments");
    if (n == 0) return MI_ICP_OK;
    const float* d_pts;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[0], &d_pts));
    const uint32_t* order;
    TRY(morton_order(c, d_pts, n, &order, false));
    TRY(from_device(c, order, order_out, (size_t)n, mem_kind));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

int mi_icp_debug_morton_order(mi_icp_ctx* c, const float* xyz, int64_t n, uint32_t* order_out) {
    if (n <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_morton_order: bad arguments");
    return mi_icp_spatial_order(c, xyz, n, order_out, MI_ICP_HOST);
}

}  // extern "C"
