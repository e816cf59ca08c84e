// mi_icp.hip -- libmi_icp.so's core: context life cycle, the correspondence search, the reduction into the 6x6
// system and the device-resident registration loop behind the C ABI declared in include/mi_icp.h.  gfx950 only.
//
// The loop mirrors registration::RegistrationICP (registration/registration.cu:121-172) but keeps everything
// device-resident: one nearest-neighbour launch + one reduction launch (whose last block also takes the loop's
// step) per iteration; the host only enqueues iterations and looks at a `done` flag between chunks.  Nothing is
// allocated inside the loop.  (The other translation units: csrc/ctx.h.)
#include "ctx.h"
#include "fused_small.h"
#include "loop_step_kernel.h"
#include "lzf.h"
#include "nn_search.h"
#include "reduce.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

namespace mi {
namespace eng {

// elements per thread that travel together in reduce_pt2pl_kernel: 2 from kPt2PlTwoFrom source points up (round 6:
// profiles/r06_reduce_shape_sweep.txt -- the step 2 % shorter at 10M and 5M points on 512 blocks), 4 below (an eighth of the
// bench's source takes 16 elements per thread on ~300 blocks: in pairs that is eight dependent round trips instead of four)
constexpr int64_t kPt2PlTwoFrom = (int64_t)4 << 20;

// sources of at least this many points make their own seeds for a first pass (tuning knob MI_ICP_COARSE_MIN)
static int64_t coarse_first_min() {
    static const int64_t v = [] { const char* s = std::getenv("MI_ICP_COARSE_MIN"); return s ? std::atoll(s) : (int64_t)1 << 16; }();
    return v;
}

// The target has kd cells and its groups' planes: a query's own leaf is a binary descent away (nn_search.h
// locate_by_planes).  MI_ICP_NO_LOCATE_PLANES: A/B switch -- the greedy record descent (locate_leaves), no re-location.
bool planes_available(const mi_icp_ctx* c) {
    static const bool off = std::getenv("MI_ICP_NO_LOCATE_PLANES") != nullptr;
    return !off && c->cell_levels >= 0 && c->gplanes.p != nullptr && c->cell_planes.p != nullptr && c->cell_gstart.p != nullptr &&
           c->leaf_first > 1u && c->nleaf > 0;
}

// seeds for every source point under T (or the loop's transform) into nn_idx; gated: only if the loop's last step asks
int launch_locate_by_planes(mi_icp_ctx* c, const Xform& X, const DevLoop* loop, int gated) {
    const int grid = (int)std::min<int64_t>(blocks_for(c->ns), 8192);
    locate_by_planes<<<grid, 256, 0, c->stream>>>((const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, (int)c->ns,
                                                 (const float2*)c->cell_planes.p, c->cell_levels, (const uint32_t*)c->cell_gstart.p,
                                                 (const float2*)c->gplanes.p, (uint32_t)c->nleaf, X, loop, gated, (int32_t*)c->nn_idx.p);
    KCHK(c);
    return MI_ICP_OK;
}

int launch_nn(mi_icp_ctx* c, const Mat4& T, float r2, bool seed, unsigned long long* stats, const DevLoop* loop) {
    if (c->ns <= 0) return MI_ICP_OK;
    int32_t* idx = (int32_t*)c->nn_idx.p;
    // (inside the registration loop the distances are not stored: nothing reads them there, and every
    // entry point that hands distances out runs its own search first)
    float* d2 = (float*)c->nn_d2.p;
    if (c->nt <= 0) {
        fill_i32<<<blocks_for(c->ns), 256, 0, c->stream>>>(idx, c->ns, -1);
        KCHK(c);
        c->nn_valid = true;
        c->n_user_pairs = -1;
        return MI_ICP_OK;
    }
    const Xform X = make_xform(T);
    const bool use_seed = seed && c->nn_valid;
    // Inside a registration loop the halos are used if they are there and asked for if they are not (loop_run
    // decides about building them); a one-shot seeded search builds them on the spot.
    static const bool always_wait = std::getenv("MI_ICP_WAIT_LINKS") != nullptr;  // A/B switch (soak tests)
    if (!loop && use_seed) TRY(ensure_links(c));
    if (loop && always_wait) {
        TRY(start_links_async(c));
        TRY(ensure_links(c));
        c->halo_use = halo_poll(c);
    }
    const bool have_halo = loop ? c->halo_use : halo_poll(c);
    EvTimer t(c, 0, loop != nullptr);
    const float* links = have_halo ? (const float*)c->thalo.p : nullptr;
    uint32_t* want = (loop && !have_halo && !c->links_inflight && c->links_allowed) ? (uint32_t*)c->halo_want.p : nullptr;
    bool self_seeded = false;
    static const bool no_coarse = std::getenv("MI_ICP_NO_COARSE_FIRST") != nullptr;  // A/B switch
    auto launch = [&](bool seeded, const float* sx, const float* sy, const float* sz, int64_t ns, int32_t* out_idx,
                      float* out_d2) {
        const uint32_t npackets = (uint32_t)((ns + 63) / 64);
        const uint32_t nblocks = (npackets + kNNPacketsPerBlock - 1) / kNNPacketsPerBlock;
        const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
#define MI_NN_ARGS sx, sy, sz, (int)ns, (const float*)c->nodes.p, (const float*)c->tblk.p, (const float*)lreg_of(c), \
                   links, c->leaf_first, X, loop, r2, nblocks, out_idx, out_d2, stats, want
        if (stats) {
            if (seeded) nn_packet_kernel<true, true><<<grid, kNNThreads, 0, c->stream>>>(MI_NN_ARGS);
            else nn_packet_kernel<false, true><<<grid, kNNThreads, 0, c->stream>>>(MI_NN_ARGS);
        } else if (seeded && loop && c->stamps_on) {  // (mi_icp_debug_set_step_stamps: the same kernel + two stamps per wave)
            nn_packet_kernel<true, false, true><<<grid, kNNThreads, 0, c->stream>>>(MI_NN_ARGS);
        } else {
            if (seeded) nn_packet_kernel<true, false><<<grid, kNNThreads, 0, c->stream>>>(MI_NN_ARGS);
            else nn_packet_kernel<false, false><<<grid, kNNThreads, 0, c->stream>>>(MI_NN_ARGS);
        }
#undef MI_NN_ARGS
    };
    // No previous matches, but the target's halos are there: every query takes the leaf a greedy descent lands in
    // as its seed (nn_search.h: locate_leaves) and the seeded search does the rest.  (Without halos every lane
    // whose seed leaf's region does not finish it walks up from there -- under the displacement a registration
    // starts with that is most packets, and costs more than the walk from the root: 10M points 3.9 against 1.2 ms.)
    if (!use_seed && !stats && !no_coarse && c->leaf_first > 1u && c->ns >= coarse_first_min() && have_halo) {
        if (planes_available(c)) {
            TRY(launch_locate_by_planes(c, X, loop, 0));
        } else {
            locate_leaves<<<blocks_for(c->ns), 256, 0, c->stream>>>((const float*)c->sx.p, (const float*)c->sy.p,
                                                                    (const float*)c->sz.p, (int)c->ns,
                                                                    (const float*)c->nodes.p, c->leaf_first,
                                                                    (uint32_t)c->nleaf, X, loop, idx);
            KCHK(c);
        }
        self_seeded = true;
    }
    c->last_search_kind = use_seed ? 1 : (self_seeded ? 2 : 0);
    launch(use_seed || self_seeded, (const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, c->ns, idx,
           loop ? nullptr : d2);
    KCHK(c);
    c->nn_valid = true;
    c->n_user_pairs = -1;
    return MI_ICP_OK;
}

int occupancy_loop(int which) {
    int blocks = -1;
    hipError_t e = hipErrorInvalidValue;
    if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, nn_packet_kernel<true, false>, kNNThreads, 0);
    else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, nn_packet_kernel<false, false>, kNNThreads, 0);
    else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reduce_pt2pl_kernel<2, 1>, kReduceThreads, 0);
    else return -1;
    return e == hipSuccess ? blocks : -2;
}

}  // namespace eng
}  // namespace mi

namespace {

int ensure_inverse_maps(mi_icp_ctx* c) {
    if (!c->inv_s_valid && c->ns > 0) {
        int32_t* inv;
        TRY(ensure(c, c->inv_s, (size_t)c->ns, &inv));
        invert_perm_source<<<blocks_for(c->ns), 256, 0, c->stream>>>((const int32_t*)c->sperm.p, (int)c->ns, inv);
        KCHK(c);
        c->inv_s_valid = true;
    }
    if (!c->inv_t_valid && c->nt > 0) {
        int32_t* inv;
        TRY(ensure(c, c->inv_t, (size_t)c->nt, &inv));
        HIPCHK(c, hipMemsetAsync(inv, 0xff, sizeof(int32_t) * (size_t)c->nt, c->stream));
        invert_perm_target<<<blocks_for(c->nts), 256, 0, c->stream>>>((const int32_t*)c->tidx.p, (int)c->nts, inv);
        KCHK(c);
        c->inv_t_valid = true;
    }
    return MI_ICP_OK;
}

template <int EST, int MODE>
void launch_reduce_t(mi_icp_ctx* c, const ReduceArgs& a, const Xform& X, const DevLoop* loop, int grid,
                     double* partial, uint32_t* ticket, double* out) {
    reduce_kernel<EST, MODE><<<grid, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, out);
}

// elements per thread below which the reduction uses fewer than its 1024 blocks: 16 measured best on
// 1.25M-5M point shards (fewer partials for the finishing block)
constexpr int kReduceElemsPerThread = 16;


bool known_estimator(int est) {
    return est == kEstP2P || est == kEstPt2Pl || est == kEstSym || est == kEstColored || est == kEstGICP;
}

bool estimator_ready(const mi_icp_ctx* c, int est) {
    switch (est) {
        case kEstP2P: return true;
        case kEstPt2Pl: return c->t_has_nrm;                  // transformation_estimation.cu:199-200
        case kEstSym: return c->t_has_nrm && c->s_has_nrm;    // :293-294
        case kEstGICP: return c->t_has_cov && c->s_has_cov;   // generalized_icp.cu:156-159
        case kEstColored:                                     // colored_icp.cu:222-224
            return c->t_has_nrm && c->t_has_int && c->t_has_grad && c->s_has_int;
        default: return false;
    }
}

// Accumulate sys[32] on the device (into c->sys_dev) for the current correspondences.
// When the estimator's inputs are missing only the statistics ([28], [29]) are formed.
// fuse_step: the loop's step may ride in the reduction's finishing block (reduce.h STEP; single GPU only);
// *stepped tells whether it did
int launch_reduce(mi_icp_ctx* c, int est, int mode, const Mat4& T, DevLoop* loop = nullptr, bool fuse_step = false,
                  bool* stepped = nullptr) {
    if (stepped) *stepped = false;
    double *partial, *sys;
    uint32_t* ticket;
    TRY(ensure(c, c->partial, (size_t)kReduceBlocks * kSysSize, &partial));
    TRY(ensure(c, c->sys_dev, kSysSize, &sys));
    if (!c->ticket.p) {
        TRY(ensure(c, c->ticket, 64, &ticket));
        HIPCHK(c, hipMemsetAsync(ticket, 0, 256, c->stream));
    }
    ticket = (uint32_t*)c->ticket.p;
    ReduceArgs a;
    a.sx = (const float*)c->sx.p;
    a.sy = (const float*)c->sy.p;
    a.sz = (const float*)c->sz.p;
    a.snrm = (const float4*)c->snrm.p;
    a.scov = (const float*)c->scov.p;
    a.tblk = (const float*)c->tblk.p;
    a.tnrm = (const float4*)c->tnrm.p;
    a.trec = c->t_has_rec ? (const float*)c->trec.p : nullptr;
    a.tcov = (const float*)c->tcov.p;
    a.tgrad = (const float4*)c->tgrad.p;
    a.sint = (const float*)c->sint.p;
    a.sqrt_lambda_geometric = std::sqrt(c->lambda_geometric);
    a.sqrt_lambda_photometric = std::sqrt(1.0f - c->lambda_geometric);
    a.nn_idx = (const int32_t*)c->nn_idx.p;
    a.pairs = nullptr;
    a.inv_s = a.inv_t = nullptr;
    a.ns = (int)c->ns;
    a.nt = (int)c->nt;
    a.count = c->ns;
    if (c->n_user_pairs >= 0) {
        TRY(ensure_inverse_maps(c));
        a.pairs = (const int32_t*)c->user_pairs.p;
        a.inv_s = (const int32_t*)c->inv_s.p;
        a.inv_t = (const int32_t*)c->inv_t.p;
        a.count = c->n_user_pairs;
    }
    if (c->ns <= 0 || c->nt <= 0 || (!a.pairs && !c->nn_valid)) a.count = 0;
    // >= 16 elements per thread up to 1024 blocks: enough blocks to hide the gather latency,
    // few enough partials for the finishing block
    // ... but at least one block per CU while there is one element per thread to give it
    const int64_t wide = std::min<int64_t>(256, blocks_for(a.count, kReduceThreads));
    const int grid = (int)std::max<int64_t>(
            wide, std::min<int64_t>(kReduceBlocks, blocks_for(a.count, kReduceThreads * kReduceElemsPerThread)));
    const Xform X = make_xform(T);
    if (!estimator_ready(c, est)) {
        est = kEstP2P;
        mode = 1;
    }
    static const bool no_fast_reduce = std::getenv("MI_ICP_NO_FAST_REDUCE") != nullptr;  // A/B switch
    if (est == kEstPt2Pl && mode == 0 && !a.pairs && a.trec && a.count > 0 && !no_fast_reduce) {
        // four elements in flight per thread; at most 512 blocks (2 per CU): measured best on the 10M bench
        // (256 / 512 / 1024 / 2048 blocks: 0.090 / 0.079 / 0.080 / 0.091 ms; 6 or 8 elements in flight on 512,
        // 768 or 1024 blocks: 0.078 - 0.084 ms -- the kernel sits at ~5.1 TB/s of the ~6.3 a pure stream reaches)
        const int g2 = std::min(grid, 512);
        EvTimer t(c, 1, loop != nullptr);
        const MailArgs no_mail = {nullptr, nullptr, 0, 1, 0u, nullptr, nullptr};
        const bool mail = mail_on(c);
        const bool two = a.count >= kPt2PlTwoFrom;
#define MI_PT2PL(STEP, STAMP, MAILARGS)                                                                                       \
    do {                                                                                                                     \
        if (two) reduce_pt2pl_kernel<2, STEP, STAMP><<<g2, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, sys, MAILARGS); \
        else reduce_pt2pl_kernel<4, STEP, STAMP><<<g2, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, sys, MAILARGS);     \
    } while (0)
        if (fuse_step && loop && mail) {  // N ranks on one node: exchange + step in the finishing block
            if (c->stamps_on) MI_PT2PL(2, true, mail_args(c));
            else MI_PT2PL(2, false, mail_args(c));
            if (stepped) *stepped = true;
        } else if (fuse_step && loop && !c->comm && !c->mail_dev) {
            if (c->stamps_on) MI_PT2PL(1, true, no_mail);
            else MI_PT2PL(1, false, no_mail);
            if (stepped) *stepped = true;
        } else {
            MI_PT2PL(0, false, no_mail);
        }
#undef MI_PT2PL
        KCHK(c);
        return MI_ICP_OK;
    }
    {
        EvTimer t(c, 1, loop != nullptr);
        switch (est * 2 + mode) {
            case kEstP2P * 2 + 0: launch_reduce_t<kEstP2P, 0>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstP2P * 2 + 1: launch_reduce_t<kEstP2P, 1>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstPt2Pl * 2 + 0: launch_reduce_t<kEstPt2Pl, 0>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstPt2Pl * 2 + 1: launch_reduce_t<kEstPt2Pl, 1>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstSym * 2 + 0: launch_reduce_t<kEstSym, 0>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstSym * 2 + 1: launch_reduce_t<kEstSym, 1>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstColored * 2 + 0: launch_reduce_t<kEstColored, 0>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstColored * 2 + 1: launch_reduce_t<kEstColored, 1>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstGICP * 2 + 0: launch_reduce_t<kEstGICP, 0>(c, a, X, loop, grid, partial, ticket, sys); break;
            case kEstGICP * 2 + 1: launch_reduce_t<kEstGICP, 1>(c, a, X, loop, grid, partial, ticket, sys); break;
            default: return fail(c, MI_ICP_ERR_INVALID, "unknown estimation type %d", est);
        }
        KCHK(c);
    }
    return MI_ICP_OK;
}

// all-reduce across ranks (if any), copy to the host, synchronise
int fetch_system(mi_icp_ctx* c, double* out) {
    double* sys = (double*)c->sys_dev.p;
    TRY(allreduce_system(c));
    HIPCHK(c, hipMemcpyAsync(c->sys_host, sys, kSysSize * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    const bool mail = mail_on(c);
    int32_t* err_host = reinterpret_cast<int32_t*>(c->sys_host + 40);  // (spare words of the pinned buffer)
    if (mail) HIPCHK(c, hipMemcpyAsync(err_host, (const int32_t*)c->mail_state.p + 1, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    if (mail && *err_host) return comm_failed(c, "the ranks' exchange timed out (mailbox): a peer did not post its sums");
    std::memcpy(out, c->sys_host, kSysSize * sizeof(double));
    return MI_ICP_OK;
}

// host step of ComputeTransformation for the built-in estimators (one-shot entry points)
Mat4 solve_update(const mi_icp_ctx* c, int est, const double* sys, float det_thresh) {
    const int64_t n_model = c->ns_global > 0 ? c->ns_global : c->ns;
    return mi::solve_update(est, estimator_ready(c, est), sys, det_thresh, n_model);
}

void stats_from_system(const mi_icp_ctx* c, const double* sys, float* fitness, float* rmse) {
    mi::stats_from_system(sys, c->ns_global > 0 ? c->ns_global : c->ns, fitness, rmse);
}

}  // namespace

extern "C" {

const char* mi_icp_version(void) { return "mi_icp 0.1 (gfx950)"; }

int mi_icp_create(int device, mi_icp_ctx** out) {
    if (!out) return MI_ICP_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return MI_ICP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MI_ICP_ERR_NO_DEVICE;
    mi_icp_ctx* c = new mi_icp_ctx();
    c->device = device;
    bool ok = hipHostMalloc((void**)&c->sys_host, 64 * sizeof(double), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&c->f_host, 64 * sizeof(float), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&c->u_host, (16 + kWantSlots) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->loop_host, sizeof(DevLoop), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    // (lowest priority: the halo build fills what the context's own stream leaves idle)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    ok = ok && hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_least) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_links, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
        for (int i = 0; i < mi_icp_ctx::kEvPairs && ok; ++i)
            ok = hipEventCreate(&c->evp[k][i][0]) == hipSuccess && hipEventCreate(&c->evp[k][i][1]) == hipSuccess;
    if (!ok) {
        mi_icp_destroy(c);
        return MI_ICP_ERR_HIP;
    }
    *out = c;
    return MI_ICP_OK;
}

void mi_icp_destroy(mi_icp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->aux) mi_icp_destroy(c->aux);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_links) (void)hipEventDestroy(c->ev_links);
    comm_release(c);
    DevBuf* all[] = {&c->trec, &c->tidx, &c->thalo, &c->tlinks_tmp, &c->halo_want, &c->loop_hist, &c->tblk, &c->tnrm, &c->tcov, &c->tgrad, &c->sint, &c->nodes, &c->inv_t, &c->cell_planes, &c->cell_samples, &c->cell_cstart,
                     &c->cell_gstart, &c->sx, &c->sy, &c->sz,
                     &c->sperm, &c->snrm, &c->scov, &c->nn_idx, &c->nn_d2, &c->inv_s,
                     &c->user_pairs, &c->keys0, &c->keys1, &c->vals0, &c->vals1, &c->hist,
                     &c->scan_tmp, &c->bounds_part, &c->bounds, &c->partial, &c->sys_dev,
                     &c->dense_idx, &c->flags, &c->pairs_out, &c->seg_start, &c->loop_dev, &c->ticket, &c->mail_state, &c->alt[0],
                     &c->alt[1], &c->alt[2], &c->alt[3], &c->alt[4], &c->alt[5], &c->alt[6], &c->alt[7], &c->alt[8], &c->stage[0],
                     &c->stage[1], &c->stage[2], &c->stage[3], &c->stage[4], &c->stage[5], &c->knn_idx, &c->tscale, &c->vpay[0], &c->vpay[1],
                     &c->vpay[2], &c->vpay[3], &c->vpay[4], &c->vpay[5], &c->stamps, &c->knn_flags, &c->gplanes, &c->src_bounds, &c->cell_boxes, &c->cell_hist, &c->vx_tab};
    for (DevBuf* b : all) release(*b);
    if (c->sys_host) (void)hipHostFree(c->sys_host);
    if (c->cell_total_host) (void)hipHostFree(c->cell_total_host);
    if (c->f_host) (void)hipHostFree(c->f_host);
    if (c->u_host) (void)hipHostFree(c->u_host);
    if (c->od_host) (void)hipHostFree(c->od_host);
    if (c->loop_host) (void)hipHostFree(c->loop_host);
    if (c->hist_host) (void)hipHostFree(c->hist_host);
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < mi_icp_ctx::kEvPairs; ++i)
            for (int e = 0; e < 2; ++e)
                if (c->evp[k][i][e]) (void)hipEventDestroy(c->evp[k][i][e]);
    for (int i = 0; i < 4; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete c;
}

const char* mi_icp_last_error(const mi_icp_ctx* c) { return c ? c->err.c_str() : "null context"; }

int mi_icp_set_iteration_callback(mi_icp_ctx* c, mi_icp_iteration_fn fn, void* user) {
    if (!c) return MI_ICP_ERR_INVALID;
    c->iter_fn = fn;
    c->iter_user = user;
    return MI_ICP_OK;
}

int mi_icp_set_stream(mi_icp_ctx* c, void* hip_stream) {
    if (c && c->aux) c->aux->stream = (hipStream_t)hip_stream;
    TRY(check_ctx(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;
    return MI_ICP_OK;
}

int mi_icp_synchronize(mi_icp_ctx* c) {
    TRY(check_ctx(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    return MI_ICP_OK;
}

int mi_icp_set_profiling(mi_icp_ctx* c, int enable) {
    TRY(check_ctx(c));
    c->profiling = enable != 0;
    for (double& v : c->prof) v = 0.0;
    c->ev_pending_nn = c->ev_pending_red = false;
    return MI_ICP_OK;
}

int mi_icp_get_profile(mi_icp_ctx* c, double* out8) {
    if (!c || !out8) return MI_ICP_ERR_INVALID;
    std::memcpy(out8, c->prof, sizeof(c->prof));
    return MI_ICP_OK;
}

int mi_icp_set_global_source_count(mi_icp_ctx* c, int64_t n_total) {
    if (!c || n_total < 0) return MI_ICP_ERR_INVALID;
    c->ns_global = n_total;
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
static int export_dense_idx(mi_icp_ctx* c, int32_t** dense_out) {
    int32_t* dense;
    TRY(ensure(c, c->dense_idx, (size_t)std::max<int64_t>(c->ns, 1), &dense));
    if (c->ns > 0) {
        export_dense<<<blocks_for(c->ns), 256, 0, c->stream>>>(
                (const int32_t*)c->nn_idx.p, (const float*)c->nn_d2.p, (const int32_t*)c->sperm.p,
                (const int32_t*)c->tidx.p, (int)c->ns, dense, nullptr);
        KCHK(c);
    }
    *dense_out = dense;
    return MI_ICP_OK;
}

int mi_icp_search_radius_1nn(mi_icp_ctx* c, const float* T, float radius, int32_t* idx_out,
                             float* d2_out, int mem_kind, double* stats) {
    TRY(check_ctx(c));
    if (c->ns <= 0) return fail(c, MI_ICP_ERR_STATE, "search: no source set");
    const Mat4 M = load_T(T);
    const float r2 = radius * radius;  // kdtree_flann.inl:119-120
    TRY(launch_nn(c, M, r2, true));  // seeded when a previous search exists (same result, see evaluate_registration)
    if (c->nt <= 0) {
        float* d2 = (float*)c->nn_d2.p;
        fill_i32<<<blocks_for(c->ns), 256, 0, c->stream>>>((int32_t*)d2, c->ns, 0x7f800000);
        KCHK(c);
    }
    if (idx_out || d2_out) {
        int32_t* dense;
        float* dense_d2 = nullptr;
        TRY(ensure(c, c->dense_idx, (size_t)c->ns, &dense));
        if (d2_out) TRY(ensure(c, c->flags, (size_t)c->ns, (float**)&dense_d2));
        export_dense<<<blocks_for(c->ns), 256, 0, c->stream>>>(
                (const int32_t*)c->nn_idx.p, (const float*)c->nn_d2.p, (const int32_t*)c->sperm.p,
                (const int32_t*)c->tidx.p, (int)c->ns, dense, dense_d2);
        KCHK(c);
        TRY(from_device(c, dense, idx_out, (size_t)c->ns, mem_kind));
        if (d2_out) TRY(from_device(c, dense_d2, d2_out, (size_t)c->ns, mem_kind));
    }
    if (stats) {
        double sys[kSysSize];
        TRY(launch_reduce(c, kEstP2P, 1, M));
        TRY(fetch_system(c, sys));
        stats[0] = sys[29];
        stats[1] = sys[28];
        stats[2] = (double)(c->ns_global > 0 ? c->ns_global : c->ns);
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        collect_events(c);
    }
    return MI_ICP_OK;
}

int mi_icp_get_correspondences(mi_icp_ctx* c, int32_t* pairs, int64_t capacity, int64_t* count,
                               int mem_kind) {
    TRY(check_ctx(c));
    if (!count) return fail(c, MI_ICP_ERR_INVALID, "get_correspondences: count is null");
    *count = 0;
    if (c->n_user_pairs >= 0) {
        *count = c->n_user_pairs;
        if (pairs && capacity >= c->n_user_pairs)
            TRY(from_device(c, (const int32_t*)c->user_pairs.p, pairs, (size_t)c->n_user_pairs * 2, mem_kind));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MI_ICP_OK;
    }
    if (!c->nn_valid || c->ns <= 0 || c->nt <= 0) return MI_ICP_OK;
    int32_t* dense;
    TRY(export_dense_idx(c, &dense));
    uint32_t *flags, *tmp;
    int32_t* out;
    TRY(ensure(c, c->flags, (size_t)c->ns, &flags));
    TRY(ensure(c, c->scan_tmp, (size_t)scan_num_tiles(c->ns) + 2, &tmp));
    TRY(ensure(c, c->pairs_out, (size_t)c->ns * 2, &out));
    corr_flags<<<blocks_for(c->ns), 256, 0, c->stream>>>(dense, (int)c->ns, flags);
    KCHK(c);
    exclusive_scan_u32(c->stream, flags, flags, c->ns, tmp);
    KCHK(c);
    corr_compact<<<blocks_for(c->ns), 256, 0, c->stream>>>(dense, flags, (int)c->ns, out);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(c->u_host, tmp + scan_num_tiles(c->ns), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t m = (int64_t)c->u_host[0];
    *count = m;
    if (pairs && capacity >= m && m > 0) {
        TRY(from_device(c, out, pairs, (size_t)m * 2, mem_kind));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MI_ICP_OK;
}

int mi_icp_set_correspondences(mi_icp_ctx* c, const int32_t* pairs, int64_t count, int mem_kind) {
    TRY(check_ctx(c));
    if (count < 0 || (count > 0 && !pairs)) return fail(c, MI_ICP_ERR_INVALID, "set_correspondences: bad arguments");
    int32_t* d;
    TRY(ensure(c, c->user_pairs, (size_t)std::max<int64_t>(count, 1) * 2, &d));
    if (count > 0)
        HIPCHK(c, hipMemcpyAsync(d, pairs, (size_t)count * 2 * sizeof(int32_t),
                                 mem_kind == MI_ICP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                 c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller may free `pairs` on return
    c->n_user_pairs = count;
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
int mi_icp_compute_system(mi_icp_ctx* c, int est, const float* T, double* out32) {
    TRY(check_ctx(c));
    if (!out32) return fail(c, MI_ICP_ERR_INVALID, "compute_system: out is null");
    if (!known_estimator(est))
        return fail(c, MI_ICP_ERR_INVALID, "unknown estimation type %d", est);
    if (!estimator_ready(c, est))
        return fail(c, MI_ICP_ERR_STATE, "estimation type %d needs normals/covariances that were not set", est);
    const Mat4 M = load_T(T);
    TRY(launch_reduce(c, est, 0, M));
    return fetch_system(c, out32);
}

int mi_icp_compute_transformation(mi_icp_ctx* c, int est, const float* T, float det_thresh,
                                  float* update16) {
    TRY(check_ctx(c));
    if (!update16) return fail(c, MI_ICP_ERR_INVALID, "compute_transformation: out is null");
    if (!known_estimator(est))
        return fail(c, MI_ICP_ERR_INVALID, "unknown estimation type %d", est);
    const Mat4 M = load_T(T);
    double sys[kSysSize];
    TRY(launch_reduce(c, est, 0, M));
    TRY(fetch_system(c, sys));
    const Mat4 u = solve_update(c, est, sys, det_thresh);
    std::memcpy(update16, u.data(), sizeof(float) * 16);
    return MI_ICP_OK;
}

int mi_icp_compute_rmse(mi_icp_ctx* c, int est, const float* T, float* rmse) {
    TRY(check_ctx(c));
    if (!rmse) return fail(c, MI_ICP_ERR_INVALID, "compute_rmse: out is null");
    if (!known_estimator(est))
        return fail(c, MI_ICP_ERR_INVALID, "unknown estimation type %d", est);
    *rmse = 0.0f;
    if (!estimator_ready(c, est)) return MI_ICP_OK;  // the reference returns 0.0
    const Mat4 M = load_T(T);
    double sys[kSysSize];
    TRY(launch_reduce(c, est, 1, M));
    TRY(fetch_system(c, sys));
    if (est == kEstColored) {
        *rmse = (float)sys[27];  // the reference returns the plain sum (colored_icp.cu:302-306)
    } else if (sys[29] > 0.0) {
        *rmse = std::sqrt((float)sys[27] / (float)sys[29]);
    }
    return MI_ICP_OK;
}

int mi_icp_solve_system(const double* sys32, float det_thresh, float* T16) {
    if (!sys32 || !T16) return MI_ICP_ERR_INVALID;
    Mat4 T;
    const bool ok = host::solve_system(sys32, det_thresh, T);
    std::memcpy(T16, T.data(), sizeof(float) * 16);
    return ok ? 1 : 0;
}

int64_t mi_icp_lzf_decompress(const void* in, int64_t in_len, void* out, int64_t out_capacity) {
    if (!in || !out || in_len < 0 || out_capacity < 0) return 0;
    return (int64_t)lzf::decompress((const uint8_t*)in, (size_t)in_len, (uint8_t*)out, (size_t)out_capacity);
}

int64_t mi_icp_lzf_compress(const void* in, int64_t in_len, void* out, int64_t out_capacity) {
    if (!in || !out || in_len < 0 || out_capacity < 0) return 0;
    return (int64_t)lzf::compress((const uint8_t*)in, (size_t)in_len, (uint8_t*)out, (size_t)out_capacity);
}

int mi_icp_kabsch_from_sums(const double* sys32, int64_t n_model, float* T16) {
    if (!sys32 || !T16 || n_model <= 0) return MI_ICP_ERR_INVALID;
    const Mat4 T = host::kabsch_from_sums(sys32, (long long)n_model);
    std::memcpy(T16, T.data(), sizeof(float) * 16);
    return MI_ICP_OK;
}

void mi_icp_vector6_to_matrix4(const float* x6, float* T16) {
    const Mat4 T = host::vector6_to_matrix4(x6);
    std::memcpy(T16, T.data(), sizeof(float) * 16);
}

// ---------------------------------------------------------------------------
int mi_icp_evaluate_registration(mi_icp_ctx* c, float max_distance, const float* T,
                                 mi_icp_result* out) {
    TRY(check_ctx(c));
    if (!out) return fail(c, MI_ICP_ERR_INVALID, "evaluate_registration: out is null");
    const Mat4 M = load_T(T);
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->transformation, M.data(), sizeof(float) * 16);
    // (a rank of a sharded job goes through the motions even with an empty shard: its peers wait for its sums)
    if (max_distance <= 0.0f || (c->ns <= 0 && c->nranks <= 1)) {  // registration.cu:40-42
        c->nn_valid = false;
        return MI_ICP_OK;
    }
    // registration.cu:114-116: the source is moved only when T is not (approximately) identity
    const Mat4 apply = host::is_identity4(M) ? host::identity4() : M;
    double sys[kSysSize];
    // seeded by the previous search of the same clouds when there is one: the result is the
    // same exact nearest neighbour (equal distances resolve to the lowest slot either way)
    TRY(launch_nn(c, apply, max_distance * max_distance, true));
    TRY(launch_reduce(c, kEstP2P, 1, apply));
    TRY(fetch_system(c, sys));
    stats_from_system(c, sys, &out->fitness, &out->inlier_rmse);
    out->n_correspondences = (int64_t)sys[29];
    out->nn_passes = 1;
    return MI_ICP_OK;
}


static void fill_result(const mi_icp_ctx* c, mi_icp_result* out) {
    const DevLoop& L = *c->loop_host;
    std::memcpy(out->transformation, L.T.data(), sizeof(float) * 16);
    out->fitness = L.fitness;
    out->inlier_rmse = L.rmse;
    out->n_correspondences = (int64_t)L.sys[29];
    out->iterations = L.iterations;
    out->nn_passes = L.passes;
}

static int loop_pull(mi_icp_ctx* c) {  // device state -> pinned mirror, synchronises
    HIPCHK(c, hipMemcpyAsync(c->loop_host, c->loop_dev.p, sizeof(DevLoop), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->loop_host->error) return comm_failed(c, "the ranks' exchange timed out (mailbox): a peer did not post its sums");
    if (c->iter_fn && c->loop_host->history != 0ull && c->loop_host->iterations > c->iter_reported) {
        // the iterations started since the last look, in order (a ring: at most kLoopHistory of them per look)
        const int upto = c->loop_host->iterations;
        const int from = std::max(c->iter_reported, upto - kLoopHistory);
        HIPCHK(c, hipMemcpyAsync(c->hist_host, c->loop_hist.p, sizeof(float) * 2 * kLoopHistory, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->iter_reported = upto;
        for (int i = from; i < upto; ++i) {
            const float* e = c->hist_host + 2 * (size_t)(i & (kLoopHistory - 1));
            c->iter_fn(c->iter_user, i, e[0], e[1]);
        }
    }
    return MI_ICP_OK;
}

// Small clouds, point-to-plane, one GPU: the whole evaluation -- seeded search, the system's rows, their
// reduction, the step -- is ONE launch (fused_small.h).  Measured, the loop of a 30-iteration call, one launch /
// two launches per iteration: 20k points 0.47 / 0.55 ms, 80k 0.61, 112k 0.70, 150k 0.78 / 0.81, 200k 0.89 / 0.84,
// 307k 1.18 / 0.91 -- past ~170k points the per-packet totals (one set of 30 sums per 64 points instead of
// one per 4096) cost more than the second launch; MI_ICP_FUSED_MAX moves the limit.
constexpr int64_t kFusedMax = 170000;
// (point-to-point: its rows are cheaper to form than to total -- on converged clean clouds the one launch wins by 10 % at
// 50k points, 5 % at 114k and loses 6 % at 170k, 20 % at 250k against search + reduction + step:
// profiles/r05_p2p_one_launch_by_size.txt)
constexpr int64_t kFusedMaxP2P = 135000;
static bool fused_iteration_applies(const mi_icp_ctx* c, bool seed) {
    static const bool off = std::getenv("MI_ICP_NO_FUSED_ITERATION") != nullptr;  // A/B switch
    static const int64_t forced = [] { const char* e = std::getenv("MI_ICP_FUSED_MAX"); return e ? std::atoll(e) : (int64_t)-1; }();
    const bool pt2pl = c->loop_est == kEstPt2Pl && estimator_ready(c, kEstPt2Pl) && c->t_has_rec && c->trec.p != nullptr;
    const bool p2p = c->loop_est == kEstP2P;
    const int64_t limit = forced >= 0 ? forced : (p2p ? kFusedMaxP2P : kFusedMax);
    return !off && seed && c->nn_valid && (pt2pl || p2p) && !c->comm && !c->mail_dev &&
           c->n_user_pairs < 0 && c->ns > 0 && c->ns <= limit && c->nt > 0;
}

static int launch_fused_iteration(mi_icp_ctx* c, DevLoop* d) {
    static const bool always_wait = std::getenv("MI_ICP_WAIT_LINKS") != nullptr;  // A/B switch (soak tests)
    if (always_wait) {
        TRY(start_links_async(c));
        TRY(ensure_links(c));
        c->halo_use = halo_poll(c);
    }
    const bool have_halo = c->halo_use;
    uint32_t* want = (!have_halo && !c->links_inflight && c->links_allowed) ? (uint32_t*)c->halo_want.p : nullptr;
    const uint32_t npackets = (uint32_t)((c->ns + 63) / 64);
    const uint32_t nblocks = (npackets + kFusedPackets - 1) / kFusedPackets;
    const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
    double *partial, *sys;
    TRY(ensure(c, c->partial, (size_t)std::max<uint32_t>(kReduceBlocks, grid) * kSysSize, &partial));
    TRY(ensure(c, c->sys_dev, kSysSize, &sys));
    if (!c->ticket.p) {
        uint32_t* ticket;
        TRY(ensure(c, c->ticket, 64, &ticket));
        HIPCHK(c, hipMemsetAsync(ticket, 0, 256, c->stream));
    }
    EvTimer t(c, 0, true);
#define MI_FUSED_ARGS (const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, (int)c->ns, (const float*)c->nodes.p, \
            (const float*)c->tblk.p, (const float*)lreg_of(c), have_halo ? (const float*)c->thalo.p : nullptr, c->leaf_first, \
            c->loop_r2, npackets, nblocks, (int32_t*)c->nn_idx.p, want, (const float*)c->trec.p, d, partial, (uint32_t*)c->ticket.p, sys
    if (c->loop_est == kEstP2P) icp_small_iteration_kernel<kEstP2P><<<grid, kReduceThreads, 0, c->stream>>>(MI_FUSED_ARGS);
    else icp_small_iteration_kernel<kEstPt2Pl><<<grid, kReduceThreads, 0, c->stream>>>(MI_FUSED_ARGS);
#undef MI_FUSED_ARGS
    KCHK(c);
    c->last_search_kind = 1;
    return MI_ICP_OK;
}

// one evaluation: search under the loop's transform, reduction, all-reduce, step kernel
static int loop_enqueue_evaluation(mi_icp_ctx* c, bool seed) {
    DevLoop* d = (DevLoop*)c->loop_dev.p;
    // RE-LOCATION (loop.h): while this loop's steps are still large the seeded search is preceded by a launch that
    // does nothing unless the step just taken moved the source by more than about a leaf's width -- then every seed is
    // replaced by the leaf the moved query falls into.  Armed per chunk by loop_run; needs the halos (a located seed
    // without them walks like a stale one: measured on the bench's cold call, whose second search -- the queries a few
    // thousandths of a spacing from their partners after the first step -- leaves 0.85 lanes per packet unfinished from
    // located seeds against 1.18 from the first pass's matches, 0.21 against 0.196 ms, and the descent costs 0.12).
    if (seed && c->relocate_armed && c->halo_use && c->nn_valid && c->n_user_pairs < 0 && c->ns > 0 && c->nt > 0) {
        const Xform none = {};
        TRY(launch_locate_by_planes(c, none, d, 1));
    }
    if (fused_iteration_applies(c, seed)) return launch_fused_iteration(c, d);
    const Mat4 I = host::identity4();
    TRY(launch_nn(c, I, c->loop_r2, seed, nullptr, d));
    bool stepped = false;
    TRY(launch_reduce(c, c->loop_est, 0, I, d, true, &stepped));
    if (stepped) return MI_ICP_OK;  // (point-to-plane: the reduction's last block exchanged the sums, if need be, and took the step)
    const bool mail = mail_on(c);
    if (!mail) TRY(allreduce_system(c));  // (with a mailbox the step kernel starts with the exchange)
    const MailArgs no_mail = {nullptr, nullptr, 0, 1, 0u, nullptr, nullptr};
    loop_step_kernel<<<1, kStepThreads, 0, c->stream>>>(d, (double*)c->sys_dev.p, 0, mail ? mail_args(c) : no_mail);
    KCHK(c);
    return MI_ICP_OK;
}

// Enqueue up to `budget` iterations in chunks, looking at `done` between chunks -- and, while the target has
// no halos, at how many lanes of the seeded searches asked for one.  The first seeded iteration of a
// registration is still displaced and asks whatever the data; the second one tells noise from convergence.  So a
// large source's first two seeded iterations are chunks of their own: if more than 40 % of the lanes ask in the
// first, or more than ~3 % still do in the second, the halos are built -- on the private stream, and the loop's
// stream waits for them: an iteration that walks instead costs a 10M-point loop half of what the build does.
// Small sources (a walk costs them little, a host synchronisation much) decide at their first regular chunk's end.
// Clean data leaves a few lanes in a few thousand asking (a converged query within rounding of a face of its
// match's region): their packets' one-record walks are ~4 % of a search -- not worth a build to one registration,
// worth it to a target that keeps being registered against: after kHaloLongRun iterations on the same target the
// build is started in the background and taken up whenever it is done.
// twice what the halos and their build's scratch take (~128 + ~56 bytes per slot) must be free on the device
static bool halo_memory_free(const mi_icp_ctx* c) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return free_b >= (size_t)c->nts * 368u;
}

static int loop_run(mi_icp_ctx* c, int budget) {
    constexpr int kChunk = 8;
    constexpr int64_t kLarge = 500000, kHaloLongRun = 40, kHaloVeryLongRun = 1000;
    int chunk = kChunk;
    while (budget > 0) {
        const bool no_halo = !c->links_ready && !c->links_inflight && c->links_allowed && c->nt > 0;
        const bool undecided = no_halo && !c->halo_declined;
        // (a build the loop's decision started -- or one started with the loop on a context that has asked
        // before, or behind a small target's tree: the stream waits for what is left of it rather than walk)
        if (c->links_inflight && !c->halo_declined) TRY(ensure_links(c));
        // (a short remainder rides along: one host synchronisation less than it would cost.  The chunks of one call grow
        // -- 8, 16, 32, 32 ...: a look at the loop is ~30 us of copies, synchronisation and relaunch, two iterations of a
        // 100k-point loop; a loop that has not converged within its first chunks is unlikely to in the next few, and an
        // iteration enqueued past the end costs ~3 us.  The sizes depend on the budget alone: every rank enqueues alike.)
        int n = (budget <= chunk + chunk / 2) ? budget : chunk;
        const bool grown = n == chunk;
        // (with several ranks the chunking must not depend on anything a rank sees alone: every rank has to
        // enqueue the same number of evaluations -- an in-library RCCL all-reduce is a host-side call per
        // evaluation, and a rank that stops at `done` after fewer of them would leave its peers' calls unmatched)
        const bool several_ranks = c->comm != nullptr || c->mail_dev != nullptr;
        if (undecided && c->ns >= kLarge && !several_ranks) n = 1;
        const int passes_before = c->loop_host->passes;
        c->halo_use = halo_poll(c);
        const bool carried = c->relocate_armed && c->halo_use;  // this chunk's iterations carry the gated re-location launches
        const int relocations_before = c->loop_host->relocations;
        for (int i = 0; i < n; ++i) TRY(loop_enqueue_evaluation(c, true));
        // (a loop that has declined keeps counting -- a target registered against for long may still earn its halos -- but
        // looks at the 4-KB counter only every eighth chunk: the copy is ~1 us per iteration of an 8-way shard's 36-us step)
        const bool look = no_halo && (!c->halo_declined || (++c->halo_chunks & 7) == 0);
        if (look) HIPCHK(c, hipMemcpyAsync(c->u_host + 16, c->halo_want.p, kWantSlots * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        TRY(loop_pull(c));
        const int executed = c->loop_host->passes - passes_before;
        collect_pooled(c, executed);
        // (a chunk that carried the launches and never needed one: the steps have become small, and they only shrink)
        if (carried && c->loop_host->relocations == relocations_before) c->relocate_armed = false;
        // ... and they may grow again (point-to-plane sliding, an escape from a plateau): the step keeps sizing itself on
        // the device whether or not the launches ride along, so a chunk without them that took a large step arms the next
        // (large sources only: a stale seed's climb costs a 10M-point search milliseconds, a 100k-point one less than the
        // eight 5-us launches an armed chunk carries -- the reference's own benchmark call, 113k points sliding along
        // themselves, 1.66 -> 1.71 ms with the re-arming at every size)
        else if (!c->relocate_armed && c->relocate_possible && c->ns >= kLarge && c->loop_host->relocations != relocations_before) c->relocate_armed = true;
        budget -= n;
        c->halo_iters += executed;
        c->halo_iters_unseen += executed;
        if (look) {
            const int64_t seen_iters = c->halo_iters_unseen;
            c->halo_iters_unseen = 0;
            // (a 32-bit device counter that keeps counting through a long stepping loop: the difference is taken
            // modulo 2^32, so a wrap between two looks costs nothing)
            uint32_t counted = 0u;  // (nn_search.h kWantSlots: the counter's words, summed modulo 2^32)
            for (uint32_t k = 0; k < kWantSlots; ++k) counted += c->u_host[16 + k];
            const int64_t asked = (int64_t)(uint32_t)(counted - (uint32_t)c->halo_want_seen);  // by this chunk's iterations
            c->halo_want_seen = (int64_t)counted;
            c->halo_asked += asked;
            c->halo_lanes += c->ns * std::max<int64_t>(seen_iters, 0);
            if (undecided) {
                ++c->halo_looks;
                const int64_t per = std::max(executed, 1);
                const bool many = asked * 32 > c->ns * per, most = asked * 5 > 2 * c->ns * per;
                if (!many) {
                    c->halo_declined = true;
                } else if (most || c->halo_looks >= 2 || c->ns < kLarge) {
                    c->halo_sticky = true;
                    ++c->prof[6];
                    TRY(start_links_async(c));
                }
            } else if (((c->halo_iters >= kHaloLongRun && c->halo_asked * 100 >= c->halo_lanes) ||
                        (c->halo_iters >= kHaloVeryLongRun && c->halo_asked > 0)) && halo_memory_free(c)) {
                // (in the background: halo_declined stays, nothing waits.  Until round 6 ANY lane that had ever asked
                // started this build after 40 iterations -- 2.2 ms of GPU time and 1.6 GB at 10M points inside the loop
                // of a caller whose data hardly reads a halo: one window in seven of the headline bench 70 % slow.  What
                // the halos save such a loop is the tail of its searches -- the few packets that take a one-record walk:
                // 2-3 us per iteration, 2 % of a 10M-point step, 7 % of an eighth's -- which pays for the build after
                // ~1000 iterations.  So: at least 1 % of the lanes asking per iteration after 40, or anybody asking
                // after 1000 (a map that keeps being registered against), and twice the build's memory free.)
                ++c->prof[6];
                TRY(start_links_async(c));
            }
        }
        if (c->loop_host->done) break;
        if (grown && n == chunk) chunk = std::min(chunk * 2, 32);
    }
    return MI_ICP_OK;
}

static int loop_begin(mi_icp_ctx* c, int est, float max_distance, const float* init, float det_thresh,
                      int max_iterations, float rel_fitness, float rel_rmse) {
    if (!known_estimator(est))
        return fail(c, MI_ICP_ERR_INVALID, "unknown estimation type %d", est);
    TRY(comm_usable(c));
    DevLoop& L = *c->loop_host;
    std::memset(&L, 0, sizeof(L));
    L.est = est;
    L.det_thresh = det_thresh;
    L.T = load_T(init);
    L.A = host::is_identity4(L.T) ? host::identity4() : L.T;  // registration.cu:148-150
    L.X = xform_from(L.A);
    L.max_iterations = max_iterations;
    L.rel_fitness = rel_fitness;
    L.rel_rmse = rel_rmse;
    L.n_source_global = c->ns_global > 0 ? c->ns_global : c->ns;
    L.ready = estimator_ready(c, est) ? 1 : 0;
    L.history = 0ull;
    L.stamps = 0ull;
    // re-location (loop.h): sized only where the descent exists and the source is large enough to make its own seeds
    const bool can_locate = planes_available(c) && c->ns >= coarse_first_min() && c->src_bounds.p != nullptr && c->nt > 0;
    L.near2_ptr = can_locate ? (uint64_t)(uintptr_t)((const float*)c->nodes.p + kRecordNear2) : 0ull;
    L.src_bounds_ptr = can_locate ? (uint64_t)(uintptr_t)c->src_bounds.p : 0ull;
    c->relocate_armed = c->relocate_possible = can_locate;
    if (c->stamps_on) {  // (mi_icp_debug_set_step_stamps: armed -- minima at all ones -- before the loop's first launch)
        unsigned long long* st;
        TRY(ensure(c, c->stamps, kStampWords, &st));
        unsigned long long init[kStampWords] = {};
        init[0] = init[2] = ~0ull;
        HIPCHK(c, hipMemcpyAsync(st, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (`init` is a local)
        L.stamps = (uint64_t)(uintptr_t)st;
    }
    c->iter_reported = 0;
    if (c->iter_fn) {
        float* hist;
        TRY(ensure(c, c->loop_hist, (size_t)2 * kLoopHistory, &hist));
        if (!c->hist_host) HIPCHK(c, hipHostMalloc((void**)&c->hist_host, sizeof(float) * 2 * kLoopHistory, hipHostMallocDefault));
        L.history = (uint64_t)(uintptr_t)hist;
    }
    c->loop_active = false;
    c->loop_est = est;
    c->loop_r2 = max_distance * max_distance;
    // (a rank of a sharded job goes through the motions even with an empty shard: its peers wait for its sums)
    if (max_distance <= 0.0f || (c->ns <= 0 && c->nranks <= 1)) {
        // the reference logs an error and keeps going; every pass then yields an
        // empty result and identity updates, so the answer is `init` unchanged
        c->nn_valid = false;
        L.done = 1;
        return MI_ICP_OK;
    }
    DevLoop* d;
    TRY(ensure(c, c->loop_dev, 1, &d));
    HIPCHK(c, hipMemcpyAsync(d, &L, sizeof(DevLoop), hipMemcpyHostToDevice, c->stream));
    c->loop_active = true;
    // The first pass has no previous matches; launch_nn picks how it starts.  Halos: a context whose loops
    // have asked for them before starts the build now, next to the first pass; otherwise the first seeded
    // iteration says whether this loop needs them (loop_run).
    c->halo_declined = false;
    c->halo_want_seen = 0;
    c->halo_looks = 0;
    c->halo_iters_unseen = 0;
    c->halo_chunks = 0;
    c->ran_loop = true;
    {
        uint32_t* want;
        TRY(ensure(c, c->halo_want, kWantSlots, &want));
        HIPCHK(c, hipMemsetAsync(want, 0, kWantSlots * sizeof(uint32_t), c->stream));
    }
    c->halo_use = halo_poll(c);
    // A context whose loops have asked for halos before builds them NOW, on the loop's own stream, ahead of the first
    // pass -- which then starts from the queries' own seeds (launch_nn).  (Round 3 started the build on the private
    // stream next to the first pass and the match-order re-sort: the 2.5-ms build and those streaming kernels fought
    // for the memory system -- match_order_keys 28 us alone, 1.7 ms beside leaf_halo_build; leaf_halo_collect 0.73 ->
    // 1.7 ms -- and the loop waited for the build at its first seeded iteration anyway.)
    if (c->halo_sticky && !c->halo_use) {
        if (c->links_inflight) {
            TRY(start_links_async(c));
        } else {
            TRY(ensure_links(c));
            c->halo_use = halo_poll(c);
        }
    }
    static const bool no_resort = std::getenv("MI_ICP_NO_RESORT") != nullptr;  // A/B switch for tuning
    const bool resort = !no_resort && c->ns >= 32768 && (max_iterations >= 4 || max_iterations == 0);
    // (Round 5 tried the match-order sort AHEAD of the first search, on the leaves the queries fall into
    // (locate_by_planes): the first search gains nothing from packets that share their lines -- 0.85 ms against 0.79 --
    // and every later iteration of a clean registration loses ~20 %, because the order then follows where the queries
    // STARTED, not what they match: an 8-way shard's step 0.0393 ms instead of 0.0374.  Taken out; EXPERIMENTS.md.)
    TRY(loop_enqueue_evaluation(c, false));
    // from here on the packets follow the target's order (pays for itself in ~4 iterations)
    if (resort) TRY(resort_source_by_match(c));
    return MI_ICP_OK;
}

int mi_icp_icp_begin(mi_icp_ctx* c, int est, float max_distance, const float* init,
                     float det_thresh, mi_icp_result* out) {
    TRY(check_ctx(c));
    TRY(loop_begin(c, est, max_distance, init, det_thresh, 0, -1.0f, -1.0f));
    if (c->loop_active) {
        TRY(loop_pull(c));
        collect_pooled(c, 1);
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        fill_result(c, out);
    }
    return MI_ICP_OK;
}

int mi_icp_icp_iterate(mi_icp_ctx* c, int n_iterations, mi_icp_result* out) {
    TRY(check_ctx(c));
    if (n_iterations < 0) return fail(c, MI_ICP_ERR_INVALID, "icp_iterate: negative count");
    if (c->loop_active && n_iterations > 0) {
        // re-open the loop for n more updates: the update for the next iteration is formed
        // from the system of the last evaluation (resume = step without stats/test)
        DevLoop* d = (DevLoop*)c->loop_dev.p;
        loop_step_kernel<<<1, kStepThreads, 0, c->stream>>>(d, (double*)c->sys_dev.p, n_iterations, MailArgs{nullptr, nullptr, 0, 1, 0u, nullptr, nullptr});
        KCHK(c);
        TRY(loop_run(c, n_iterations));
        // (a stepping loop never reaches mi_icp_registration_icp's exit: the halo build's candidate scratch -- 0.9 GB for
        // a 10M-point target -- is dropped here, once the build is complete; the stream has just been synchronised)
        (void)halo_poll(c);
        release_links_scratch(c);
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        fill_result(c, out);
    }
    return MI_ICP_OK;
}

int mi_icp_registration_icp(mi_icp_ctx* c, int est, float max_distance, const float* init,
                            const mi_icp_params* params, mi_icp_result* out) {
    TRY(check_ctx(c));
    if (!out) return fail(c, MI_ICP_ERR_INVALID, "registration_icp: out is null");
    mi_icp_params p = {1e-6f, 1e-6f, 30, 1e-6f};
    if (params) p = *params;
    // a negative threshold can never be undercut by |difference|: same as "never converges"
    TRY(loop_begin(c, est, max_distance, init, p.det_thresh, std::max(p.max_iteration, 0),
                   std::max(p.relative_fitness, 0.0f), std::max(p.relative_rmse, 0.0f)));
    if (c->loop_active) {
        TRY(loop_pull(c));
        collect_pooled(c, 1);
        if (!c->loop_host->done) TRY(loop_run(c, std::max(p.max_iteration, 0)));
        (void)halo_poll(c);
        release_links_scratch(c);
    }
    std::memset(out, 0, sizeof(*out));
    fill_result(c, out);
    return MI_ICP_OK;
}

}  // extern "C"
