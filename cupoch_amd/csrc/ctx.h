// ctx.h -- what the translation units of libmi_icp.so share: the context (mi_icp_ctx), its buffers, the error /
// allocation / staging helpers, and the declarations of the host-side functions one unit offers the others.
//   mi_icp.hip       context life cycle, the correspondence search, the reduction, the device-resident loop
//   mi_build.hip     target tree (kd cells, groups, levels, halos), source staging, the match-order re-sort
//   mi_geometry.hip  Transform / bounds / affine / covariances / VoxelDownSample / depth frames / RGB-D odometry / colours
//   mi_knn.hip       EstimateNormals, KDTreeFlann::SearchKNN / SearchRadius, colour gradients, Colored ICP's entry
//   mi_comm.hip      the ranks' exchange: mailbox, device inboxes, in-library RCCL, self-test and choice
//   mi_debug.hip     include/mi_icp_debug.h (test-only entry points)
// Kernels without template parameters are `static` in their headers, so a header may be included by several units.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mi_icp.h"
#include "../../include/mi_icp_debug.h"
#include "device_utils.h"
#include "host_solver.h"
#include "loop.h"
#include "mailbox.h"
#include "primitives.h"

namespace mi {
namespace eng {

using host::Mat4;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace eng
}  // namespace mi

struct mi_icp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- target (Morton order) ----
    int64_t nt = 0;
    int nleaf = 0;
    int64_t nts = 0;  // sorted positions of the target incl. padding slots (kd_cells.h)
    uint32_t leaf_first = 1, nrecords = 0;  // 8-ary tree: first last-level node id, record count
    bool t_has_nrm = false, t_has_cov = false, t_has_int = false, t_has_grad = false, t_has_rec = false;
    mi::eng::DevBuf tblk, tnrm, trec, tcov, tgrad, nodes, inv_t, tidx, thalo, tlinks_tmp;  // (leaf regions: the leaf lines' fourth rows, lreg_of)
    mi::eng::DevBuf cell_planes, cell_samples, cell_cstart, cell_gstart, cell_boxes, cell_hist;  // (kd_planes.h: the sample's boxes and histograms)
    mi::eng::DevBuf gplanes;   // every group's own 511 split planes (kd_build.h): with cell_planes / cell_gstart the binary descent of locate_by_planes
    int cell_levels = -1;      // levels of cell planes of the present target; < 0: no kd cells (Morton-run fallback tree), nothing to descend
    uint32_t* cell_total_host = nullptr;  // pinned
    bool inv_t_valid = false;
    bool links_ready = false, links_allowed = false;  // leaf_halo.h
    // the halos are built on a private stream (start_links_async)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_links = nullptr;
    bool links_inflight = false;
    int last_search_kind = -1;  // mi_icp_debug.h
    int last_voxel_path = -1;   // mi_icp_debug.h
    // Halos are built when a registration loop's searches ask for them (nn_search.h counts the lanes one would
    // serve): clean data never does.  A context whose loops have asked before starts the build with the loop.
    bool halo_sticky = false;
    bool ran_loop = false;  // (a context that has registered before and gets a SMALL target starts the build behind the tree)
    bool halo_declined = false;  // this loop's searches have been looked at and did not ask
    int64_t halo_iters = 0;      // seeded iterations against this target since it was set ...
    int64_t halo_asked = 0;      // ... and the lanes that asked for a halo in them
    int64_t halo_lanes = 0;      // ... out of this many lanes (source points x iterations looked at)
    int64_t halo_iters_unseen = 0;  // iterations since the counter was last looked at
    uint32_t halo_chunks = 0;    // chunks of a declined loop (it looks at the counter every eighth)
    int64_t halo_want_seen = 0;  // the counter's value at the last look (it is zeroed when a loop begins)
    int halo_looks = 0;          // looks of this loop while undecided
    bool halo_use = false;       // the loop's launches take the halos (looked up once per chunk: an event query costs microseconds)
    mi::eng::DevBuf halo_want;            // the counter (nn_search.h kWantSlots words, summed by the host)

    // ---- source (Morton order) ----
    int64_t ns = 0, ns_global = 0;
    bool s_has_nrm = false, s_has_cov = false, s_has_int = false;
    float lambda_geometric = 0.968f;  // colored ICP (colored_icp.cu:47-51)
    mi::eng::DevBuf sx, sy, sz, sperm, snrm, scov, sint, nn_idx, nn_d2, inv_s;
    mi::eng::DevBuf alt[9];  // second set of the source arrays (match-order re-sort ping-pong)
    bool inv_s_valid = false;
    bool nn_valid = false;  // nn_idx holds a search result (usable as seed / correspondences)
    mi::eng::DevBuf src_bounds;    // min[3], max[3] of the staged source (the loop's step sizes the displacement of its corners: loop.h)
    bool relocate_armed = false;   // this loop's next chunk of iterations carries the gated re-location launches (loop_run)
    bool relocate_possible = false;  // ... this loop's step sizes its displacement (loop_begin): the launches may be armed again

    // ---- explicit correspondence set ----
    mi::eng::DevBuf user_pairs;
    int64_t n_user_pairs = -1;  // < 0: use the nearest-neighbour result

    // ---- scratch ----
    mi::eng::DevBuf keys0, keys1, vals0, vals1, hist, scan_tmp, bounds_part, bounds;
    mi::eng::DevBuf partial, sys_dev, dense_idx, flags, pairs_out, seg_start;
    mi::eng::DevBuf stage[6];
    mi::eng::DevBuf tscale;   // scratch of kd_build.h tree_scale
    mi::eng::DevBuf knn_idx, knn_flags;  // the k-NN lists' index rows, [XCD][row][slot][lane], and the rows' claim flags (knn_normals.h KnnSlab)
    float vx_refused_voxel = 0.0f;  // the last voxel size / cloud size the dense path's plan turned away (mi_icp_voxel_downsample)
    int64_t vx_refused_n = 0;
    int vx_order = 0;         // LDS adds of one instruction served in lane order (voxel_dense.h)?  0: not checked yet, 1: yes, -1: no
    mi::eng::DevBuf vx_tab;   // VoxelDownSample of a dense grid (voxel_dense.h): the [tile][bucket] table, bucket starts, status and control words
    mi::eng::DevBuf vpay[6];  // VoxelDownSample: two sets of payload arrays (points, normals, colours) the radix passes alternate between
    double* sys_host = nullptr;  // pinned, 32 doubles + spare
    float* f_host = nullptr;     // pinned, 16 floats
    uint32_t* u_host = nullptr;  // pinned, 16 + kWantSlots words ([0]: counts read back by the one-shot entry points, [16..]: the halo_want counter's words)
    void* od_host = nullptr;     // pinned OdState mirror (odometry), allocated on first use

    // ---- registration loop (device-resident, loop.h) ----
    mi::eng::DevBuf loop_dev, ticket;
    mi::DevLoop* loop_host = nullptr;  // pinned mirror of the device state
    bool loop_active = false;
    mi_icp_iteration_fn iter_fn = nullptr;  // per-iteration report (mi_icp_set_iteration_callback)
    void* iter_user = nullptr;
    mi::eng::DevBuf loop_hist;
    float* hist_host = nullptr;  // pinned, kLoopHistory * 2 floats
    int iter_reported = 0;       // iterations of this loop the callback has seen
    float loop_r2 = 0.0f;
    int loop_est = 0;

    // ---- multi-GPU ----
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    // the node's mailbox (mailbox.h): POSIX shared memory registered with HIP, or null
    mi::MailBox* mail_host = nullptr;
    mi::MailBox* mail_dev = nullptr;
    size_t mail_bytes = 0;
    std::string mail_name;
    bool mail_linked = false;   // the name still exists and is this context's to remove
    // device inboxes (mailbox.h): this rank's, the peers' as opened through HIP IPC, and the device-side table of all
    unsigned long long* inbox = nullptr;
    unsigned long long* inbox_peer[mi::kMailRanks] = {};
    mi::eng::DevBuf inbox_table;
    bool comm_broken = false;   // an exchange has failed: the ranks' counters are apart
    // how the ranks exchange their sums: 0 nothing to exchange, 1 the box's host-memory words, 2 device inboxes,
    // 3 in-library RCCL all-reduce.  Set when the communicator is made, changed by mi_icp_comm_autotune.
    int xchg = 0;
    uint32_t tune_epoch = 0;    // this rank's count of host-side gathers through the box (box_gather)
    mi::eng::DevBuf mail_state;  // [0]: this rank's exchange counter, [1]: error flag of the one-shot exchange

    // ---- private scratch context: PointCloud::EstimateNormals builds its own tree there, so
    // that the target / source / loop state of THIS context survive the call ----
    mi_icp_ctx* aux = nullptr;

    // ---- instrumentation ----
    mi::eng::DevBuf stamps;    // loop.h "where an iteration's time goes" (mi_icp_debug_set_step_stamps)
    bool stamps_on = false;
    bool profiling = false;
    static constexpr int kEvPairs = 16;   // per kind: one pair per launch of a chunk
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evp[2][kEvPairs][2] = {};
    int evp_n[2] = {0, 0};
    bool ev_pending_nn = false, ev_pending_red = false;
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace mi {
namespace eng {

// the leaves' region records: the fourth row of every leaf line (device_utils.h: kLeafRegOffset, kLeafRegStride)
inline float* lreg_of(const mi_icp_ctx* c) { return c->tblk.p ? (float*)c->tblk.p + mi::kLeafRegOffset : nullptr; }

inline int fail(mi_icp_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), MI_ICP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                        hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

#define KCHK(c) HIPCHK(c, hipGetLastError())

#define TRY(expr)                \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != MI_ICP_OK) return rc_; \
    } while (0)

template <class T>
int ensure(mi_icp_ctx* c, DevBuf& b, size_t count, T** out) {
    const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    if (b.bytes < bytes) {
        if (b.p) {
            // buffers may still be in use by enqueued work
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        HIPCHK(c, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    *out = (T*)b.p;
    return MI_ICP_OK;
}

inline void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// device view of a caller buffer (copied through a context-owned staging buffer
// when it lives in host memory)
template <class T>
int to_device(mi_icp_ctx* c, const T* src, size_t count, int mem_kind, DevBuf& stage,
              const T** out) {
    if (!src || count == 0) {
        *out = nullptr;
        return MI_ICP_OK;
    }
    if (mem_kind == MI_ICP_DEVICE) {
        *out = src;
        return MI_ICP_OK;
    }
    T* d;
    TRY(ensure(c, stage, count, &d));
    HIPCHK(c, hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *out = d;
    return MI_ICP_OK;
}

template <class T>
int from_device(mi_icp_ctx* c, const T* dev, T* dst, size_t count, int mem_kind) {
    if (!dst || count == 0) return MI_ICP_OK;
    HIPCHK(c, hipMemcpyAsync(dst, dev, count * sizeof(T),
                             mem_kind == MI_ICP_DEVICE ? hipMemcpyDeviceToDevice
                                                       : hipMemcpyDeviceToHost,
                             c->stream));
    return MI_ICP_OK;
}

inline int blocks_for(int64_t n, int per = 256) { return (int)std::max<int64_t>(1, (n + per - 1) / per); }

inline Xform make_xform(const Mat4& T) { return xform_from(T); }

inline Mat4 load_T(const float* T) {
    if (!T) return host::identity4();
    Mat4 m;
    std::memcpy(m.data(), T, sizeof(float) * 16);
    return m;
}

struct EvTimer {
    mi_icp_ctx* c;
    int slot;  // 0: nn, 1: reduce
    hipEvent_t stop = nullptr;
    bool pooled;
    EvTimer(mi_icp_ctx* ctx, int s, bool in_loop) : c(ctx), slot(s), pooled(in_loop) {
        if (!c->profiling) return;
        if (pooled) {
            if (c->evp_n[slot] >= mi_icp_ctx::kEvPairs) return;
            const int i = c->evp_n[slot]++;
            (void)hipEventRecord(c->evp[slot][i][0], c->stream);
            stop = c->evp[slot][i][1];
        } else {
            (void)hipEventRecord(c->ev[slot * 2], c->stream);
            stop = c->ev[slot * 2 + 1];
        }
    }
    ~EvTimer() {
        if (!stop) return;
        (void)hipEventRecord(stop, c->stream);
        if (!pooled) (slot == 0 ? c->ev_pending_nn : c->ev_pending_red) = true;
    }
};

inline void collect_events(mi_icp_ctx* c) {  // call after the stream has been synchronised
    float ms = 0.0f;
    if (c->ev_pending_nn && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) {
        c->prof[0] += ms;
        c->prof[1] += 1;
    }
    if (c->ev_pending_red && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) {
        c->prof[2] += ms;
        c->prof[3] += 1;
    }
    c->ev_pending_nn = c->ev_pending_red = false;
}

// pooled events of a loop chunk: only the first `executed` launches did real work
inline void collect_pooled(mi_icp_ctx* c, int executed) {
    for (int slot = 0; slot < 2; ++slot) {
        for (int i = 0; i < c->evp_n[slot] && i < executed; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->evp[slot][i][0], c->evp[slot][i][1]) == hipSuccess) {
                c->prof[slot * 2] += ms;
                c->prof[slot * 2 + 1] += 1;
            }
        }
        c->evp_n[slot] = 0;
    }
}

// the ranks exchange through the mailbox (host-memory words or device inboxes), not through RCCL
inline bool mail_on(const mi_icp_ctx* c) { return c->mail_dev != nullptr && (c->xchg == 1 || c->xchg == 2); }

inline int check_ctx(mi_icp_ctx* c) {
    if (!c) return MI_ICP_ERR_INVALID;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, MI_ICP_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
    return MI_ICP_OK;
}

// ---- mi_build.hip
int compute_bounds(mi_icp_ctx* c, const float* pts, int64_t n, float** bounds_out);  // min[3], max[3], extent into c->bounds
int sort_buffers(mi_icp_ctx* c, int64_t n, SortBuffers* sb);
int morton_order(mi_icp_ctx* c, const float* pts, int64_t n, const uint32_t** order, bool kd_refine,
                 const float* grid_bounds = nullptr, int grid_bits = 0, float** own_bounds = nullptr);
int ensure_links(mi_icp_ctx* c);          // the halos complete before the next kernel on the context's stream
int start_links_async(mi_icp_ctx* c);     // ... started on the private stream
bool halo_poll(mi_icp_ctx* c);            // are they there?  never waits
int drain_links(mi_icp_ctx* c);
void release_links_scratch(mi_icp_ctx* c);
int resort_source_by_match(mi_icp_ctx* c);
int occupancy_build(int which);           // mi_icp_debug_occupancy: the kernels each unit owns
// ---- mi_icp.hip
int launch_nn(mi_icp_ctx* c, const Mat4& T, float r2, bool seed, unsigned long long* stats = nullptr,
              const DevLoop* loop = nullptr);
int occupancy_loop(int which);
bool planes_available(const mi_icp_ctx* c);
int launch_locate_by_planes(mi_icp_ctx* c, const Xform& X, const DevLoop* loop, int gated);
// ---- mi_geometry.hip
int occupancy_geometry(int which);
// ---- mi_comm.hip
MailArgs mail_args(const mi_icp_ctx* c);
void mailbox_close(mi_icp_ctx* c);
int comm_failed(mi_icp_ctx* c, const char* what);
int comm_usable(mi_icp_ctx* c);
int allreduce_system(mi_icp_ctx* c);
void comm_release(mi_icp_ctx* c);         // mi_icp_destroy: the mailbox and the communicator

}  // namespace eng
}  // namespace mi
