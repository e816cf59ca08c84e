// mi_icp.hip -- libmi_icp.so: context, host orchestration and the C ABI
// declared in include/mi_icp.h.  gfx950 only.
//
// The host loop mirrors registration::RegistrationICP
// (registration/registration.cu:121-172) but keeps everything device-resident:
// one nearest-neighbour launch + one reduction launch (+ its 1-block finish)
// per iteration, one 256-byte D2H copy, the 6x6 solve on the host, and the new
// 4x4 passed back as a kernel argument.  Nothing is allocated inside the loop.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cctype>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_icp.h"
#include "../../include/mi_icp_debug.h"
#include "device_utils.h"
#include "depth_kernels.h"
#include "fused_small.h"
#include "geometry_kernels.h"
#include "host_solver.h"
#include "kd_build.h"
#include "kd_cells.h"
#include "kd_refine.h"
#include "knn_normals.h"
#include "lbvh.h"
#include "lzf.h"
#include "leaf_halo.h"
#include "loop.h"
#include "nn_search.h"
#include "odometry.h"
#include "primitives.h"
#include "reduce.h"

using namespace mi;
using host::Mat4;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // (optional)
};

bool load_rccl(Rccl& r) {
    if (r.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return false;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    return r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
}

Rccl g_rccl;

}  // namespace

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
    DevBuf tblk, tnrm, trec, tcov, tgrad, nodes, inv_t, tidx, thalo, tlinks_tmp;  // (leaf regions: the leaf lines' fourth rows, lreg_of)
    DevBuf cell_planes, cell_samples, cell_cstart, cell_gstart;
    uint32_t* cell_total_host = nullptr;  // pinned
    bool inv_t_valid = false;
    bool links_ready = false, links_allowed = false;  // leaf_halo.h
    // the halos are built on a private stream (start_links_async)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_links = nullptr;
    bool links_inflight = false;
    int last_search_kind = -1;  // mi_icp_debug.h
    // Halos are built when a registration loop's searches ask for them (nn_search.h counts the lanes one would
    // serve): clean data never does.  A context whose loops have asked before starts the build with the loop.
    bool halo_sticky = false;
    bool ran_loop = false;  // (a context that has registered before and gets a SMALL target starts the build behind the tree)
    bool halo_declined = false;  // this loop's searches have been looked at and did not ask
    int64_t halo_iters = 0;      // seeded iterations against this target since it was set ...
    int64_t halo_asked = 0;      // ... and the lanes that asked for a halo in them
    int64_t halo_want_seen = 0;  // the counter's value at the last look (it is zeroed when a loop begins)
    int halo_looks = 0;          // looks of this loop while undecided
    bool halo_use = false;       // the loop's launches take the halos (looked up once per chunk: an event query costs microseconds)
    DevBuf halo_want;            // the counter (one word)

    // ---- source (Morton order) ----
    int64_t ns = 0, ns_global = 0;
    bool s_has_nrm = false, s_has_cov = false, s_has_int = false;
    float lambda_geometric = 0.968f;  // colored ICP (colored_icp.cu:47-51)
    DevBuf sx, sy, sz, sperm, snrm, scov, sint, nn_idx, nn_d2, inv_s;
    DevBuf alt[9];  // second set of the source arrays (match-order re-sort ping-pong)
    bool inv_s_valid = false;
    bool nn_valid = false;  // nn_idx holds a search result (usable as seed / correspondences)

    // ---- explicit correspondence set ----
    DevBuf user_pairs;
    int64_t n_user_pairs = -1;  // < 0: use the nearest-neighbour result

    // ---- scratch ----
    DevBuf keys0, keys1, vals0, vals1, hist, scan_tmp, bounds_part, bounds;
    DevBuf partial, sys_dev, dense_idx, flags, pairs_out, seg_start;
    DevBuf stage[6];
    DevBuf tscale;   // scratch of kd_build.h tree_scale
    DevBuf knn_idx;  // candidate indices of the small k-NN lists, [packet][slot][lane] (knn_normals.h)
    DevBuf vpay[6];  // VoxelDownSample: two sets of payload arrays (points, normals, colours) the radix passes alternate between
    double* sys_host = nullptr;  // pinned, 32 doubles + spare
    float* f_host = nullptr;     // pinned, 16 floats
    uint32_t* u_host = nullptr;  // pinned, 16 words ([0]: counts read back by the one-shot entry points, [8]: the halo_want counter)
    void* od_host = nullptr;     // pinned OdState mirror (odometry), allocated on first use

    // ---- registration loop (device-resident, loop.h) ----
    DevBuf loop_dev, ticket;
    DevLoop* loop_host = nullptr;  // pinned mirror of the device state
    bool loop_active = false;
    mi_icp_iteration_fn iter_fn = nullptr;  // per-iteration report (mi_icp_set_iteration_callback)
    void* iter_user = nullptr;
    DevBuf loop_hist;
    float* hist_host = nullptr;  // pinned, kLoopHistory * 2 floats
    int iter_reported = 0;       // iterations of this loop the callback has seen
    float loop_r2 = 0.0f;
    int loop_est = 0;

    // ---- multi-GPU ----
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    // the node's mailbox (mailbox.h): POSIX shared memory registered with HIP, or null
    MailBox* mail_host = nullptr;
    MailBox* mail_dev = nullptr;
    size_t mail_bytes = 0;
    std::string mail_name;
    bool mail_linked = false;   // the name still exists and is this context's to remove
    // device inboxes (mailbox.h): this rank's, the peers' as opened through HIP IPC, and the device-side table of all
    unsigned long long* inbox = nullptr;
    unsigned long long* inbox_peer[kMailRanks] = {};
    DevBuf inbox_table;
    bool comm_broken = false;   // an exchange has failed: the ranks' counters are apart
    // how the ranks exchange their sums: 0 nothing to exchange, 1 the box's host-memory words, 2 device inboxes,
    // 3 in-library RCCL all-reduce.  Set when the communicator is made, changed by mi_icp_comm_autotune.
    int xchg = 0;
    uint32_t tune_epoch = 0;    // this rank's count of host-side gathers through the box (box_gather)
    DevBuf mail_state;  // [0]: this rank's exchange counter, [1]: error flag of the one-shot exchange

    // ---- private scratch context: PointCloud::EstimateNormals builds its own tree there, so
    // that the target / source / loop state of THIS context survive the call ----
    mi_icp_ctx* aux = nullptr;

    // ---- instrumentation ----
    bool profiling = false;
    static constexpr int kEvPairs = 16;   // per kind: one pair per launch of a chunk
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evp[2][kEvPairs][2] = {};
    int evp_n[2] = {0, 0};
    bool ev_pending_nn = false, ev_pending_red = false;
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// the leaves' region records: the fourth row of every leaf line (device_utils.h: kLeafRegOffset, kLeafRegStride)
static inline float* lreg_of(const mi_icp_ctx* c) { return c->tblk.p ? (float*)c->tblk.p + mi::kLeafRegOffset : nullptr; }

namespace {

int fail(mi_icp_ctx* c, int code, const char* fmt, ...) {
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

void release(DevBuf& b) {
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

Xform make_xform(const Mat4& T) { return xform_from(T); }

Mat4 load_T(const float* T) {
    if (!T) return host::identity4();
    Mat4 m;
    std::memcpy(m.data(), T, sizeof(float) * 16);
    return m;
}

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
                 const float* grid_bounds = nullptr, int grid_bits = 0, float** own_bounds = nullptr) {
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
    int levels;            // ncells = 2^levels
};

int kd_cell_layout(mi_icp_ctx* c, const float* pts, int64_t n, CellLayout* out) {
    const int d = cell_levels_for(n);
    const int ncells = 1 << d;
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    float2* planes;
    TRY(ensure(c, c->cell_planes, (size_t)ncells * 2, &planes));
    if (d > 0) {
        const int64_t S = std::min<int64_t>(n, (int64_t)kCellSamples * ncells);
        float* samp;
        TRY(ensure(c, c->cell_samples, (size_t)S * 3, &samp));
        cells_sample_gather<<<blocks_for(S), 256, 0, c->stream>>>(pts, n, S, samp);
        KCHK(c);
        const int stages = (d + kCellStageLevels - 1) / kCellStageLevels;
        int base = 0;
        int cur = 0;
        for (int st = 0; st < stages; ++st) {
            const int levels = (st == 0) ? d - kCellStageLevels * (stages - 1) : kCellStageLevels;
            if (base > 0) {  // samples grouped by their depth-`base` cell
                cells_assign<uint64_t><<<blocks_for(S), 256, 0, c->stream>>>(samp, S, planes, base, sb.keys[0], sb.vals[0]);
                KCHK(c);
                cur = radix_sort_pairs(c->stream, sb, S, base);
                KCHK(c);
            }
            cells_planes<<<1 << base, kKdThreads, 0, c->stream>>>(samp, S, sb.keys[cur], sb.vals[cur], base, levels, planes);
            KCHK(c);
            base += levels;
        }
    }
    uint32_t *cstart, *gstart;
    TRY(ensure(c, c->cell_cstart, (size_t)ncells + 2, &cstart));
    TRY(ensure(c, c->cell_gstart, (size_t)ncells, &gstart));
    cells_assign<uint32_t><<<blocks_for(n), 256, 0, c->stream>>>(pts, n, planes, d, (uint32_t*)sb.keys[0], sb.vals[0]);
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
    out->levels = d;
    return MI_ICP_OK;
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

void collect_events(mi_icp_ctx* c) {  // call after the stream has been synchronised
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
void collect_pooled(mi_icp_ctx* c, int executed) {
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

// Every leaf's halo (leaf_halo.h): what lets a seeded query whose cube pokes out of its leaf's region
// finish without a tree walk.  Built once per target: right behind the tree on a context that has
// registered before (mi_icp_set_target), otherwise by the first registration loop / seeded search
// (one-shot searches, k-NN and normal estimation on a fresh context never pay for it).
int build_links(mi_icp_ctx* c, hipStream_t st) {
    static const bool no_links = std::getenv("MI_ICP_NO_LINKS") != nullptr;  // A/B switch
    if (!c->links_allowed || no_links) return MI_ICP_OK;  // (every leaf's largest reach is 0 as built: no query asks for a line)
    float* halo;
    const size_t ntiles = ((size_t)c->nleaf + 63) / 64;
    TRY(ensure(c, c->thalo, ntiles * 64 * kHaloLines * kHaloLineFloats, &halo));
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
    static const bool sync_links = std::getenv("MI_ICP_LINKS_SYNC") != nullptr;  // A/B switch
    if (c->nt <= 0 || c->links_ready || c->links_inflight || !c->links_allowed) return MI_ICP_OK;
    if (sync_links) return ensure_links(c);
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

// ---- nearest-neighbour pass --------------------------------------------------
constexpr int64_t kHaloAheadMax = 2000000;  // targets below this get their halos right behind the tree on a context that has registered before
// sources of at least this many points make their own seeds for a first pass (tuning knob MI_ICP_COARSE_MIN)
static int64_t coarse_first_min() {
    static const int64_t v = [] { const char* s = std::getenv("MI_ICP_COARSE_MIN"); return s ? std::atoll(s) : (int64_t)1 << 16; }();
    return v;
}

int launch_nn(mi_icp_ctx* c, const Mat4& T, float r2, bool seed, unsigned long long* stats = nullptr,
              const DevLoop* loop = nullptr) {
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
        locate_leaves<<<blocks_for(c->ns), 256, 0, c->stream>>>((const float*)c->sx.p, (const float*)c->sy.p,
                                                                (const float*)c->sz.p, (int)c->ns,
                                                                (const float*)c->nodes.p, c->leaf_first,
                                                                (uint32_t)c->nleaf, X, loop, idx);
        KCHK(c);
        self_seeded = true;
    }
    c->last_search_kind = use_seed ? 1 : (self_seeded ? 2 : 0);
    static const bool first_solo = std::getenv("MI_ICP_FIRST_SOLO") != nullptr;  // experiment: every lane walks on its own
    if (first_solo && !use_seed && !self_seeded && !stats && c->leaf_first >= 1u) {
        const uint32_t npackets = (uint32_t)((c->ns + 63) / 64);
        const uint32_t grid = ((npackets + 7u) / 8u) * 8u;
        nn_solo_kernel<<<grid, kNNThreads, 0, c->stream>>>((const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, (int)c->ns,
                                                          (const float*)c->nodes.p, (const float*)c->tblk.p, c->leaf_first, X, loop, r2,
                                                          npackets, idx, loop ? nullptr : d2);
        KCHK(c);
        c->nn_valid = true;
        c->n_user_pairs = -1;
        return MI_ICP_OK;
    }
    launch(use_seed || self_seeded, (const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, c->ns, idx,
           loop ? nullptr : d2);
    KCHK(c);
    c->nn_valid = true;
    c->n_user_pairs = -1;
    return MI_ICP_OK;
}

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
// 1.25M-5M point shards (fewer partials for the finishing block); tuning knob MI_ICP_REDUCE_EPT
static int reduce_elems_per_thread() {
    static const int v = [] { const char* s = std::getenv("MI_ICP_REDUCE_EPT"); const int k = s ? std::atoi(s) : 16; return k > 0 ? k : 16; }();
    return v;
}

MailArgs mail_args(const mi_icp_ctx* c);  // (below, with the communicator code)
// the ranks exchange through the mailbox (host-memory words or device inboxes), not through RCCL
inline bool mail_on(const mi_icp_ctx* c) { return c->mail_dev != nullptr && (c->xchg == 1 || c->xchg == 2); }

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
            wide, std::min<int64_t>(kReduceBlocks, blocks_for(a.count, kReduceThreads * reduce_elems_per_thread())));
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
        static const bool no_fused_step = std::getenv("MI_ICP_NO_FUSED_STEP") != nullptr;  // A/B switch
        EvTimer t(c, 1, loop != nullptr);
        const MailArgs no_mail = {nullptr, nullptr, 0, 1, 0u, nullptr, nullptr};
        const bool mail = mail_on(c);
        if (fuse_step && loop && mail && !no_fused_step) {  // N ranks on one node: exchange + step in the finishing block
            reduce_pt2pl_kernel<4, 2><<<g2, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, sys, mail_args(c));
            if (stepped) *stepped = true;
        } else if (fuse_step && loop && !c->comm && !c->mail_dev && !no_fused_step) {
            reduce_pt2pl_kernel<4, 1><<<g2, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, sys, no_mail);
            if (stepped) *stepped = true;
        } else {
            reduce_pt2pl_kernel<4, 0><<<g2, kReduceThreads, 0, c->stream>>>(a, X, loop, partial, ticket, sys, no_mail);
        }
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

MailArgs mail_args(const mi_icp_ctx* c) {
    MailArgs m;
    m.box = c->mail_dev;
    m.seq_dev = (uint32_t*)c->mail_state.p;
    const bool direct = c->inbox != nullptr && c->xchg == 2;
    m.inbox = direct ? c->inbox : nullptr;
    m.peers = direct ? (unsigned long long* const*)c->inbox_table.p : nullptr;
    m.rank = c->rank;
    m.nranks = c->nranks;
    static const uint32_t limit = [] { const char* e = std::getenv("MI_ICP_MAIL_SPIN_LIMIT"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : kMailSpinLimit; }();
    m.spin_limit = limit;
    return m;
}

// Device inboxes: nobody may free an inbox a peer's kernel could still write to.  Every rank closes what it opened
// and says so in the box; an inbox is freed once every peer has (or after 2 s: a peer that died holds no kernel).
void inbox_close(mi_icp_ctx* c) {
    if (!c->inbox) return;
    (void)hipStreamSynchronize(c->stream);
    for (int r = 0; r < c->nranks && r < kMailRanks; ++r)
        if (r != c->rank && c->inbox_peer[r]) (void)hipIpcCloseMemHandle(c->inbox_peer[r]);
    for (auto& p : c->inbox_peer) p = nullptr;
    if (c->mail_host) {
        MailBox* box = c->mail_host;
        __atomic_store_n(&box->inbox_closed[c->rank], 1u, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool all = true;
            for (int r = 0; r < c->nranks && r < kMailRanks; ++r) all = all && __atomic_load_n(&box->inbox_closed[r], __ATOMIC_ACQUIRE) != 0u;
            if (all || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    (void)hipFree(c->inbox);
    c->inbox = nullptr;
    (void)hipGetLastError();
}

// ... set up after the box itself (every rank is attached): inbox, handle into the box, wait for the peers', open
// them.  All ranks end in the same mode: a rank that fails says so in the box before the others look.
bool inbox_open(mi_icp_ctx* c, MailBox* box, int nranks, int rank, const std::function<bool()>& late) {
    auto wait_all = [&](uint32_t state) {
        for (;;) {
            bool all = true;
            for (int r = 0; r < nranks; ++r) all = all && __atomic_load_n(&box->inbox_state[r], __ATOMIC_ACQUIRE) >= state;
            if (all) return true;
            if (late()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };
    auto give_up = [&] { __atomic_store_n(&box->device_failed, 1u, __ATOMIC_RELEASE); };
    const size_t bytes = kMailInboxWords * sizeof(unsigned long long);
    void* mine = nullptr;
    if (hipExtMallocWithFlags(&mine, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(mine, 0, bytes) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&box->inbox[rank], mine) != hipSuccess) {
        (void)hipGetLastError();
        give_up();
    }
    __atomic_store_n(&box->inbox_state[rank], 1u, __ATOMIC_RELEASE);
    if (!wait_all(1u)) give_up();
    bool ok = __atomic_load_n(&box->device_failed, __ATOMIC_ACQUIRE) == 0u;
    if (ok) {
        for (int r = 0; r < nranks && ok; ++r) {
            if (r == rank) {
                c->inbox_peer[r] = (unsigned long long*)mine;
            } else {
                void* p = nullptr;
                if (hipIpcOpenMemHandle(&p, box->inbox[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    (void)hipGetLastError();
                    give_up();
                    ok = false;
                } else {
                    c->inbox_peer[r] = (unsigned long long*)p;
                }
            }
        }
    }
    __atomic_store_n(&box->inbox_state[rank], 2u, __ATOMIC_RELEASE);
    if (!wait_all(2u)) give_up();
    ok = __atomic_load_n(&box->device_failed, __ATOMIC_ACQUIRE) == 0u;
    unsigned long long** table = nullptr;
    if (ok && (ensure(c, c->inbox_table, kMailRanks, &table) != MI_ICP_OK ||
               hipMemcpy(table, c->inbox_peer, sizeof(c->inbox_peer), hipMemcpyHostToDevice) != hipSuccess)) {
        // (too late to tell the others: they will wait for this rank's posts in vain and time out; cannot happen short of an out-of-memory)
        ok = false;
    }
    if (!ok) {
        for (int r = 0; r < nranks; ++r)
            if (r != rank && c->inbox_peer[r]) (void)hipIpcCloseMemHandle(c->inbox_peer[r]);
        for (auto& p : c->inbox_peer) p = nullptr;
        __atomic_store_n(&box->inbox_closed[rank], 1u, __ATOMIC_RELEASE);
        if (mine) (void)hipFree(mine);
        (void)hipGetLastError();
        return false;
    }
    c->inbox = (unsigned long long*)mine;
    return true;
}

void mailbox_close(mi_icp_ctx* c) {
    inbox_close(c);
    if (c->mail_host) {
        (void)hipHostUnregister(c->mail_host);
        (void)munmap(c->mail_host, c->mail_bytes);
    }
    // (the name is rank 0's to remove, and only while it still refers to this box: once every rank has
    // attached rank 0 unlinks it at once, so that no later job -- or crash -- finds it)
    if (c->mail_linked && !c->mail_name.empty()) (void)shm_unlink(c->mail_name.c_str());
    c->mail_linked = false;
    c->mail_host = c->mail_dev = nullptr;
    c->mail_name.clear();
    c->xchg = c->comm ? 3 : 0;
    c->tune_epoch = 0;
}

static long mail_attach_timeout_ms() {
    static const long v = [] { const char* e = std::getenv("MI_ICP_MAIL_ATTACH_MS"); const long t = e ? std::atol(e) : 0; return t > 0 ? t : 30000L; }();
    return v;
}

// Rank 0 creates and zeroes the box and waits until every other rank has mapped AND registered it with
// HIP (`attached`), then declares it in use (`go`) and removes the name.  The others open the name, wait for
// `ready`, refuse a box that is in use already (a leftover of another job under the same name: its `go` is
// set -- they retry until rank 0 has replaced it), register, attach and wait for `go`.  All ranks of a job
// pass the same name.  MI_ICP_MAIL_ATTACH_MS: how long anybody waits (default 30 s).
int mailbox_open(mi_icp_ctx* c, const std::string& name, int nranks, int rank) {
    mailbox_close(c);
    if (nranks > kMailRanks) return fail(c, MI_ICP_ERR_COMM, "mailbox: %d ranks (at most %d)", nranks, kMailRanks);
    const size_t bytes = (sizeof(MailBox) + 4095) / 4096 * 4096;
    const auto t0 = std::chrono::steady_clock::now();
    const auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(mail_attach_timeout_ms()); };
    MailBox* box = nullptr;
    void* dev = nullptr;
    auto drop = [&](void* p) {
        if (dev) (void)hipHostUnregister(p);
        dev = nullptr;
        (void)munmap(p, bytes);
    };
    auto map_fd = [&](int fd) -> void* {
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        return p == MAP_FAILED ? nullptr : p;
    };
    auto reg = [&](void* p) {
        if (hipHostRegister(p, bytes, hipHostRegisterMapped) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostUnregister(p);
            dev = nullptr;
            return false;
        }
        return true;
    };
    if (rank == 0) {
        (void)shm_unlink(name.c_str());  // a stale box of a crashed job
        int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
            if (fd >= 0) close(fd);
            (void)shm_unlink(name.c_str());
            return fail(c, MI_ICP_ERR_COMM, "mailbox: cannot create shared memory %s", name.c_str());
        }
        void* p = map_fd(fd);
        if (!p || !reg(p)) {
            if (p) (void)munmap(p, bytes);
            (void)shm_unlink(name.c_str());
            return fail(c, MI_ICP_ERR_COMM, "mailbox: cannot map / register shared memory %s", name.c_str());
        }
        box = (MailBox*)p;
        std::memset(p, 0, bytes);
        box->nranks = (uint32_t)nranks;
        // device inboxes (mailbox.h) are set up next to the box unless MI_ICP_MAILBOX=host says not to; they are USED
        // when MI_ICP_MAILBOX=device or mi_icp_comm_autotune finds them faster
        const char* mode = std::getenv("MI_ICP_MAILBOX");
        box->device_mode = (mode && std::strcmp(mode, "host") == 0) ? 0u : 1u;
        __atomic_store_n(&box->ready, 1u, __ATOMIC_RELEASE);
        while (__atomic_load_n(&box->attached, __ATOMIC_ACQUIRE) != (uint32_t)(nranks - 1)) {
            if (late()) {
                drop(p);
                (void)shm_unlink(name.c_str());
                return fail(c, MI_ICP_ERR_COMM, "mailbox: not all of the %d other ranks attached to %s in time", nranks - 1, name.c_str());
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        __atomic_store_n(&box->go, 1u, __ATOMIC_RELEASE);
        (void)shm_unlink(name.c_str());  // every rank holds its mapping: the name has done its job
    } else {
        for (;;) {
            if (late()) return fail(c, MI_ICP_ERR_COMM, "mailbox: no usable shared memory %s appeared in time", name.c_str());
            int fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
                if (fd >= 0) close(fd);
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                continue;
            }
            void* p = map_fd(fd);
            if (!p) return fail(c, MI_ICP_ERR_COMM, "mailbox: mmap failed");
            box = (MailBox*)p;
            bool usable = false;
            while (!late()) {
                if (__atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 0u) break;            // in use: not ours
                if (__atomic_load_n(&box->ready, __ATOMIC_ACQUIRE) == 1u) {
                    usable = true;
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (!usable || __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 0u) {
                (void)munmap(p, bytes);
                box = nullptr;
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                continue;
            }
            if (box->nranks != (uint32_t)nranks) {
                const uint32_t made_for = box->nranks;
                (void)munmap(p, bytes);
                return fail(c, MI_ICP_ERR_COMM, "mailbox: %s was made for %u ranks, not %d", name.c_str(), made_for, nranks);
            }
            if (!reg(p)) {
                (void)munmap(p, bytes);
                return fail(c, MI_ICP_ERR_COMM, "mailbox: hipHostRegister failed");
            }
            (void)__atomic_fetch_add(&box->attached, 1u, __ATOMIC_ACQ_REL);
            // While waiting for `go`: is the NAME still this box?  A crashed job's leftover (ready, never started) under a
            // reused name looks like ours; rank 0 replaces it (unlink + create), after which the name leads to another
            // inode -- this mapping is then dropped and the name opened again (the wait used to run into the
            // attach time-out, and rank 0's with it).
            bool replaced = false;
            for (int polls = 0; __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 1u; ++polls) {
                if (late()) {  // (e.g. the box was a crashed job's and rank 0 never came)
                    drop(p);
                    return fail(c, MI_ICP_ERR_COMM, "mailbox: rank 0 did not start %s in time", name.c_str());
                }
                if (polls % 100 == 99) {
                    struct stat now;
                    const int fd2 = shm_open(name.c_str(), O_RDWR, 0600);
                    const bool other = fd2 >= 0 && fstat(fd2, &now) == 0 && (now.st_ino != st.st_ino || now.st_dev != st.st_dev);
                    if (fd2 >= 0) close(fd2);
                    if (other && __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 1u) {
                        replaced = true;
                        break;
                    }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (replaced) {
                drop(p);
                box = nullptr;
                continue;
            }
            break;
        }
    }
    uint32_t* state;
    TRY(ensure(c, c->mail_state, 64, &state));
    HIPCHK(c, hipMemsetAsync(state, 0, 64 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->mail_host = box;
    c->mail_dev = (MailBox*)dev;
    c->mail_bytes = bytes;
    c->mail_name = name;
    c->mail_linked = false;  // (rank 0 has removed the name already)
    c->comm_broken = false;
    c->nranks = nranks;
    c->rank = rank;
    if (box->device_mode) (void)inbox_open(c, box, nranks, rank, late);  // (failing that, on every rank alike: the box's own words)
    {
        const char* mode = std::getenv("MI_ICP_MAILBOX");
        c->xchg = (c->inbox && mode && std::strcmp(mode, "device") == 0) ? 2 : 1;
    }
    return MI_ICP_OK;
}

// A failed exchange leaves the ranks' exchange counters apart: whatever they post from now on could be taken
// for another exchange's.  The mailbox is given up and every call that would exchange fails until the
// communicator has been destroyed / initialised again.
int comm_failed(mi_icp_ctx* c, const char* what) {
    mailbox_close(c);
    c->comm_broken = true;
    c->loop_active = false;
    return fail(c, MI_ICP_ERR_COMM, "%s; the communicator is void: destroy it and initialise a new one", what);
}

int comm_usable(mi_icp_ctx* c) {
    if (c->comm_broken)
        return fail(c, MI_ICP_ERR_COMM, "the communicator is void after a failed exchange (timed out): destroy it and initialise a new one");
    return MI_ICP_OK;
}

int allreduce_system(mi_icp_ctx* c) {
    TRY(comm_usable(c));
    if (mail_on(c)) {  // one-shot exchange through the mailbox
        int32_t* state = (int32_t*)c->mail_state.p;
        mail_allreduce_kernel<<<1, 64, 0, c->stream>>>(mail_args(c), (double*)c->sys_dev.p, state + 1);
        KCHK(c);
        return MI_ICP_OK;
    }
    if (!c->comm) return MI_ICP_OK;
    double* sys = (double*)c->sys_dev.p;
    ncclResult_t r = g_rccl.AllReduce(sys, sys, kSysSize, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail(c, MI_ICP_ERR_COMM, "ncclAllReduce failed (%d)", (int)r);
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

int check_ctx(mi_icp_ctx* c) {
    if (!c) return MI_ICP_ERR_INVALID;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, MI_ICP_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
    return MI_ICP_OK;
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
// mi_icp_debug_solve_both: the step's serial and wave-wide solves on the same systems
namespace mi {
__global__ __launch_bounds__(64) void solve_both_kernel(const double* systems, float det_thresh, float* out_serial,
                                                        float* out_wave, int32_t* ok_serial, int32_t* ok_wave) {
    __shared__ double s_sys[32];
    const int n = (int)blockIdx.x;
    if (threadIdx.x < 32) s_sys[threadIdx.x] = systems[(int64_t)n * 32 + threadIdx.x];
    __syncthreads();
    host::Mat4 W;
    const bool okw = wave_solve_system(s_sys, det_thresh, W);
    if (threadIdx.x < 16) out_wave[(int64_t)n * 16 + threadIdx.x] = select16(W.m, (int)threadIdx.x);
    if (threadIdx.x == 0) {
        ok_wave[n] = okw ? 1 : 0;
        host::Mat4 S;
        ok_serial[n] = host::solve_system(s_sys, det_thresh, S) ? 1 : 0;
        for (int e = 0; e < 16; ++e) out_serial[(int64_t)n * 16 + e] = S.m[e];
    }
}
}  // namespace mi

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
              hipHostMalloc((void**)&c->u_host, 16 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
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
    mailbox_close(c);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    DevBuf* all[] = {&c->trec, &c->tidx, &c->thalo, &c->tlinks_tmp, &c->halo_want, &c->loop_hist, &c->tblk, &c->tnrm, &c->tcov, &c->tgrad, &c->sint, &c->nodes, &c->inv_t, &c->cell_planes, &c->cell_samples, &c->cell_cstart,
                     &c->cell_gstart, &c->sx, &c->sy, &c->sz,
                     &c->sperm, &c->snrm, &c->scov, &c->nn_idx, &c->nn_d2, &c->inv_s,
                     &c->user_pairs, &c->keys0, &c->keys1, &c->vals0, &c->vals1, &c->hist,
                     &c->scan_tmp, &c->bounds_part, &c->bounds, &c->partial, &c->sys_dev,
                     &c->dense_idx, &c->flags, &c->pairs_out, &c->seg_start, &c->loop_dev, &c->ticket, &c->mail_state, &c->alt[0],
                     &c->alt[1], &c->alt[2], &c->alt[3], &c->alt[4], &c->alt[5], &c->alt[6], &c->alt[7], &c->alt[8], &c->stage[0],
                     &c->stage[1], &c->stage[2], &c->stage[3], &c->stage[4], &c->stage[5], &c->knn_idx, &c->tscale, &c->vpay[0], &c->vpay[1],
                     &c->vpay[2], &c->vpay[3], &c->vpay[4], &c->vpay[5]};
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
    static const bool no_trec = std::getenv("MI_ICP_NO_TREC") != nullptr;  // A/B switch
    if (d_nrm) TRY(ensure(c, c->tnrm, (size_t)nts, &tnrm));
    if (d_nrm && !no_trec) TRY(ensure(c, c->trec, (size_t)nts * 6, &trec));
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
        static const float link_delta = [] { const char* e = std::getenv("MI_ICP_LINK_DELTA"); const float v = e ? (float)std::atof(e) : 0.0f; return v > 0.0f ? v : 0.25f; }();
        ga.link_delta = link_delta;
        static const float region_margin = [] { const char* e = std::getenv("MI_ICP_REGION_MARGIN"); const float v = e ? (float)std::atof(e) : 0.0f; return v > 0.0f ? v : 0.5f; }();
        ga.region_margin = region_margin;
        kd_build_groups<<<(unsigned)lay.ngroups, kKdThreads, 0, c->stream>>>(ga);
        KCHK(c);
        first = leaf_first >> 9;  // the groups' own boxes sit in the records of this level
        used = ((uint32_t)lay.ngroups + 7u) / 8u;
    }
    int above_groups = 1;  // 8-ary levels between `first` and the groups' level
    for (; first > 1u; first /= 8u, ++above_groups) {
        const uint32_t count = ((used + 7u) / 8u) * 8u;
        build_level<<<blocks_for(count), 256, 0, c->stream>>>(nodes, first, used, count, upper_flag, lay.planes,
                                                              lay.levels, lay.levels - 3 * above_groups);
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
    c->halo_iters = c->halo_asked = 0;
    c->links_allowed = !no_cells && (uint32_t)nleaf <= kLinkIdMask;
    c->nt = n;
    c->nts = nts;
    c->nleaf = nleaf;
    c->leaf_first = leaf_first;
    c->nrecords = nrecords;
    // A context that has run a registration loop will run another.  For a small target (frame-to-frame callers:
    // KinFu, odometry) the halos are started right away, on the private stream, next to the staging of the source:
    // they cost that little, and the loop's first seeded iterations find them ready.  For a large one the build
    // would fight the staging for the memory system; there the loop's own searches say whether it is wanted.
    if (c->ran_loop && c->links_allowed && n < kHaloAheadMax) TRY(start_links_async(c));
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
    // Measured: the in-group kd split on top of the Morton order (kd_refine.h) costs more here
    // (1.2 ms at 10M) than it saves in that one pass (0.15 ms); MI_ICP_SOURCE_KD=1 turns it on.
    static const bool source_kd = std::getenv("MI_ICP_SOURCE_KD") != nullptr;
    const uint32_t* order;
    TRY(morton_order(c, d_pts, n, &order, source_kd));

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

// ---- device-resident registration loop (loop.h) -------------------------------------------
// Re-order the staged source by its current matches (lbvh.h: match_order_keys).
// Enqueue-only; the second set of source arrays becomes the live one.
static int resort_source_by_match(mi_icp_ctx* c) {
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
static bool fused_iteration_applies(const mi_icp_ctx* c, bool seed) {
    static const bool off = std::getenv("MI_ICP_NO_FUSED_ITERATION") != nullptr;  // A/B switch
    static const int64_t limit = [] { const char* e = std::getenv("MI_ICP_FUSED_MAX"); return e ? std::atoll(e) : kFusedMax; }();
    return !off && seed && c->nn_valid && c->loop_est == kEstPt2Pl && estimator_ready(c, kEstPt2Pl) && c->t_has_rec &&
           c->trec.p != nullptr && !c->comm && !c->mail_dev && c->n_user_pairs < 0 && c->ns > 0 && c->ns <= limit &&
           c->nt > 0;
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
    icp_small_iteration_kernel<<<grid, kReduceThreads, 0, c->stream>>>(
            (const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, (int)c->ns, (const float*)c->nodes.p,
            (const float*)c->tblk.p, (const float*)lreg_of(c), have_halo ? (const float*)c->thalo.p : nullptr, c->leaf_first,
            c->loop_r2, npackets, nblocks, (int32_t*)c->nn_idx.p, want, (const float*)c->trec.p, d, partial, (uint32_t*)c->ticket.p, sys);
    KCHK(c);
    c->last_search_kind = 1;
    return MI_ICP_OK;
}

// Mid-sized sources (a rank's share of a sharded registration): search + rows + reduction + exchange + step in ONE
// launch with the sums kept in registers across a wave's packets (fused_small.h, icp_mid_iteration_kernel).
// MI_ICP_MID_MAX: the largest source that takes it (0: never).
constexpr int64_t kMidMax = 0;  // (set from the measurements: DESIGN section 5 / EXPERIMENTS.md)
static bool mid_iteration_applies(const mi_icp_ctx* c, bool seed) {
    static const int64_t limit = [] { const char* e = std::getenv("MI_ICP_MID_MAX"); return e ? std::atoll(e) : kMidMax; }();
    static const int64_t fused_limit = [] { const char* e = std::getenv("MI_ICP_FUSED_MAX"); return e ? std::atoll(e) : kFusedMax; }();
    return limit > 0 && seed && c->nn_valid && c->loop_est == kEstPt2Pl && estimator_ready(c, kEstPt2Pl) && c->t_has_rec &&
           c->trec.p != nullptr && c->n_user_pairs < 0 && c->ns > fused_limit && c->ns <= limit && c->nt > 0;
}

static int launch_mid_iteration(mi_icp_ctx* c, DevLoop* d, bool* stepped) {
    const bool have_halo = c->halo_use;
    uint32_t* want = (!have_halo && !c->links_inflight && c->links_allowed) ? (uint32_t*)c->halo_want.p : nullptr;
    const uint32_t npackets = (uint32_t)((c->ns + 63) / 64);
    // one round of waves at the kernel's occupancy: 4 per SIMD, 16 per CU (MI_ICP_MID_WAVES: waves per CU aimed at)
    static const uint32_t waves_per_cu = [] { const char* e = std::getenv("MI_ICP_MID_WAVES"); const int v = e ? std::atoi(e) : 0; return (uint32_t)(v > 0 ? v : 16); }();
    static const uint32_t ncu = [] { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); return (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? (uint32_t)p.multiProcessorCount : 256u; }();
    const uint32_t target_waves = waves_per_cu * ncu;
    const uint32_t ppw = std::max(1u, (npackets + target_waves - 1) / target_waves);
    const uint32_t nwaves = (npackets + ppw - 1) / ppw;
    const uint32_t nblocks = (nwaves + kFusedPackets - 1) / kFusedPackets;
    const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
    double *partial, *sys;
    TRY(ensure(c, c->partial, (size_t)std::max<uint32_t>(kReduceBlocks, grid) * kSysSize, &partial));
    TRY(ensure(c, c->sys_dev, kSysSize, &sys));
    if (!c->ticket.p) {
        uint32_t* ticket;
        TRY(ensure(c, c->ticket, 64, &ticket));
        HIPCHK(c, hipMemsetAsync(ticket, 0, 256, c->stream));
    }
    const MailArgs no_mail = {nullptr, nullptr, 0, 1, 0u, nullptr, nullptr};
    EvTimer t(c, 0, true);
#define MI_MID_ARGS (const float*)c->sx.p, (const float*)c->sy.p, (const float*)c->sz.p, (int)c->ns, (const float*)c->nodes.p, \
            (const float*)c->tblk.p, (const float*)lreg_of(c), have_halo ? (const float*)c->thalo.p : nullptr, c->leaf_first, \
            c->loop_r2, npackets, ppw, nblocks, (int32_t*)c->nn_idx.p, want, (const float*)c->trec.p, d, partial, (uint32_t*)c->ticket.p, sys
    if (mail_on(c)) {
        icp_mid_iteration_kernel<2><<<grid, kReduceThreads, 0, c->stream>>>(MI_MID_ARGS, mail_args(c));
        *stepped = true;
    } else if (!c->comm && !c->mail_dev) {
        icp_mid_iteration_kernel<1><<<grid, kReduceThreads, 0, c->stream>>>(MI_MID_ARGS, no_mail);
        *stepped = true;
    } else {
        icp_mid_iteration_kernel<0><<<grid, kReduceThreads, 0, c->stream>>>(MI_MID_ARGS, no_mail);
        *stepped = false;
    }
#undef MI_MID_ARGS
    KCHK(c);
    c->last_search_kind = 1;
    return MI_ICP_OK;
}

// one evaluation: search under the loop's transform, reduction, all-reduce, step kernel
static int loop_enqueue_evaluation(mi_icp_ctx* c, bool seed) {
    DevLoop* d = (DevLoop*)c->loop_dev.p;
    if (fused_iteration_applies(c, seed)) return launch_fused_iteration(c, d);
    if (mid_iteration_applies(c, seed)) {
        bool mid_stepped = false;
        TRY(launch_mid_iteration(c, d, &mid_stepped));
        if (mid_stepped) return MI_ICP_OK;
        TRY(allreduce_system(c));  // (RCCL between the sums and the step)
        loop_step_kernel<<<1, kStepThreads, 0, c->stream>>>(d, (double*)c->sys_dev.p, 0, MailArgs{nullptr, nullptr, 0, 1, 0u, nullptr, nullptr});
        KCHK(c);
        return MI_ICP_OK;
    }
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
static int loop_run(mi_icp_ctx* c, int budget) {
    constexpr int kChunk = 8;
    constexpr int64_t kLarge = 500000, kHaloLongRun = 40;
    while (budget > 0) {
        const bool no_halo = !c->links_ready && !c->links_inflight && c->links_allowed && c->nt > 0;
        const bool undecided = no_halo && !c->halo_declined;
        // (a build the loop's decision started -- or one started with the loop on a context that has asked
        // before, or behind a small target's tree: the stream waits for what is left of it rather than walk)
        if (c->links_inflight && !c->halo_declined) TRY(ensure_links(c));
        // (a short remainder rides along: one host synchronisation less than it would cost)
        int n = (budget <= kChunk + kChunk / 2) ? budget : kChunk;
        // (with several ranks the chunking must not depend on anything a rank sees alone: every rank has to
        // enqueue the same number of evaluations -- an in-library RCCL all-reduce is a host-side call per
        // evaluation, and a rank that stops at `done` after fewer of them would leave its peers' calls unmatched)
        const bool several_ranks = c->comm != nullptr || c->mail_dev != nullptr;
        if (undecided && c->ns >= kLarge && !several_ranks) n = 1;
        const int passes_before = c->loop_host->passes;
        c->halo_use = halo_poll(c);
        for (int i = 0; i < n; ++i) TRY(loop_enqueue_evaluation(c, true));
        if (no_halo) HIPCHK(c, hipMemcpyAsync(c->u_host + 8, c->halo_want.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        TRY(loop_pull(c));
        const int executed = c->loop_host->passes - passes_before;
        collect_pooled(c, executed);
        budget -= n;
        c->halo_iters += executed;
        if (no_halo) {
            // (a 32-bit device counter that keeps counting through a long stepping loop: the difference is taken
            // modulo 2^32, so a wrap between two looks costs nothing)
            const int64_t asked = (int64_t)(uint32_t)(c->u_host[8] - (uint32_t)c->halo_want_seen);  // by this chunk's iterations
            c->halo_want_seen = (int64_t)c->u_host[8];
            c->halo_asked += asked;
            if (undecided) {
                ++c->halo_looks;
                const int64_t per = std::max(executed, 1);
                const bool many = asked * 32 > c->ns * per, most = asked * 5 > 2 * c->ns * per;
                if (!many) {
                    c->halo_declined = true;
                } else if (most || c->halo_looks >= 2 || c->ns < kLarge) {
                    c->halo_sticky = true;
                    TRY(start_links_async(c));
                }
            } else if (c->halo_iters >= kHaloLongRun && c->halo_asked > 0) {
                TRY(start_links_async(c));  // (in the background: halo_declined stays, nothing waits)
            }
        }
        if (c->loop_host->done) break;
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
    c->ran_loop = true;
    {
        uint32_t* want;
        TRY(ensure(c, c->halo_want, 64, &want));
        HIPCHK(c, hipMemsetAsync(want, 0, sizeof(uint32_t), c->stream));
    }
    c->halo_use = halo_poll(c);
    // A context whose loops have asked for halos before builds them NOW, on the loop's own stream, ahead of the first
    // pass -- which then starts from the queries' own seeds (launch_nn).  (Round 3 started the build on the private
    // stream next to the first pass and the match-order re-sort: the 2.5-ms build and those streaming kernels fought
    // for the memory system -- match_order_keys 28 us alone, 1.7 ms beside leaf_halo_build; leaf_halo_collect 0.73 ->
    // 1.7 ms -- and the loop waited for the build at its first seeded iteration anyway.  MI_ICP_LINKS_ASYNC=1: that form.)
    if (c->halo_sticky && !c->halo_use) {
        static const bool links_async = std::getenv("MI_ICP_LINKS_ASYNC") != nullptr;  // A/B switch
        if (links_async || c->links_inflight) {
            TRY(start_links_async(c));
        } else {
            TRY(ensure_links(c));
            c->halo_use = halo_poll(c);
        }
    }
    TRY(loop_enqueue_evaluation(c, false));
    // from here on the packets follow the target's order (pays for itself in ~4 iterations)
    static const bool no_resort = std::getenv("MI_ICP_NO_RESORT") != nullptr;  // A/B switch for tuning
    if (!no_resort && c->ns >= 32768 && (max_iterations >= 4 || max_iterations == 0))
        TRY(resort_source_by_match(c));
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

// ---------------------------------------------------------------------------
int mi_icp_transform(mi_icp_ctx* c, const float* T, float* xyz, float* normals, float* covs,
                     int64_t n, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0) return fail(c, MI_ICP_ERR_INVALID, "transform: negative size");
    if (n == 0 || (!xyz && !normals && !covs)) return MI_ICP_OK;
    const Xform X = make_xform(load_T(T));
    const float *dp, *dn, *dc;
    TRY(to_device(c, (const float*)xyz, (size_t)n * 3, mem_kind, c->stage[0], &dp));
    TRY(to_device(c, (const float*)normals, (size_t)n * 3, mem_kind, c->stage[1], &dn));
    TRY(to_device(c, (const float*)covs, (size_t)n * 9, mem_kind, c->stage[2], &dc));
    transform_cloud<<<blocks_for(n), 256, 0, c->stream>>>(X, (float*)dp, (float*)dn, (float*)dc, n);
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, dp, xyz, xyz ? (size_t)n * 3 : 0, mem_kind));
        TRY(from_device(c, dn, normals, normals ? (size_t)n * 3 : 0, mem_kind));
        TRY(from_device(c, dc, covs, covs ? (size_t)n * 9 : 0, mem_kind));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // pointcloud.cu:297 cudaDeviceSynchronize
    return MI_ICP_OK;
}

// GeometryBase3D::GetMinBound / GetMaxBound / GetCenter (geometry/pointcloud.cu:205-215)
int mi_icp_compute_bounds(mi_icp_ctx* c, const float* xyz, int64_t n, int mem_kind, float* min3, float* max3,
                          float* center3) {
    TRY(check_ctx(c));
    if (n < 0 || (n > 0 && !xyz)) return fail(c, MI_ICP_ERR_INVALID, "compute_bounds: bad size/pointer");
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    if (n == 0) {  // the reference returns zero vectors for an empty cloud
        if (min3) std::memcpy(min3, zero, sizeof(zero));
        if (max3) std::memcpy(max3, zero, sizeof(zero));
        if (center3) std::memcpy(center3, zero, sizeof(zero));
        return MI_ICP_OK;
    }
    const float* d_pts;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[0], &d_pts));
    float* bnd;
    TRY(compute_bounds(c, d_pts, n, &bnd));  // min[3], max[3], extent
    float* rec;
    TRY(ensure(c, c->flags, 16, &rec));
    HIPCHK(c, hipMemcpyAsync(rec, bnd, 7 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (center3) {
        double* part;
        TRY(ensure(c, c->partial, (size_t)kReduceBlocks * kSysSize, &part));
        const int blocks = (int)std::min<int64_t>(kCenterBlocks, blocks_for(n));
        center_partial<<<blocks, 256, 0, c->stream>>>(d_pts, n, part);
        KCHK(c);
        center_final<<<1, 64, 0, c->stream>>>(part, blocks, n, rec);
        KCHK(c);
    }
    HIPCHK(c, hipMemcpyAsync(c->f_host, rec, 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (min3) std::memcpy(min3, c->f_host, 3 * sizeof(float));
    if (max3) std::memcpy(max3, c->f_host + 3, 3 * sizeof(float));
    if (center3) std::memcpy(center3, c->f_host + 7, 3 * sizeof(float));
    return MI_ICP_OK;
}

// GeometryBase3D::Translate / Scale / Rotate (geometry/pointcloud.cu:225-242)
int mi_icp_affine(mi_icp_ctx* c, const float* R9, float scale, int use_scale, const float* center3,
                  const float* translate3, float* xyz, float* normals, float* covs, int64_t n, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0) return fail(c, MI_ICP_ERR_INVALID, "affine: negative size");
    if (n == 0 || (!xyz && !normals && !covs)) return MI_ICP_OK;
    Affine A;
    std::memset(&A, 0, sizeof(A));
    A.use_r = R9 != nullptr;
    A.use_s = use_scale != 0;
    A.use_c = center3 != nullptr;
    A.use_t = translate3 != nullptr;
    A.s = scale;
    if (R9)   // column-major (Eigen::Matrix3f::data()) -> row-major
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) A.r[r * 3 + q] = R9[q * 3 + r];
    if (center3) std::memcpy(A.c, center3, sizeof(A.c));
    if (translate3) std::memcpy(A.t, translate3, sizeof(A.t));
    const float *dp, *dn, *dc;
    TRY(to_device(c, (const float*)xyz, (size_t)n * 3, mem_kind, c->stage[0], &dp));
    TRY(to_device(c, (const float*)normals, (size_t)n * 3, mem_kind, c->stage[1], &dn));
    TRY(to_device(c, (const float*)covs, (size_t)n * 9, mem_kind, c->stage[2], &dc));
    affine_cloud<<<blocks_for(n), 256, 0, c->stream>>>(A, const_cast<float*>(dp), const_cast<float*>(dn),
                                                        const_cast<float*>(dc), n);
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, dp, xyz, (size_t)n * 3, mem_kind));
        TRY(from_device(c, dn, normals, (size_t)n * 3, mem_kind));
        TRY(from_device(c, dc, covs, (size_t)n * 9, mem_kind));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MI_ICP_OK;
}

int mi_icp_covariances_from_normals(mi_icp_ctx* c, const float* normals, int64_t n, float epsilon,
                                    float* covs, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0 || (n > 0 && (!normals || !covs))) return fail(c, MI_ICP_ERR_INVALID, "covariances_from_normals: bad arguments");
    if (n == 0) return MI_ICP_OK;
    const float* dn;
    TRY(to_device(c, normals, (size_t)n * 3, mem_kind, c->stage[1], &dn));
    float* dc = covs;
    if (mem_kind == MI_ICP_HOST) TRY(ensure(c, c->stage[2], (size_t)n * 9, &dc));
    cov_from_normals<<<blocks_for(n), 256, 0, c->stream>>>(dn, n, epsilon, dc);
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) TRY(from_device(c, (const float*)dc, covs, (size_t)n * 9, mem_kind));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

// VoxelDownSample for grids whose packed (x, y, z) key fits 32 bits (geometry_kernels.h, "the path for grids ..."):
// keys -> radix passes on the bits above the lowest L that carry the payload -> runs of equal key >> L -> which voxels
// occur in each run -> their output positions -> means.  Two host synchronisations in the whole call (the bounds that
// place the grid, the voxel count that sizes the output), as before.
static int voxel_downsample_keys32(mi_icp_ctx* c, const float* dp, const float* dn, const float* dcol, int64_t n,
                                   const VoxelGrid& g, int bits, float* out_xyz, float* out_normals, float* out_colors,
                                   int64_t* m, int mem_kind) {
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    uint32_t* const keys[2] = {reinterpret_cast<uint32_t*>(sb.keys[0]), reinterpret_cast<uint32_t*>(sb.keys[1])};
    voxel_keys32<<<blocks_for(n), 256, 0, c->stream>>>(dp, n, g, keys[0]);
    KCHK(c);
    // the lowest L <= 5 key bits stay unsorted where that saves a pass (21 bits: 2 passes, L = 5; 24 bits: 3, L = 0)
    int passes = std::max(0, (bits - 5 + 7) / 8);
    int L = std::min(5, std::max(0, bits - 8 * passes));
    // ... but only where runs are long enough to give a wave work: with more possible runs than an eighth of the points
    // (a fine grid over a sparse cloud: most runs a point or two) the key is sorted whole and 8 lanes take a voxel
    if (L > 0 && (bits - L >= 31 || ((int64_t)1 << (bits - L)) > n / 8)) {
        L = 0;
        passes = (bits + 7) / 8;
    }
    const Pay3* first[3] = {reinterpret_cast<const Pay3*>(dp), reinterpret_cast<const Pay3*>(dn), reinterpret_cast<const Pay3*>(dcol)};
    Pay3* scratch[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    for (int set = 0; set < std::min(passes, 2); ++set)
        for (int a = 0; a < 3; ++a)
            if (first[a]) TRY(ensure(c, c->vpay[set * 3 + a], (size_t)n, &scratch[set][a]));
    const Pay3* pay[3];
    const int cur = radix_sort_payload32(c->stream, keys, first, scratch, sb.hist, sb.scan_tmp, n, L, bits, pay);
    KCHK(c);
    const uint32_t* skeys = keys[cur];
    // runs of equal key >> L
    const int ntiles = scan_num_tiles(n);
    uint32_t *run_start, *mask = nullptr, *voff = nullptr, *tmp = sb.scan_tmp;
    TRY(ensure(c, c->seg_start, (size_t)n + 4, &run_start));
    vox_head_sums<<<ntiles, kScanThreads, 0, c->stream>>>(skeys, (int)n, L, tmp);
    scan_tile_offsets<<<1, kScanThreads, 0, c->stream>>>(tmp, ntiles);
    vox_head_apply<<<ntiles, kScanThreads, 0, c->stream>>>(skeys, (int)n, L, tmp, ntiles, run_start);
    KCHK(c);
    uint32_t* nruns = run_start + n + 2;  // (R, written by vox_head_apply; kept apart: the scan below reuses tmp)
    const uint32_t* total = nruns;
    if (L > 0) {
        const int64_t rmax = (bits - L >= 31) ? n : std::min<int64_t>(n, (int64_t)1 << (bits - L));
        TRY(ensure(c, c->flags, (size_t)n, &mask));
        TRY(ensure(c, c->dense_idx, (size_t)n, &voff));
        vox_run_masks<<<blocks_for(rmax * 16), 256, 0, c->stream>>>(skeys, run_start, nruns, rmax, L, mask, voff);
        KCHK(c);
        exclusive_scan_u32(c->stream, voff, voff, rmax, tmp);
        KCHK(c);
        total = tmp + scan_num_tiles(rmax);
    }
    HIPCHK(c, hipMemcpyAsync(c->u_host, total, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t nvox = (int64_t)c->u_host[0];
    float *op = out_xyz, *on = out_normals, *oc = out_colors;
    if (mem_kind == MI_ICP_HOST) {
        TRY(ensure(c, c->stage[3], (size_t)nvox * 3, &op));
        if (dn) TRY(ensure(c, c->stage[4], (size_t)nvox * 3, &on));
        if (dcol) TRY(ensure(c, c->stage[5], (size_t)nvox * 3, &oc));
    }
    if (L > 0) {  // a wave per run
        const int64_t rmax = (bits - L >= 31) ? n : std::min<int64_t>(n, (int64_t)1 << (bits - L));
        voxel_means_wave<<<(unsigned)rmax, 64, 0, c->stream>>>(skeys, pay[0], pay[1], pay[2], run_start, voff, mask, nruns, rmax, L,
                                                                      op, dn ? on : nullptr, dcol ? oc : nullptr);
    } else {      // a run is a voxel: 8 lanes each
        voxel_means_runs<<<blocks_for(nvox * 8), 256, 0, c->stream>>>(skeys, pay[0], pay[1], pay[2], run_start, voff, mask, nruns, L,
                                                                     nvox, op, dn ? on : nullptr, dcol ? oc : nullptr);
    }
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, (const float*)op, out_xyz, (size_t)nvox * 3, mem_kind));
        if (dn) TRY(from_device(c, (const float*)on, out_normals, (size_t)nvox * 3, mem_kind));
        if (dcol) TRY(from_device(c, (const float*)oc, out_colors, (size_t)nvox * 3, mem_kind));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *m = nvox;
    return MI_ICP_OK;
}

int mi_icp_voxel_downsample(mi_icp_ctx* c, const float* xyz, const float* normals,
                            const float* colors, int64_t n, float voxel, float* out_xyz,
                            float* out_normals, float* out_colors, int64_t* m, int mem_kind) {
    TRY(check_ctx(c));
    if (!m) return fail(c, MI_ICP_ERR_INVALID, "voxel_downsample: m is null");
    *m = 0;
    if (n < 0 || n > 0x7fffff00ll) return fail(c, MI_ICP_ERR_INVALID, "voxel_downsample: bad size");
    if (n == 0 || !(voxel > 0.0f)) return MI_ICP_OK;  // down_sample.cu:173-176
    if (!xyz || !out_xyz || (normals && !out_normals) || (colors && !out_colors))
        return fail(c, MI_ICP_ERR_INVALID, "voxel_downsample: null buffer");

    const float *dp, *dn, *dcol;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[0], &dp));
    TRY(to_device(c, normals, (size_t)n * 3, mem_kind, c->stage[1], &dn));
    TRY(to_device(c, colors, (size_t)n * 3, mem_kind, c->stage[2], &dcol));

    float* bnd;
    TRY(compute_bounds(c, dp, n, &bnd));
    HIPCHK(c, hipMemcpyAsync(c->f_host, bnd, 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    VoxelGrid g;
    float ext = 0.0f;
    int bits[3];
    {
        const float* b = c->f_host;
        const float origin[3] = {b[0] - voxel * 0.5f, b[1] - voxel * 0.5f, b[2] - voxel * 0.5f};
        for (int d = 0; d < 3; ++d) ext = std::fmax(ext, (b[3 + d] + voxel * 0.5f) - origin[d]);
        if (voxel * (float)INT32_MAX < ext) return MI_ICP_OK;  // down_sample.cu:186-189
        g.ox = origin[0];
        g.oy = origin[1];
        g.oz = origin[2];
        g.voxel = voxel;
        for (int d = 0; d < 3; ++d) {
            const double cells = std::floor(((double)b[3 + d] - (double)origin[d]) / (double)voxel) + 2.0;
            int nb = 1;
            while (nb < 32 && (double)(1ull << nb) < cells) ++nb;
            bits[d] = nb;
        }
        g.bits_y = bits[1];
        g.bits_z = bits[2];
    }

    static const bool old_voxel = std::getenv("MI_ICP_VOXEL_OLD") != nullptr;  // A/B switch: the (64-bit key, index) sort + gather
    if (bits[0] + bits[1] + bits[2] <= 32 && !old_voxel)
        return voxel_downsample_keys32(c, dp, dn, dcol, n, g, bits[0] + bits[1] + bits[2], out_xyz, out_normals, out_colors, m,
                                       mem_kind);

    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    const uint32_t* order;
    const uint64_t* packed_sorted = nullptr;  // sorted voxel keys when one key identifies the voxel
    const int nb = blocks_for(n);
    if (bits[0] + bits[1] + bits[2] <= 64) {
        voxel_keys<<<nb, 256, 0, c->stream>>>(dp, n, g, -1, nullptr, sb.keys[0], sb.vals[0]);
        KCHK(c);
        const int cur = radix_sort_pairs(c->stream, sb, n, bits[0] + bits[1] + bits[2]);
        order = sb.vals[cur];
        packed_sorted = sb.keys[cur];
    } else {
        // three stable sorts, least significant axis first
        const uint32_t* prev = nullptr;
        for (int axis = 2; axis >= 0; --axis) {
            uint32_t* tmp_order = nullptr;
            if (prev) {  // keys are rebuilt from the current order; keep it out of the sort's way
                TRY(ensure(c, c->seg_start, (size_t)n + 1, &tmp_order));
                HIPCHK(c, hipMemcpyAsync(tmp_order, prev, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            }
            voxel_keys<<<nb, 256, 0, c->stream>>>(dp, n, g, axis, tmp_order, sb.keys[0], sb.vals[0]);
            KCHK(c);
            prev = sb.vals[radix_sort_pairs(c->stream, sb, n, bits[axis])];
            if (prev != sb.vals[0] && axis > 0) {
                // next round writes keys[0]/vals[0]; the result already sits in the other pair
            }
        }
        order = prev;
    }
    KCHK(c);

    uint32_t *head, *pos, *seg_start, *tmp;
    TRY(ensure(c, c->flags, (size_t)n, &head));
    TRY(ensure(c, c->dense_idx, (size_t)n, (uint32_t**)&pos));
    // `order` may live in seg_start's buffer only in the fallback's intermediate rounds, never at the end
    TRY(ensure(c, c->scan_tmp, (size_t)scan_num_tiles(n) + 2, &tmp));
    if (packed_sorted) voxel_heads_keys<<<nb, 256, 0, c->stream>>>(packed_sorted, n, head);
    else voxel_heads<<<nb, 256, 0, c->stream>>>(dp, n, g, order, head);
    KCHK(c);
    exclusive_scan_u32(c->stream, head, pos, n, tmp);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(c->u_host, tmp + scan_num_tiles(n), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t nvox = (int64_t)c->u_host[0];
    TRY(ensure(c, c->seg_start, (size_t)n + 1, &seg_start));
    voxel_seg_starts<<<nb, 256, 0, c->stream>>>(head, pos, n, seg_start);
    KCHK(c);

    float *op = out_xyz, *on = out_normals, *oc = out_colors;
    if (mem_kind == MI_ICP_HOST) {
        TRY(ensure(c, c->stage[3], (size_t)nvox * 3, &op));
        if (dn) TRY(ensure(c, c->stage[4], (size_t)nvox * 3, &on));
        if (dcol) TRY(ensure(c, c->stage[5], (size_t)nvox * 3, &oc));
    }
    voxel_means<<<blocks_for(nvox * 8), 256, 0, c->stream>>>(dp, dn, dcol, order, seg_start, nvox, n, op,
                                                            dn ? on : nullptr, dcol ? oc : nullptr);
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, (const float*)op, out_xyz, (size_t)nvox * 3, mem_kind));
        if (dn) TRY(from_device(c, (const float*)on, out_normals, (size_t)nvox * 3, mem_kind));
        if (dcol) TRY(from_device(c, (const float*)oc, out_colors, (size_t)nvox * 3, mem_kind));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *m = nvox;
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
// PointCloud::CreateFromDepthImage / CreateFromRGBDImage (geometry/pointcloud_factory.cu)
static bool invert4(const float* M, float* out) {  // column-major general inverse, in double
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) {
            a[r][k] = (double)M[k * 4 + r];
            a[r][4 + k] = (r == k) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 0.0)) return false;
        if (piv != col)
            for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[col][k]);
        const double d = a[col][col];
        for (int k = 0; k < 8; ++k) a[col][k] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0)
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[col][k];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) out[k * 4 + r] = (float)a[r][4 + k];
    return true;
}

int mi_icp_create_from_depth(mi_icp_ctx* c, const void* depth, int depth_type, const void* color, int color_type,
                             int width, int height, const float* intrinsic4, const float* extrinsic,
                             float depth_scale, float depth_trunc, float depth_cutoff, int stride, int rgbd,
                             int compute_normals, int valid_only, float* out_xyz, float* out_normals,
                             float* out_colors, int64_t* m, int mem_kind) {
    TRY(check_ctx(c));
    if (!m) return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: m is null");
    *m = 0;
    if (width < 0 || height < 0 || stride < 1 || !intrinsic4 || (depth_type != MI_ICP_DEPTH_F32 && depth_type != MI_ICP_DEPTH_U16) ||
        (color_type != MI_ICP_COLOR_NONE && color_type != MI_ICP_COLOR_U8X3 && color_type != MI_ICP_COLOR_F32X1))
        return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: bad arguments");
    if (rgbd && (stride != 1 || depth_type != MI_ICP_DEPTH_F32))
        return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: an RGB-D image has a float depth and stride 1");
    if (!rgbd && (color || compute_normals || !valid_only))
        return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: colours, normals and valid_only = 0 belong to the RGB-D form");
    if ((color != nullptr) != (color_type != MI_ICP_COLOR_NONE))
        return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: color and color_type disagree");
    const int64_t npix = (int64_t)width * height;
    const int64_t count = (int64_t)(width / stride) * (height / stride);
    if (npix > 0x7fffff00ll) return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: image too large");
    if (count == 0) return MI_ICP_OK;
    if (!depth || !out_xyz || (color && !out_colors) || (compute_normals && !out_normals))
        return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: null buffer");

    DepthArgs a;
    const size_t dbytes = (size_t)npix * (depth_type == MI_ICP_DEPTH_U16 ? 2 : 4);
    const size_t cbytes = color ? (size_t)npix * (color_type == MI_ICP_COLOR_U8X3 ? 3 : 4) : 0;
    const uint8_t *dd, *dc;
    TRY(to_device(c, (const uint8_t*)depth, dbytes, mem_kind, c->stage[0], &dd));
    TRY(to_device(c, (const uint8_t*)color, cbytes, mem_kind, c->stage[1], &dc));
    a.depth = dd;
    a.color = dc;
    a.width = width;
    a.height = height;
    a.stride = stride;
    a.depth_u16 = depth_type == MI_ICP_DEPTH_U16;
    a.color_kind = color_type;
    a.rgbd = rgbd ? 1 : 0;
    a.depth_scale = (int)depth_scale;  // image.cu:340-343 holds both as int
    a.depth_trunc = (int)depth_trunc;
    a.depth_cutoff = depth_cutoff;
    a.fx = intrinsic4[0];
    a.fy = intrinsic4[1];
    a.cx = intrinsic4[2];
    a.cy = intrinsic4[3];
    const Mat4 E = load_T(extrinsic);
    if (!invert4(E.data(), a.pose)) return fail(c, MI_ICP_ERR_INVALID, "create_from_depth: singular extrinsic");

    const int nb = blocks_for(count);
    uint32_t* pos = nullptr;
    int64_t kept = count;
    if (valid_only) {
        uint32_t* tmp;
        TRY(ensure(c, c->flags, (size_t)count, &pos));
        TRY(ensure(c, c->scan_tmp, (size_t)scan_num_tiles(count) + 2, &tmp));
        depth_valid_flags<<<nb, 256, 0, c->stream>>>(a, count, pos);
        KCHK(c);
        exclusive_scan_u32(c->stream, pos, pos, count, tmp);
        KCHK(c);
        HIPCHK(c, hipMemcpyAsync(c->u_host, tmp + scan_num_tiles(count), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        kept = (int64_t)c->u_host[0];
    }
    float *op = out_xyz, *on = compute_normals ? out_normals : nullptr, *oc = color ? out_colors : nullptr;
    if (mem_kind == MI_ICP_HOST) {
        TRY(ensure(c, c->stage[3], (size_t)count * 3, &op));
        if (on) TRY(ensure(c, c->stage[4], (size_t)count * 3, &on));
        if (oc) TRY(ensure(c, c->stage[5], (size_t)count * 3, &oc));
    }
    depth_emit<<<nb, 256, 0, c->stream>>>(a, count, pos, op, on, oc);
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, (const float*)op, out_xyz, (size_t)kept * 3, mem_kind));
        if (on) TRY(from_device(c, (const float*)on, out_normals, (size_t)kept * 3, mem_kind));
        if (oc) TRY(from_device(c, (const float*)oc, out_colors, (size_t)kept * 3, mem_kind));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *m = kept;
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
// odometry::ComputeRGBDOdometry (odometry/odometry.cu); helpers above the extern "C" block
static int rgbd_odometry_impl(mi_icp_ctx* c, const float* source_color, const float* source_depth,
                              const float* target_color, const float* target_depth, int width, int height,
                              const float* intrinsic4, const float* odo_init, int jacobian,
                              const mi_icp_odometry_option* option, int* success, float* transformation16,
                              double* information36, int mem_kind, bool weighted, const float* prev_twist6,
                              float* twist6) {
    TRY(check_ctx(c));
    if (twist6)
        for (int i = 0; i < 6; ++i) twist6[i] = 0.0f;
    if (!success || !transformation16 || !information36 || !intrinsic4 || !option)
        return fail(c, MI_ICP_ERR_INVALID, "compute_rgbd_odometry: null argument");
    *success = 0;
    const Mat4 I4 = host::identity4();
    std::memcpy(transformation16, I4.data(), 16 * sizeof(float));
    for (int i = 0; i < 36; ++i) information36[i] = (i % 7 == 0) ? 1.0 : 0.0;
    if (width <= 0 || height <= 0 || (int64_t)width * height > 0x3fffffffll || !source_color || !source_depth ||
        !target_color || !target_depth)
        return fail(c, MI_ICP_ERR_INVALID, "compute_rgbd_odometry: bad image arguments");
    if (jacobian != MI_ICP_ODOMETRY_COLOR_TERM && jacobian != MI_ICP_ODOMETRY_HYBRID_TERM)
        return fail(c, MI_ICP_ERR_INVALID, "compute_rgbd_odometry: unknown jacobian type %d", jacobian);
    const int L = option->num_levels;
    if (L < 1 || L > MI_ICP_ODOMETRY_MAX_LEVELS || (width >> (L - 1)) < 1 || (height >> (L - 1)) < 1)
        return fail(c, MI_ICP_ERR_INVALID, "compute_rgbd_odometry: bad number of pyramid levels");

    const int64_t n0 = (int64_t)width * height;
    const float *in_sc, *in_sd, *in_tc, *in_td;
    TRY(to_device(c, source_color, (size_t)n0, mem_kind, c->stage[0], &in_sc));
    TRY(to_device(c, source_depth, (size_t)n0, mem_kind, c->stage[1], &in_sd));
    TRY(to_device(c, target_color, (size_t)n0, mem_kind, c->stage[2], &in_tc));
    TRY(to_device(c, target_depth, (size_t)n0, mem_kind, c->stage[3], &in_td));

    // one arena: per level colour + depth of both frames, a scratch image, and (target) 4 gradient images
    int lw[MI_ICP_ODOMETRY_MAX_LEVELS], lh[MI_ICP_ODOMETRY_MAX_LEVELS];
    size_t total = 0;
    for (int l = 0; l < L; ++l) {
        lw[l] = l ? lw[l - 1] / 2 : width;
        lh[l] = l ? lh[l - 1] / 2 : height;
        total += (size_t)lw[l] * lh[l] * 8;
    }
    total += (size_t)n0 + 64;
    float* arena;
    TRY(ensure(c, c->stage[4], total, &arena));
    double* sums;
    TRY(ensure(c, c->sys_dev, kSysSize, &sums));
    float *col[2][MI_ICP_ODOMETRY_MAX_LEVELS], *dep[2][MI_ICP_ODOMETRY_MAX_LEVELS], *grad[4][MI_ICP_ODOMETRY_MAX_LEVELS];
    {
        float* p = arena;
        for (int l = 0; l < L; ++l) {
            const size_t n = (size_t)lw[l] * lh[l];
            for (int s = 0; s < 2; ++s) {
                col[s][l] = p;
                p += n;
                dep[s][l] = p;
                p += n;
            }
            for (int g = 0; g < 4; ++g) {
                grad[g][l] = p;
                p += n;
            }
        }
    }
    float* scratch = arena + (total - (size_t)n0 - 64);
    auto blocks = [](int64_t n) { return (int)((n + kOdThreads - 1) / kOdThreads); };

    // ---- InitializeRGBDOdometry (odometry.cu:498-528)
    for (int s = 0; s < 2; ++s) {
        od_filter3<0, false><<<blocks(n0), kOdThreads, 0, c->stream>>>(s ? in_tc : in_sc, width, height, col[s][0], 0.0f, 0.0f);
        od_filter3<0, true><<<blocks(n0), kOdThreads, 0, c->stream>>>(s ? in_td : in_sd, width, height, dep[s][0],
                                                                       option->min_depth, option->max_depth);
    }
    KCHK(c);
    OdCamera cam[MI_ICP_ODOMETRY_MAX_LEVELS];
    {
        const float k0[9] = {intrinsic4[0], 0.0f, intrinsic4[2], 0.0f, intrinsic4[1], intrinsic4[3], 0.0f, 0.0f, 1.0f};
        std::memcpy(cam[0].k, k0, sizeof(k0));
        for (int l = 1; l < L; ++l) {  // CreateCameraMatrixPyramid (:332-347)
            for (int i = 0; i < 9; ++i) cam[l].k[i] = (float)(0.5 * (double)cam[l - 1].k[i]);
            cam[l].k[8] = 1.0f;
        }
    }
    // the running transformation and everything derived from it live on the device (OdState);
    // the host enqueues the whole run and synchronises once, at the end
    float* state_mem;
    TRY(ensure(c, c->stage[5], sizeof(OdState) / sizeof(float) + 16, &state_mem));
    OdState* state = reinterpret_cast<OdState*>(state_mem);
    const Mat4 init = load_T(odo_init);
    if (!c->od_host) HIPCHK(c, hipHostMalloc(&c->od_host, sizeof(OdState) + 64, hipHostMallocDefault));
    OdState* hst = reinterpret_cast<OdState*>(c->od_host);
    if (weighted) {  // the weighted variant's constants and its velocity, once
        std::memset(hst, 0, sizeof(OdState));
        hst->vel = I4;
        hst->sigma2 = option->sigma2_init;
        hst->nu = option->nu;
        for (int i = 0; i < 6; ++i) {
            hst->prev_twist[i] = prev_twist6 ? prev_twist6[i] : 0.0f;
            hst->inv_sigma[i] = option->inv_sigma_mat_diag[i];
        }
        HIPCHK(c, hipMemcpyAsync(state, hst, sizeof(OdState), hipMemcpyHostToDevice, c->stream));
    }
    // (two pinned slots: an asynchronous copy reads its host source when it executes, so the second
    // value must not overwrite the first one's source)
    Mat4* t_slots[2] = {&hst->T, reinterpret_cast<Mat4*>(reinterpret_cast<char*>(c->od_host) + sizeof(OdState))};
    int t_slot = 0;
    auto set_T = [&](const Mat4& T) -> int {
        Mat4* src = t_slots[t_slot++ & 1];
        *src = T;
        HIPCHK(c, hipMemcpyAsync(&state->T, src, sizeof(Mat4), hipMemcpyHostToDevice, c->stream));
        return MI_ICP_OK;
    };
    HIPCHK(c, hipMemsetAsync(sums, 0, 32 * sizeof(double), c->stream));
    OdArgs a{};
    a.out = sums;
    a.state = state;
    a.max_depth_diff = option->max_depth_diff;
    auto level_args = [&](int l) {
        a.depth_s = dep[0][l];
        a.depth_t = dep[1][l];
        a.color_s = col[0][l];
        a.color_t = col[1][l];
        a.dx_color = grad[0][l];
        a.dy_color = grad[1][l];
        a.dx_depth = grad[2][l];
        a.dy_depth = grad[3][l];
        a.w = lw[l];
        a.h = lh[l];
    };
    auto grid_for = [&](int l) {
        const int64_t n = (int64_t)lw[l] * lh[l];
        return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + kOdThreads - 1) / kOdThreads));
    };
    {   // NormalizeIntensity (:416-436) over the correspondences under odo_init
        TRY(set_T(init));
        od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[0], 0);
        level_args(0);
        od_accumulate<kOdMeans><<<grid_for(0), kOdThreads, 0, c->stream>>>(a);
        od_scale_by_mean<<<blocks(n0), kOdThreads, 0, c->stream>>>(col[0][0], n0, sums, 0);
        od_scale_by_mean<<<blocks(n0), kOdThreads, 0, c->stream>>>(col[1][0], n0, sums, 1);
        KCHK(c);
    }
    // ---- pyramids (rgbdimage.cu:96-112, image_factory.cu:251-278): colour Gaussian3 + Downsample,
    // depth Downsample only; Sobel3Dx / Sobel3Dy of the target per level (RGBDImage::FilterPyramid)
    for (int l = 1; l < L; ++l) {
        const int64_t np = (int64_t)lw[l - 1] * lh[l - 1], nn = (int64_t)lw[l] * lh[l];
        for (int s = 0; s < 2; ++s) {
            od_filter3<0, false><<<blocks(np), kOdThreads, 0, c->stream>>>(col[s][l - 1], lw[l - 1], lh[l - 1], scratch, 0.0f, 0.0f);
            od_downsample<<<blocks(nn), kOdThreads, 0, c->stream>>>(scratch, lw[l - 1], lh[l - 1], col[s][l]);
            od_downsample<<<blocks(nn), kOdThreads, 0, c->stream>>>(dep[s][l - 1], lw[l - 1], lh[l - 1], dep[s][l]);
        }
    }
    for (int l = 0; l < L; ++l) {
        const int64_t n = (int64_t)lw[l] * lh[l];
        od_filter3<1, false><<<blocks(n), kOdThreads, 0, c->stream>>>(col[1][l], lw[l], lh[l], grad[0][l], 0.0f, 0.0f);
        od_filter3<2, false><<<blocks(n), kOdThreads, 0, c->stream>>>(col[1][l], lw[l], lh[l], grad[1][l], 0.0f, 0.0f);
        od_filter3<1, false><<<blocks(n), kOdThreads, 0, c->stream>>>(dep[1][l], lw[l], lh[l], grad[2][l], 0.0f, 0.0f);
        od_filter3<2, false><<<blocks(n), kOdThreads, 0, c->stream>>>(dep[1][l], lw[l], lh[l], grad[3][l], 0.0f, 0.0f);
    }
    KCHK(c);

    // ---- ComputeMultiscale (:708-764): one accumulate + one step launch per iteration
    {
        bool zero = true;
        for (int i = 0; i < 16; ++i) zero = zero && (init.data()[i] == 0.0f);
        TRY(set_T(zero ? I4 : init));
        od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[L - 1], 0);  // terms for the coarsest level; zeroes the sums
    }
    for (int level = L - 1; level >= 0; --level) {
        level_args(level);
        const int iters = option->iterations[L - level - 1];
        for (int iter = 0; iter < iters; ++iter) {
            // the next evaluation: this level again, the next finer one, or level 0 (information matrix)
            const int next = (iter + 1 < iters) ? level : std::max(level - 1, 0);
            if (weighted) {  // two passes: the weights' normalisation, then the weighted system
                od_accumulate<kOdWeightSum><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
                od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[level], 3);
                od_accumulate<kOdWeighted><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
                od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[next], 2);
                continue;
            }
            if (jacobian == MI_ICP_ODOMETRY_COLOR_TERM) od_accumulate<kOdColor><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
            else od_accumulate<kOdHybrid><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
            od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[next], 1);
        }
        if (iters <= 0 && level > 0) od_step<<<1, 64, 0, c->stream>>>(state, sums, cam[level - 1], 0);
    }
    KCHK(c);
    // CreateInformationMatrix (:349-394): I + sum G^T G over the final correspondences
    level_args(0);
    od_accumulate<kOdInformation><<<grid_for(0), kOdThreads, 0, c->stream>>>(a);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(c->sys_host, sums, 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&hst->T, &state->T, sizeof(Mat4), hipMemcpyDeviceToHost, c->stream));
    if (weighted) HIPCHK(c, hipMemcpyAsync(&hst->vel, &state->vel, sizeof(Mat4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (weighted && twist6) od_matrix4_to_vector6(hst->vel, twist6);
    {
        int k = 0;
        for (int r = 0; r < 6; ++r)
            for (int q = r; q < 6; ++q, ++k) {
                information36[r * 6 + q] += c->sys_host[k];
                if (q != r) information36[q * 6 + r] += c->sys_host[k];
            }
        std::memcpy(transformation16, hst->T.data(), 16 * sizeof(float));
        *success = 1;  // without its determinant check the solver never reports failure (utility/eigen.cu:76-122)
    }
    return MI_ICP_OK;
}

int mi_icp_compute_rgbd_odometry(mi_icp_ctx* c, const float* source_color, const float* source_depth,
                                 const float* target_color, const float* target_depth, int width, int height,
                                 const float* intrinsic4, const float* odo_init, int jacobian,
                                 const mi_icp_odometry_option* option, int* success, float* transformation16,
                                 double* information36, int mem_kind) {
    return rgbd_odometry_impl(c, source_color, source_depth, target_color, target_depth, width, height, intrinsic4,
                              odo_init, jacobian, option, success, transformation16, information36, mem_kind, false,
                              nullptr, nullptr);
}

int mi_icp_compute_weighted_rgbd_odometry(mi_icp_ctx* c, const float* source_color, const float* source_depth,
                                          const float* target_color, const float* target_depth, int width, int height,
                                          const float* intrinsic4, const float* odo_init, const float* prev_twist6,
                                          const mi_icp_odometry_option* option, int* success, float* transformation16,
                                          float* twist6, double* information36, int mem_kind) {
    if (!twist6) return c ? fail(c, MI_ICP_ERR_INVALID, "compute_weighted_rgbd_odometry: twist6 is null") : MI_ICP_ERR_INVALID;
    return rgbd_odometry_impl(c, source_color, source_depth, target_color, target_depth, width, height, intrinsic4,
                              odo_init, MI_ICP_ODOMETRY_HYBRID_TERM, option, success, transformation16, information36,
                              mem_kind, true, prev_twist6, twist6);
}

static int estimate_normals_impl(mi_icp_ctx* c, const float* xyz, int64_t n, int knn, float r2,
                                 float* normals, int mem_kind) {
    TRY(check_ctx(c));
    if (n < 0 || (n > 0 && (!xyz || !normals))) return fail(c, MI_ICP_ERR_INVALID, "estimate_normals: bad arguments");
    if (knn > kKnnLimit) return fail(c, MI_ICP_ERR_INVALID, "estimate_normals: more than %d neighbours (knn::NUM_MAX_NN) are not supported", kKnnLimit);
    if (n == 0) return MI_ICP_OK;
    // The cloud gets a tree of its own in a private scratch context: a registration in flight on
    // this context (user estimators may call EstimateNormals between iterations) keeps its
    // target, source, correspondences and loop state.
    if (!c->aux) {
        const int rc = mi_icp_create(c->device, &c->aux);
        if (rc != MI_ICP_OK) return fail(c, rc, "estimate_normals: cannot create the scratch context");
    }
    mi_icp_ctx* a = c->aux;
    a->stream = c->stream;
    auto run = [&]() -> int {
        TRY(mi_icp_set_target(a, xyz, nullptr, nullptr, n, mem_kind));
        float* dn = normals;
        if (mem_kind == MI_ICP_HOST) TRY(ensure(a, a->stage[1], (size_t)n * 3, &dn));
        const int cap = knn_capacity(knn), waves = knn_waves(cap);
        const uint32_t nblocks = (uint32_t)((a->nleaf + waves * 8 - 1) / (waves * 8));
        const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
        int32_t* slab;
        TRY(ensure(a, a->knn_idx, (size_t)nblocks * waves * cap * 64, &slab));
#define MI_NRM_ARGS (const float*)a->nodes.p, (const float*)a->tblk.p, (const int32_t*)a->tidx.p, a->leaf_first, a->nts, a->nleaf, knn, r2, nblocks, \
                    dn, nullptr, nullptr, slab
        if (cap == kMaxKnn) knn_normals_kernel<0, kMaxKnn><<<grid, waves * 64, 0, a->stream>>>(MI_NRM_ARGS);
        else if (cap == kMaxKnnMid) knn_normals_kernel<0, kMaxKnnMid><<<grid, waves * 64, 0, a->stream>>>(MI_NRM_ARGS);
        else knn_normals_kernel<0, kMaxKnnBig><<<grid, waves * 64, 0, a->stream>>>(MI_NRM_ARGS);
#undef MI_NRM_ARGS
        KCHK(a);
        if (mem_kind == MI_ICP_HOST) TRY(from_device(a, (const float*)dn, normals, (size_t)n * 3, mem_kind));
        HIPCHK(a, hipStreamSynchronize(a->stream));
        return MI_ICP_OK;
    };
    const int rc = run();
    if (rc != MI_ICP_OK) return fail(c, rc, "estimate_normals: %s", a->err.c_str());
    return MI_ICP_OK;
}

int mi_icp_estimate_normals_knn(mi_icp_ctx* c, const float* xyz, int64_t n, int knn, float* normals,
                                int mem_kind) {
    return estimate_normals_impl(c, xyz, n, knn, INFINITY, normals, mem_kind);
}

int mi_icp_estimate_normals_radius(mi_icp_ctx* c, const float* xyz, int64_t n, float radius, int max_nn,
                                   float* normals, int mem_kind) {
    return estimate_normals_impl(c, xyz, n, max_nn, radius * radius, normals, mem_kind);
}

// ---------------------------------------------------------------------------
// knn::KDTreeFlann::SearchKNN / SearchRadius (knn/kdtree_flann.inl:46-122)
int mi_icp_search_knn(mi_icp_ctx* c, const float* queries, int64_t nq, int knn, float radius, int32_t* idx_out,
                      float* d2_out, int64_t* found, int mem_kind) {
    TRY(check_ctx(c));
    if (found) *found = 0;
    if (nq < 0 || knn < 0 || (nq > 0 && (!queries || !idx_out || !d2_out)))
        return fail(c, MI_ICP_ERR_INVALID, "search_knn: bad arguments");
    if (knn > kKnnLimit) return fail(c, MI_ICP_ERR_INVALID, "search_knn: more than %d neighbours (knn::NUM_MAX_NN) are not supported", kKnnLimit);
    if (c->nt <= 0) return fail(c, MI_ICP_ERR_STATE, "search_knn: no target cloud (mi_icp_set_target)");
    if (nq == 0 || knn == 0) return MI_ICP_OK;
    // the queries are staged exactly like an ICP source (Morton-ordered SoA + permutation)
    TRY(mi_icp_set_source(c, queries, nullptr, nullptr, nq, mem_kind));
    int32_t* d_idx = idx_out;
    float* d_d2 = d2_out;
    if (mem_kind == MI_ICP_HOST) {
        TRY(ensure(c, c->stage[4], (size_t)nq * knn, (int32_t**)&d_idx));
        TRY(ensure(c, c->stage[5], (size_t)nq * knn, &d_d2));
    }
    unsigned long long* cnt;
    TRY(ensure(c, c->flags, 8, (unsigned long long**)&cnt));
    HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), c->stream));
    const uint32_t npackets = (uint32_t)((nq + 63) / 64);
    const int cap = knn_capacity(knn), waves = knn_waves(cap);
    const uint32_t nblocks = (npackets + waves - 1) / waves;
    const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
    int32_t* slab;
    TRY(ensure(c, c->knn_idx, (size_t)nblocks * waves * cap * 64, &slab));
#define MI_KNN_ARGS (const float*)c->nodes.p, (const float*)c->tblk.p, (const int32_t*)c->tidx.p, c->leaf_first, (const float*)c->sx.p, \
                    (const float*)c->sy.p, (const float*)c->sz.p, (const int32_t*)c->sperm.p, (int)nq, c->nleaf, knn, \
                    radius > 0.0f ? radius * radius : INFINITY, nblocks, d_idx, d_d2, cnt, slab
    if (cap == kMaxKnn) knn_search_kernel<kMaxKnn><<<grid, waves * 64, 0, c->stream>>>(MI_KNN_ARGS);
    else if (cap == kMaxKnnMid) knn_search_kernel<kMaxKnnMid><<<grid, waves * 64, 0, c->stream>>>(MI_KNN_ARGS);
    else knn_search_kernel<kMaxKnnBig><<<grid, waves * 64, 0, c->stream>>>(MI_KNN_ARGS);
#undef MI_KNN_ARGS
    KCHK(c);
    if (mem_kind == MI_ICP_HOST) {
        TRY(from_device(c, (const int32_t*)d_idx, idx_out, (size_t)nq * knn, mem_kind));
        TRY(from_device(c, (const float*)d_d2, d2_out, (size_t)nq * knn, mem_kind));
    }
    HIPCHK(c, hipMemcpyAsync(c->sys_host, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (found) *found = (int64_t) * reinterpret_cast<unsigned long long*>(c->sys_host);
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
// Colored ICP (registration/colored_icp.cu)
int mi_icp_set_target_colors(mi_icp_ctx* c, const float* rgb, int mem_kind) {
    TRY(check_ctx(c));
    c->t_has_int = c->t_has_grad = false;
    if (!rgb || c->nt <= 0) return MI_ICP_OK;
    if (!c->t_has_nrm)  // the intensities ride in the normals' 4th lane; colored ICP needs normals anyway
        return fail(c, MI_ICP_ERR_STATE, "set_target_colors: the target has no normals");
    const float* d_rgb;
    TRY(to_device(c, rgb, (size_t)c->nt * 3, mem_kind, c->stage[1], &d_rgb));
    target_intensity<<<blocks_for(c->nts), 256, 0, c->stream>>>((const int32_t*)c->tidx.p, d_rgb, (int)c->nts,
                                                              (float4*)c->tnrm.p);
    KCHK(c);
    c->t_has_int = true;
    return MI_ICP_OK;
}

int mi_icp_set_source_colors(mi_icp_ctx* c, const float* rgb, int mem_kind) {
    TRY(check_ctx(c));
    c->s_has_int = false;
    if (!rgb || c->ns <= 0) return MI_ICP_OK;
    const float* d_rgb;
    float* sint;
    TRY(to_device(c, rgb, (size_t)c->ns * 3, mem_kind, c->stage[4], &d_rgb));
    TRY(ensure(c, c->sint, (size_t)c->ns, &sint));
    source_intensity<<<blocks_for(c->ns), 256, 0, c->stream>>>((const int32_t*)c->sperm.p, d_rgb, (int)c->ns, sint);
    KCHK(c);
    c->s_has_int = true;
    return MI_ICP_OK;
}

int mi_icp_set_lambda_geometric(mi_icp_ctx* c, float lambda_geometric) {
    if (!c) return MI_ICP_ERR_INVALID;
    // colored_icp.cu:49-50: out-of-range values fall back to the default
    c->lambda_geometric = (lambda_geometric < 0.0f || lambda_geometric > 1.0f) ? 0.968f : lambda_geometric;
    return MI_ICP_OK;
}

int mi_icp_compute_color_gradients(mi_icp_ctx* c, float radius, int max_nn, float* gradients_out, int mem_kind) {
    TRY(check_ctx(c));
    c->t_has_grad = false;
    if (c->nt <= 0) return MI_ICP_OK;
    if (!c->t_has_nrm || !c->t_has_int)
        return fail(c, MI_ICP_ERR_STATE, "compute_color_gradients: the target needs normals and colours");
    if (max_nn > kKnnLimit)
        return fail(c, MI_ICP_ERR_INVALID, "compute_color_gradients: more than %d neighbours (knn::NUM_MAX_NN) are not supported", kKnnLimit);
    const int64_t n = c->nt;
    float4* tgrad;
    TRY(ensure(c, c->tgrad, (size_t)c->nts, &tgrad));
    float* dg = gradients_out;
    if (gradients_out && mem_kind == MI_ICP_HOST) TRY(ensure(c, c->stage[1], (size_t)n * 3, &dg));
    const int cap = knn_capacity(max_nn), waves = knn_waves(cap);
    const uint32_t nblocks = (uint32_t)((c->nleaf + waves * 8 - 1) / (waves * 8));
    const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
    int32_t* slab;
    TRY(ensure(c, c->knn_idx, (size_t)nblocks * waves * cap * 64, &slab));
#define MI_GRAD_ARGS (const float*)c->nodes.p, (const float*)c->tblk.p, (const int32_t*)c->tidx.p, c->leaf_first, c->nts, c->nleaf, max_nn, \
                     radius * radius, nblocks, dg, (const float4*)c->tnrm.p, tgrad, slab
    if (cap == kMaxKnn) knn_normals_kernel<1, kMaxKnn><<<grid, waves * 64, 0, c->stream>>>(MI_GRAD_ARGS);
    else if (cap == kMaxKnnMid) knn_normals_kernel<1, kMaxKnnMid><<<grid, waves * 64, 0, c->stream>>>(MI_GRAD_ARGS);
    else knn_normals_kernel<1, kMaxKnnBig><<<grid, waves * 64, 0, c->stream>>>(MI_GRAD_ARGS);
#undef MI_GRAD_ARGS
    KCHK(c);
    c->t_has_grad = true;
    if (gradients_out) {
        if (mem_kind == MI_ICP_HOST) TRY(from_device(c, (const float*)dg, gradients_out, (size_t)n * 3, mem_kind));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MI_ICP_OK;
}

int mi_icp_registration_colored_icp(mi_icp_ctx* c, float max_distance, const float* init,
                                    const mi_icp_params* params, float lambda_geometric, mi_icp_result* out) {
    TRY(check_ctx(c));
    TRY(mi_icp_set_lambda_geometric(c, lambda_geometric));
    // colored_icp.cu:337-338: gradients over KDTreeSearchParamRadius(max_distance * 2, 30)
    if (c->nt > 0 && c->t_has_nrm && c->t_has_int)
        TRY(mi_icp_compute_color_gradients(c, max_distance * 2.0f, 30, nullptr, MI_ICP_DEVICE));
    return mi_icp_registration_icp(c, kEstColored, max_distance, init, params, out);
}

// ---------------------------------------------------------------------------
int mi_icp_comm_unique_id(char* id128) {
    if (!id128) return MI_ICP_ERR_INVALID;
    if (!load_rccl(g_rccl)) return MI_ICP_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return MI_ICP_ERR_COMM;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return MI_ICP_OK;
}

int mi_icp_comm_init(mi_icp_ctx* c, const char* id128, int nranks, int rank) {
    TRY(check_ctx(c));
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MI_ICP_ERR_INVALID, "comm_init: bad arguments");
    if (!load_rccl(g_rccl)) return fail(c, MI_ICP_ERR_COMM, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    // ncclCommInitRank blocks until EVERY rank has joined; one that never does (a crashed peer, a bootstrap socket the
    // container's network does not route) would hold the caller forever -- and a scaling run with it, although the
    // node's mailbox needs no RCCL at all.  So the call runs on a helper thread that owns nothing but its result, and
    // the caller waits MI_ICP_COMM_INIT_MS (default 120 s; <= 0: for ever) for it: past that the communicator is given
    // up (the thread is left behind, blocked; it touches nothing of this context), the call fails with
    // MI_ICP_ERR_COMM and the caller may go on with mi_icp_comm_init_local.
    struct InitResult {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        ncclResult_t r = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    static const long init_ms = [] { const char* e = std::getenv("MI_ICP_COMM_INIT_MS"); return e ? std::atol(e) : 120000L; }();
    auto res = std::make_shared<InitResult>();
    {
        const int device = c->device;
        std::thread([res, device, nranks, id, rank] {
            ncclComm_t comm = nullptr;
            ncclResult_t r = (hipSetDevice(device) == hipSuccess) ? g_rccl.CommInitRank(&comm, nranks, id, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> g(res->m);
            res->r = r;
            res->comm = comm;
            res->done = true;
            res->cv.notify_all();
        }).detach();
    }
    {
        std::unique_lock<std::mutex> g(res->m);
        if (init_ms > 0) {
            if (!res->cv.wait_for(g, std::chrono::milliseconds(init_ms), [&] { return res->done; }))
                return fail(c, MI_ICP_ERR_COMM, "ncclCommInitRank did not return within %ld ms (MI_ICP_COMM_INIT_MS): given up", init_ms);
        } else {
            res->cv.wait(g, [&] { return res->done; });
        }
    }
    const ncclResult_t r = res->r;
    c->comm = res->comm;
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return fail(c, MI_ICP_ERR_COMM, "ncclCommInitRank failed (%d)", (int)r);
    }
    c->nranks = nranks;
    c->rank = rank;
    c->xchg = 3;
    // One node: the per-iteration exchange goes through the mailbox (mailbox.h) instead of an
    // ncclAllReduce launch; the communicator stays for whatever the mailbox cannot do.  The box is named
    // after the job's unique id.  MI_ICP_NO_MAILBOX=1, more than 16 ranks or a failed set-up: RCCL only.
    const bool no_mailbox = std::getenv("MI_ICP_NO_MAILBOX") != nullptr;  // (read at every call: a caller may fall back)
    c->comm_broken = false;
    if (!no_mailbox && nranks > 1 && nranks <= kMailRanks) {
        unsigned long long h = 1469598103934665603ull;  // FNV-1a of the id
        for (int i = 0; i < 128; ++i) h = (h ^ (unsigned char)id128[i]) * 1099511628211ull;
        char name[64];
        std::snprintf(name, sizeof(name), "/mi_icp_%016llx", h);
        const int opened = mailbox_open(c, name, nranks, rank) == MI_ICP_OK ? 1 : 0;  // (c->err says why not; not fatal)
        // The ranks must AGREE on how they exchange: one that could not open the box while its peers did would
        // wait in an ncclAllReduce nobody joins, and they for a post that never comes.  So: a min over the
        // communicator that exists by now, and the mailbox only if every rank has it.
        int32_t* flag;
        TRY(ensure(c, c->mail_state, 64, (uint32_t**)&flag));
        int32_t* agree = flag + 32;  // (behind the exchange counter and its error word)
        HIPCHK(c, hipMemcpyAsync(agree, &opened, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        ncclResult_t ar = g_rccl.AllReduce(agree, agree, 1, ncclInt32, ncclMin, c->comm, c->stream);
        int32_t all = 0;
        if (ar == ncclSuccess) {
            HIPCHK(c, hipMemcpyAsync(&all, agree, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (ar != ncclSuccess || all != 1) mailbox_close(c);
    }
    if (!c->mail_dev) c->xchg = 3;
    return MI_ICP_OK;
}

int mi_icp_comm_init_local(mi_icp_ctx* c, const char* job_name, int nranks, int rank) {
    TRY(check_ctx(c));
    if (!job_name || !job_name[0] || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(c, MI_ICP_ERR_INVALID, "comm_init_local: bad arguments");
    std::string name = "/mi_icp_";
    for (const char* p = job_name; *p && name.size() < 60; ++p)
        name += (std::isalnum((unsigned char)*p) || *p == '_' || *p == '-') ? *p : '_';
    c->nranks = nranks;
    c->rank = rank;
    c->comm_broken = false;
    // (MI_ICP_MAILBOX_SOLO: a one-rank box, to time the exchange's fixed cost on a single GPU)
    if (nranks > 1 || std::getenv("MI_ICP_MAILBOX_SOLO")) {
        const int rc = mailbox_open(c, name, nranks, rank);
        if (rc != MI_ICP_OK) {
            c->nranks = 1;
            c->rank = 0;
            return rc;
        }
    }
    return MI_ICP_OK;
}

int mi_icp_comm_kind(const mi_icp_ctx* c) {
    if (!c) return 0;
    if (mail_on(c)) return c->xchg == 2 ? 3 : 2;
    return c->comm ? 1 : 0;
}

// ---- the exchange's self-test and choice --------------------------------------------------------------------
namespace {
// Every rank's CPU writes four doubles into the box and reads everybody's: a barrier and an all-gather in one, through
// the shared mapping alone (no GPU, no RCCL).  False: a rank did not show up within the attach time-out.
bool box_gather(mi_icp_ctx* c, const double v[4], double out[kMailRanks][4]) {
    MailBox* box = c->mail_host;
    const uint32_t epoch = ++c->tune_epoch;
    const int slot = (int)(epoch & 1u);
    for (int k = 0; k < 4; ++k) box->tune_val[slot][c->rank][k] = v[k];
    __atomic_store_n(&box->tune_epoch[c->rank], epoch, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int r = 0; r < c->nranks; ++r) all = all && __atomic_load_n(&box->tune_epoch[r], __ATOMIC_ACQUIRE) >= epoch;
        if (all) break;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(mail_attach_timeout_ms())) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    for (int r = 0; r < c->nranks; ++r)
        for (int k = 0; k < 4; ++k) out[r][k] = box->tune_val[slot][r][k];
    return true;
}
}  // namespace

int mi_icp_comm_autotune(mi_icp_ctx* c, int exchanges, double* lat_us3, int* info4) {
    TRY(check_ctx(c));
    if (!lat_us3 || !info4) return fail(c, MI_ICP_ERR_INVALID, "comm_autotune: null argument");
    TRY(comm_usable(c));
    for (int k = 0; k < 3; ++k) lat_us3[k] = -1.0;  // -1: path not available, -2: failed its self-test
    info4[0] = info4[1] = info4[2] = info4[3] = 0;
    const int n = std::min(std::max(exchanges > 0 ? exchanges : 200, 8), 10000);
    info4[2] = n;
    if (c->comm && g_rccl.CommCount) {
        int cnt = 0;
        if (g_rccl.CommCount(c->comm, &cnt) == ncclSuccess) info4[1] = cnt;
    }
    if (!c->mail_dev && !c->comm) return MI_ICP_OK;  // a single rank: nothing to choose
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double* buf;
    TRY(ensure(c, c->sys_dev, kSysSize, &buf));
    uint32_t* state;
    TRY(ensure(c, c->mail_state, 64, &state));
    int32_t* status = (int32_t*)state + 8;  // [8]: timed out, [9]: wrong totals
    int32_t* st_host = reinterpret_cast<int32_t*>(c->sys_host + 40);  // (spare words of the pinned buffer)
    const bool box = c->mail_dev != nullptr && c->mail_host != nullptr;
    bool verified = true;
    // the mailbox paths: n exchanges inside one launch
    constexpr uint32_t kSelfTestSpin = 1u << 20;  // ~2 s of polling: a path that does not deliver fails fast
    const int before = c->xchg;
    // test hook: MI_ICP_SELFTEST_BREAK="wrong:<path>" / "mute:<path>" makes the LAST rank post a wrong vector / nothing
    // on that path (tests/test_gpu_distributed.py: a path that fails is skipped on every rank alike, never fatal)
    int break_path = 0;
    bool break_mute = false;
    if (const char* e = std::getenv("MI_ICP_SELFTEST_BREAK")) {
        if (c->rank == c->nranks - 1 && (std::strncmp(e, "wrong:", 6) == 0 || std::strncmp(e, "mute:", 5) == 0)) {
            break_mute = e[0] == 'm';
            break_path = std::atoi(std::strchr(e, ':') + 1);
        }
    }
    for (int path = 1; path <= 2 && box; ++path) {
        if (path == 2 && !c->inbox) continue;
        double mine[4] = {0, 0, 0, 0}, all[kMailRanks][4];
        if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet at the self-test");
        c->xchg = path;
        MailArgs m = mail_args(c);
        m.spin_limit = kSelfTestSpin;
        float ms = 0.0f;
        bool ok = true;
        for (int round = 0; round < 2 && ok; ++round) {  // (a short round first: first touch of the mappings, launch skew)
            const int cnt = round == 0 ? 4 : n;
            ok = hipMemsetAsync(status, 0, 2 * sizeof(int32_t), c->stream) == hipSuccess &&
                 hipEventRecord(c->ev[0], c->stream) == hipSuccess;
            if (!ok) break;
            if (break_path == path && break_mute) {
                ok = false;  // (says nothing; its peers' kernels time out)
                break;
            }
            mail_selftest_kernel<<<1, 256, 0, c->stream>>>(m, cnt, (break_path == path) ? 0.5 : 0.0, buf, status);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(c->ev[1], c->stream) == hipSuccess &&
                 hipMemcpyAsync(st_host, status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess;
            ok = ok && st_host[0] == 0 && st_host[1] == 0;
        }
        (void)hipGetLastError();
        // this rank's figure, and its exchange counter (should a path have failed, the ranks' counters are apart)
        uint32_t seq = 0;
        (void)hipMemcpy(&seq, state, sizeof(uint32_t), hipMemcpyDeviceToHost);
        mine[0] = ok ? (double)ms * 1e3 / (double)n : 1e30;
        mine[1] = (double)seq;
        if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after a self-test");
        double worst = 0.0, top = 0.0;
        for (int r = 0; r < c->nranks; ++r) {
            worst = std::max(worst, all[r][0]);
            top = std::max(top, all[r][1]);
        }
        if (worst < 1e29) {
            lat_us3[path - 1] = worst;
        } else {
            lat_us3[path - 1] = -2.0;
            // re-align: every rank continues from the same exchange number, beyond anything posted so far
            const uint32_t fresh = (uint32_t)top + 4096u;
            HIPCHK(c, hipMemcpy(state, &fresh, sizeof(uint32_t), hipMemcpyHostToDevice));
            if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after a failed self-test");
        }
    }
    c->xchg = before;
    // the in-library RCCL all-reduce: n collectives, each behind a one-block kernel (the loop's step kernel stands
    // behind every all-reduce like that)
    if (c->comm) {
        bool ok = hipMemsetAsync(status, 0, 2 * sizeof(int32_t), c->stream) == hipSuccess;
        float ms = 0.0f;
        for (int round = 0; round < 2 && ok; ++round) {
            const int cnt = round == 0 ? 4 : n;
            ok = hipEventRecord(c->ev[0], c->stream) == hipSuccess;
            for (int it = 0; it < cnt && ok; ++it) {
                rccl_selftest_fill<<<1, 64, 0, c->stream>>>(buf, c->rank, c->nranks, it, status);
                ok = g_rccl.AllReduce(buf, buf, kSysSize, ncclDouble, ncclSum, c->comm, c->stream) == ncclSuccess;
            }
            rccl_selftest_fill<<<1, 64, 0, c->stream>>>(buf, c->rank, c->nranks, cnt, status);  // (checks the last one)
            ok = ok && hipEventRecord(c->ev[1], c->stream) == hipSuccess &&
                 hipMemcpyAsync(st_host, status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess;
            ok = ok && st_host[1] == 0;
        }
        (void)hipGetLastError();
        double lat = ok ? (double)ms * 1e3 / (double)n : 1e30;
        if (box) {
            double mine[4] = {lat, 0, 0, 0}, all[kMailRanks][4];
            if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after the RCCL self-test");
            for (int r = 0; r < c->nranks; ++r) lat = std::max(lat, all[r][0]);
        } else if (ok) {  // no box: the communicator itself carries the maximum
            double* d = buf;
            HIPCHK(c, hipMemcpy(d, &lat, sizeof(double), hipMemcpyHostToDevice));
            if (g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, c->comm, c->stream) == ncclSuccess) {
                HIPCHK(c, hipMemcpyAsync(c->sys_host, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                lat = c->sys_host[0];
            }
        }
        lat_us3[2] = lat < 1e29 ? lat : -2.0;
        verified = verified && lat < 1e29;
    }
    // the fastest path that passed on EVERY rank (the figures are the maxima over the ranks: identical everywhere)
    int best = 0;
    for (int p = 1; p <= 3; ++p)
        if (lat_us3[p - 1] >= 0.0 && (best == 0 || lat_us3[p - 1] < lat_us3[best - 1])) best = p;
    if (best == 0) return comm_failed(c, "comm_autotune: no exchange path passed its self-test on every rank");
    for (int p = 1; p <= 3; ++p) verified = verified && lat_us3[p - 1] != -2.0;
    c->xchg = best;
    info4[0] = best;
    info4[3] = verified ? 1 : 0;
    return MI_ICP_OK;
}

int mi_icp_comm_destroy(mi_icp_ctx* c) {
    TRY(check_ctx(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mailbox_close(c);
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->nranks = 1;
    c->rank = 0;
    c->comm_broken = false;
    c->xchg = 0;
    return MI_ICP_OK;
}

// ---------------------------------------------------------------------------
// test-only entry points (include/mi_icp_debug.h)
int mi_icp_debug_sort_pairs(mi_icp_ctx* c, uint64_t* keys, uint32_t* vals, int64_t n, int key_bits) {
    TRY(check_ctx(c));
    if (n < 0 || key_bits < 1 || key_bits > 64 || (n > 0 && (!keys || !vals)))
        return fail(c, MI_ICP_ERR_INVALID, "debug_sort_pairs: bad arguments");
    if (n == 0) return MI_ICP_OK;
    SortBuffers sb;
    TRY(sort_buffers(c, n, &sb));
    HIPCHK(c, hipMemcpyAsync(sb.keys[0], keys, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(sb.vals[0], vals, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    const int cur = radix_sort_pairs(c->stream, sb, n, key_bits);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(keys, sb.keys[cur], (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(vals, sb.vals[cur], (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

int mi_icp_debug_exclusive_scan(mi_icp_ctx* c, const uint32_t* in, uint32_t* out, int64_t n,
                                uint64_t* total) {
    TRY(check_ctx(c));
    if (n < 0 || (n > 0 && (!in || !out))) return fail(c, MI_ICP_ERR_INVALID, "debug_exclusive_scan: bad arguments");
    if (total) *total = 0;
    if (n == 0) return MI_ICP_OK;
    uint32_t *d, *tmp;
    TRY(ensure(c, c->flags, (size_t)n, &d));
    TRY(ensure(c, c->scan_tmp, (size_t)scan_num_tiles(n) + 2, &tmp));
    HIPCHK(c, hipMemcpyAsync(d, in, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    exclusive_scan_u32(c->stream, d, d, n, tmp);
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->u_host, tmp + scan_num_tiles(n), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (total) *total = c->u_host[0];
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

int mi_icp_debug_nn_stats8(mi_icp_ctx* c, const float* T, float radius, int use_seed, uint64_t* out8) {
    TRY(check_ctx(c));
    if (!out8 || c->ns <= 0 || c->nt <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_nn_stats: bad state/arguments");
    unsigned long long* d;
    TRY(ensure(c, c->flags, 16, (unsigned long long**)&d));
    HIPCHK(c, hipMemsetAsync(d, 0, 16 * sizeof(unsigned long long), c->stream));
    TRY(launch_nn(c, load_T(T), radius * radius, use_seed != 0, d));
    HIPCHK(c, hipMemcpyAsync(c->sys_host, d, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    if (std::getenv("MI_ICP_CENSUS_WHY")) {
        const unsigned long long* w = (const unsigned long long*)c->sys_host;
        std::fprintf(stderr, "walkers: no seed %llu, no region %llu, no halo %llu, beyond far reach %llu, far lines short %llu\n",
                     w[8], w[9], w[10], w[11], w[12]);
    }
    std::memcpy(out8, c->sys_host, 8 * sizeof(uint64_t));
    return MI_ICP_OK;
}

int mi_icp_debug_nn_stats(mi_icp_ctx* c, const float* T, float radius, int use_seed, uint64_t* out4) {
    uint64_t all[8];
    if (!out4) return MI_ICP_ERR_INVALID;
    TRY(mi_icp_debug_nn_stats8(c, T, radius, use_seed, all));
    std::memcpy(out4, all, 4 * sizeof(uint64_t));
    return MI_ICP_OK;
}

int mi_icp_debug_get_leaf_regions(mi_icp_ctx* c, float* regions_out) {
    TRY(check_ctx(c));
    if (!regions_out || c->nt <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_get_leaf_regions: no target / bad arguments");
    HIPCHK(c, hipMemcpy2DAsync(regions_out, kLeafRegFloats * sizeof(float), lreg_of(c), kLeafRegStride * sizeof(float),
                               kLeafRegFloats * sizeof(float), (size_t)c->nleaf, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

int mi_icp_debug_get_leaf_halos(mi_icp_ctx* c, float* halos_out) {
    TRY(check_ctx(c));
    if (!halos_out || c->nt <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_get_leaf_halos: no target / bad arguments");
    TRY(ensure_links(c));
    const size_t count = (size_t)c->nleaf * kHaloLines * kHaloLineFloats;
    if (!c->thalo.p) {  // no halos on this tree (MI_ICP_NO_CELLS / MI_ICP_NO_LINKS)
        std::memset(halos_out, 0, count * sizeof(float));
        return MI_ICP_OK;
    }
    HIPCHK(c, hipMemcpyAsync(halos_out, c->thalo.p, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

int mi_icp_debug_last_search_kind(const mi_icp_ctx* c) { return c ? c->last_search_kind : -1; }

int mi_icp_debug_occupancy(int which) {
    int blocks = -1;
    hipError_t e = hipErrorInvalidValue;
    switch (which) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kd_build_groups, kKdThreads, 0); break;
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, nn_packet_kernel<true, false>, kNNThreads, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, nn_packet_kernel<false, false>, kNNThreads, 0); break;
        case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reduce_pt2pl_kernel<4, 1>, kReduceThreads, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, leaf_halo_build, 64, 0); break;
        case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, rs_scatter_pay<8>, kSortThreads, 0); break;
        case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, voxel_means_wave, 64, 0); break;
        case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, icp_mid_iteration_kernel<1>, kReduceThreads, 0); break;
        default: return -1;
    }
    return e == hipSuccess ? blocks : -2;
}

int mi_icp_debug_solve_both(int device, const double* systems, int n, float det_thresh, float* out_serial,
                            float* out_wave, int32_t* ok_serial, int32_t* ok_wave) {
    if (!systems || n <= 0 || !out_serial || !out_wave || !ok_serial || !ok_wave) return MI_ICP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MI_ICP_ERR_NO_DEVICE;
    double* d_sys = nullptr;
    float* d_out = nullptr;
    int32_t* d_ok = nullptr;
    int rc = MI_ICP_ERR_HIP;
    if (hipMalloc(&d_sys, (size_t)n * 32 * sizeof(double)) == hipSuccess &&
        hipMalloc(&d_out, (size_t)n * 32 * sizeof(float)) == hipSuccess &&
        hipMalloc(&d_ok, (size_t)n * 2 * sizeof(int32_t)) == hipSuccess &&
        hipMemcpy(d_sys, systems, (size_t)n * 32 * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
        mi::solve_both_kernel<<<n, 64>>>(d_sys, det_thresh, d_out, d_out + (size_t)n * 16, d_ok, d_ok + n);
        if (hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(out_serial, d_out, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(out_wave, d_out + (size_t)n * 16, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(ok_serial, d_ok, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(ok_wave, d_ok + n, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess)
            rc = MI_ICP_OK;
    }
    (void)hipFree(d_sys);
    (void)hipFree(d_out);
    (void)hipFree(d_ok);
    return rc;
}

int mi_icp_debug_drop_seeds(mi_icp_ctx* c) {
    TRY(check_ctx(c));
    c->nn_valid = false;
    return MI_ICP_OK;
}

int mi_icp_debug_get_tree(mi_icp_ctx* c, int64_t* info5, float* records_out, float* leaf_lines_out) {
    TRY(check_ctx(c));
    if (!info5 || c->nt <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_get_tree: no target / bad arguments");
    info5[0] = c->nts;
    info5[1] = c->nleaf;
    info5[2] = (int64_t)c->leaf_first;
    info5[3] = (int64_t)c->nrecords;
    info5[4] = c->nt;
    if (records_out)
        HIPCHK(c, hipMemcpyAsync(records_out, c->nodes.p, (size_t)c->nrecords * kRecordFloats * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
    // (handed out in the form x[8] y[8] z[8] orig_idx[8]: the indices have an array of their own on the device, the
    // lines' fourth rows hold the regions -- mi_icp_debug_get_leaf_regions)
    if (leaf_lines_out) {
        HIPCHK(c, hipMemcpyAsync(leaf_lines_out, c->tblk.p, (size_t)c->nleaf * kLeafFloats * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(leaf_lines_out + kLeafRegOffset, kLeafFloats * sizeof(float), c->tidx.p, kLeaf * sizeof(int32_t),
                                   kLeaf * sizeof(int32_t), (size_t)c->nleaf, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

}  // extern "C"
