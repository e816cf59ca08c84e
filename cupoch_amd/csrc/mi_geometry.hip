// mi_geometry.hip -- the geometry entry points beside the registration: Transform, bounds / centre, Translate / Scale /
// Rotate, GICP covariances, VoxelDownSample, depth / RGB-D frame -> cloud, RGB-D odometry, colours
// (one translation unit of libmi_icp.so; csrc/ctx.h lists them)
#include "ctx.h"
#include "depth_kernels.h"
#include "geometry_kernels.h"
#include "lbvh.h"
#include "odometry.h"
#include "reduce.h"
#include "voxel_dense.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

namespace mi {
namespace eng {
int occupancy_geometry(int which) {
    int blocks = -1;
    hipError_t e = hipErrorInvalidValue;
    if (which == 5) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, rs_scatter_pay<8>, kSortThreads, 0);
    else if (which == 6) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, voxel_means_wave, 64, 0);
    else if (which == 7) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, vx_scatter<1>, kVxThreads, 0);
    else if (which == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, vx_finish<false, false>, kVxFinThreads, 0);
    else return -1;
    return e == hipSuccess ? blocks : -2;
}
}  // namespace eng
}  // namespace mi

extern "C" {

#ifdef MI_VX_CLOCKS
// measurements only (scripts/dev/voxel_dense_clocks.py): the phase clocks of the last vx_scatter / vx_finish launches
int mi_vx_clocks_dump(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vx_clk), sizeof(unsigned long long) * 2 * 4096 * kVxClkSlots) == hipSuccess ? 0 : -1;
}
#endif

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

static int vx_cu_count() {
    static const int ncu = [] { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); return (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }();
    return ncu;
}

// the order of LDS adds inside one instruction (voxel_dense.h "Ranks"), checked once per context
static int vx_order_ok(mi_icp_ctx* c, bool* ok) {
    if (c->vx_order == 0) {
        uint32_t* w;
        TRY(ensure(c, c->vx_tab, (size_t)64, &w));
        HIPCHK(c, hipMemsetAsync(w, 0, sizeof(uint32_t), c->stream));
        vx_probe_order<<<64, 256, 0, c->stream>>>(w);
        KCHK(c);
        HIPCHK(c, hipMemcpyAsync(c->u_host, w, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->vx_order = (c->u_host[0] == 0u) ? 1 : -1;
    }
    *ok = c->vx_order > 0;
    return MI_ICP_OK;
}

// VoxelDownSample of a DENSE grid (voxel_dense.h): every point moves once.  Launched BEHIND the bounds kernels without
// waiting for them: the plan is made on the device (vx_plan_kernel: a packed key of 14 ... 21 bits and enough points per
// bucket), every kernel reads it there and does nothing when the grid is not one for this path.  The caller then waits
// ONCE, for the bounds and this path's control words together.  *launched = false: nothing was started.
static int voxel_dense_launch(mi_icp_ctx* c, const float* dp, const float* dn, const float* dcol, int64_t n, float voxel,
                              float* out_xyz, float* out_normals, float* out_colors, int mem_kind,
                              bool* launched, float** op_, float** on_, float** oc_) {
    *launched = false;
    if (std::getenv("MI_ICP_NO_DENSE_VOXEL")) return MI_ICP_OK;  // A/B switch, read at every call (tests compare both paths)
    if (n < (1 << 17) || n > ((int64_t)1 << 26)) return MI_ICP_OK;
    bool ordered = false;
    TRY(vx_order_ok(c, &ordered));
    if (!ordered) return MI_ICP_OK;
    const int ncu = vx_cu_count();
    const int ntiles = (int)((n + kVxTile - 1) / kVxTile);
    const int nsegs = (ntiles + kVxSeg - 1) / kVxSeg;
    // the tables, sized for 2048 buckets: the buckets' occupied-voxel counts, the plan, [ntiles][2048], [nsegs][2048],
    // bucket_start[2049], the control words
    const size_t plan_words = (sizeof(VxDev) + 7) / 8 * 2;
    const size_t words = (size_t)kVxMaxBins + plan_words + ((size_t)ntiles + nsegs) * kVxMaxBins + kVxMaxBins + 1 + kVxCtlWords;
    uint32_t* w;
    TRY(ensure(c, c->vx_tab, words, &w));
    uint32_t* occ = w;
    VxDev* plan = reinterpret_cast<VxDev*>(w + (size_t)kVxMaxBins);
    uint32_t* tab = w + (size_t)kVxMaxBins + plan_words;
    uint32_t* seg_tot = tab + (size_t)ntiles * kVxMaxBins;
    uint32_t* bucket_start = seg_tot + (size_t)nsegs * kVxMaxBins;
    uint32_t* ctl = bucket_start + kVxMaxBins + 1;
    VxArrays a;
    const float* in[3] = {dp, dn, dcol};
    const int64_t vmax = std::min<int64_t>(n, (int64_t)1 << 22);
    Pay3* tmp[3] = {nullptr, nullptr, nullptr};  // the buckets' means before they are moved together: a slot per cell of the grid
    for (int k = 0; k < 3; ++k) {
        a.in[k] = reinterpret_cast<const Pay3*>(in[k]);
        a.out[k] = nullptr;
        if (in[k]) {
            TRY(ensure(c, c->vpay[k], (size_t)n, &a.out[k]));
            TRY(ensure(c, c->vpay[3 + k], (size_t)1 << 22, &tmp[k]));
        }
    }
    float *op = out_xyz, *on = out_normals, *oc = out_colors;
    if (mem_kind == MI_ICP_HOST) {
        TRY(ensure(c, c->stage[3], (size_t)vmax * 3, &op));
        if (dn) TRY(ensure(c, c->stage[4], (size_t)vmax * 3, &on));
        if (dcol) TRY(ensure(c, c->stage[5], (size_t)vmax * 3, &oc));
    }
    {   // the bounds (compute_bounds' two launches, the second one making the plan as well)
        float* part;
        TRY(ensure(c, c->bounds_part, (size_t)kBoundsBlocks * 6, &part));
        const int nb = (int)std::min<int64_t>(kBoundsBlocks, blocks_for(n));
        bounds_partial<<<nb, 256, 0, c->stream>>>(dp, (int)n, part);
        static const int hb_force = [] { const char* e = std::getenv("MI_ICP_VOXEL_HB"); return e ? std::atoi(e) : 0; }();  // measurements
        vx_bounds_plan<<<1, 64, 0, c->stream>>>(part, nb, voxel, (long long)n, hb_force, plan, ctl);
    }
    vx_hist<<<ntiles, kVxThreads, 0, c->stream>>>(a.in[0], (int)n, plan, tab);
    vx_colsum<<<dim3((unsigned)nsegs, (unsigned)(kVxMaxBins / 256)), 256, 0, c->stream>>>(tab, ntiles, plan, seg_tot);
    vx_colscan<<<1, 1024, 0, c->stream>>>(seg_tot, nsegs, (int)n, plan, bucket_start, ctl);
    {   // the arrays that are there, packed to the front (vx_scatter<kArrays>)
        VxArrays pk = a;
        int na = 1;
        for (int k = 1; k < 3; ++k)
            if (a.in[k]) {
                pk.in[na] = a.in[k];
                pk.out[na] = a.out[k];
                ++na;
            }
        const int grid = std::min(ntiles, ncu);
        if (na == 1) vx_scatter<1><<<grid, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, plan, tab, seg_tot, bucket_start, ctl);
        else if (na == 2) vx_scatter<2><<<grid, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, plan, tab, seg_tot, bucket_start, ctl);
        else vx_scatter<3><<<grid, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, plan, tab, seg_tot, bucket_start, ctl);
    }
#define MI_VX_FINISH(N, C)                                                                                                     \
    vx_finish<N, C><<<std::min(kVxMaxBins, ncu), kVxFinThreads, 0, c->stream>>>(a.out[0], a.out[1], a.out[2], plan, bucket_start, ctl, \
                                                                               occ, tmp[0], tmp[1], tmp[2])
    if (dn && dcol) MI_VX_FINISH(true, true);
    else if (dn) MI_VX_FINISH(true, false);
    else if (dcol) MI_VX_FINISH(false, true);
    else MI_VX_FINISH(false, false);
#undef MI_VX_FINISH
    vx_compact<<<kVxMaxBins, 256, 0, c->stream>>>(plan, ctl, occ, tmp[0], tmp[1], tmp[2], reinterpret_cast<Pay3*>(op),
                                                   reinterpret_cast<Pay3*>(dn ? on : nullptr), reinterpret_cast<Pay3*>(dcol ? oc : nullptr));
    KCHK(c);
    HIPCHK(c, hipMemcpyAsync(c->u_host, ctl, kVxCtlWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));  // (with the bounds)
    *launched = true;
    *op_ = op;
    *on_ = on;
    *oc_ = oc;
    return MI_ICP_OK;
}

// The general path's sort for LARGE clouds on fine grids (the key sorted whole, L = 0): the dense path's partition
// kernels as a radix sort of 11-bit digits -- two or three stable passes for a key of up to 32 bits where 8-bit digits
// take three or four, keys recomputed from the points in every pass instead of carried and stored, four launches a pass
// instead of five.  The plans of the passes (digit = (key >> L) & (B - 1)) are written by the host, which knows the grid
// here.  pay[]: the arrays that hold the sorted cloud.
static int voxel_wide_sort(mi_icp_ctx* c, const Pay3* const first[3], int64_t n, const VoxelGrid& grid, int bits, const Pay3* pay[3]) {
    const int npass = (bits + 10) / 11, width = (bits + npass - 1) / npass;
    const int ntiles = (int)((n + kVxTile - 1) / kVxTile);
    const int nsegs = (ntiles + kVxSeg - 1) / kVxSeg;
    static_assert(sizeof(VxDev) == 64, "three plans in 192 bytes of the pinned block");
    const size_t plan_words = 3 * sizeof(VxDev) / 4;
    const size_t words = plan_words + ((size_t)ntiles + nsegs) * kVxMaxBins + kVxMaxBins + 1 + kVxCtlWords;
    uint32_t* w;
    TRY(ensure(c, c->vx_tab, words, &w));
    VxDev* plans = reinterpret_cast<VxDev*>(w);
    uint32_t* tab = w + plan_words;
    uint32_t* seg_tot = tab + (size_t)ntiles * kVxMaxBins;
    uint32_t* bucket_start = seg_tot + (size_t)nsegs * kVxMaxBins;
    uint32_t* ctl = bucket_start + kVxMaxBins + 1;
    VxDev* hp = reinterpret_cast<VxDev*>(c->f_host + 16);  // (pinned; [0..7] hold the bounds)
    for (int p = 0; p < npass; ++p) {
        VxDev v;
        v.g.g = grid;
        v.g.inv = 1.0f / grid.voxel;
        v.g.key_mask = (bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u);
        v.bits = bits;
        v.L = p * width;
        v.hb = std::min(width, bits - p * width);
        v.B = 1 << v.hb;
        v.ok = 1;
        v.max_bucket = 0xffffffffu;
        v.empty = 0;
        v.pad = 0;
        hp[p] = v;
    }
    HIPCHK(c, hipMemcpyAsync(plans, hp, (size_t)npass * sizeof(VxDev), hipMemcpyHostToDevice, c->stream));
    Pay3* buf[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    for (int set = 0; set < std::min(npass, 2); ++set)
        for (int a = 0; a < 3; ++a)
            if (first[a]) TRY(ensure(c, c->vpay[set * 3 + a], (size_t)n, &buf[set][a]));
    for (int a = 0; a < 3; ++a) pay[a] = first[a];
    int na = 0;
    for (int a = 0; a < 3; ++a) na += first[a] ? 1 : 0;
    const int grid_sc = std::min(ntiles, vx_cu_count());
    for (int p = 0; p < npass; ++p) {
        VxArrays pk;
        int k = 0;
        for (int a = 0; a < 3; ++a) {
            pk.in[a] = nullptr;
            pk.out[a] = nullptr;
        }
        for (int a = 0; a < 3; ++a)
            if (first[a]) {
                pk.in[k] = pay[a];
                pk.out[k] = buf[p & 1][a];
                ++k;
            }
        const VxDev* d = plans + p;
        vx_hist<<<ntiles, kVxThreads, 0, c->stream>>>(pk.in[0], (int)n, d, tab);
        vx_colsum<<<dim3((unsigned)nsegs, (unsigned)(kVxMaxBins / 256)), 256, 0, c->stream>>>(tab, ntiles, d, seg_tot);
        vx_colscan<<<1, 1024, 0, c->stream>>>(seg_tot, nsegs, (int)n, d, bucket_start, ctl);
        if (na == 1) vx_scatter<1><<<grid_sc, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, d, tab, seg_tot, bucket_start, ctl);
        else if (na == 2) vx_scatter<2><<<grid_sc, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, d, tab, seg_tot, bucket_start, ctl);
        else vx_scatter<3><<<grid_sc, kVxThreads, 0, c->stream>>>(pk, (int)n, ntiles, d, tab, seg_tot, bucket_start, ctl);
        for (int a = 0; a < 3; ++a)
            if (first[a]) pay[a] = buf[p & 1][a];
    }
    KCHK(c);
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
    const Pay3* pay[3];
    const uint32_t* skeys;
    static const bool no_wide = std::getenv("MI_ICP_NO_WIDE_VOXEL_SORT") != nullptr;  // A/B switch
    if (L == 0 && n >= (1 << 17) && n <= ((int64_t)1 << 26) && bits >= 12 && !no_wide) {
        // a large cloud, the key sorted whole: 11-bit digits, the keys made once, from the sorted points
        TRY(voxel_wide_sort(c, first, n, g, bits, pay));
        voxel_keys32<<<blocks_for(n), 256, 0, c->stream>>>(reinterpret_cast<const float*>(pay[0]), n, g, keys[0]);
        KCHK(c);
        skeys = keys[0];
    } else {
        voxel_keys32<<<blocks_for(n), 256, 0, c->stream>>>(dp, n, g, keys[0]);
        KCHK(c);
        Pay3* scratch[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
        for (int set = 0; set < std::min(passes, 2); ++set)
            for (int a = 0; a < 3; ++a)
                if (first[a]) TRY(ensure(c, c->vpay[set * 3 + a], (size_t)n, &scratch[set][a]));
        const int cur = radix_sort_payload32(c->stream, keys, first, scratch, sb.hist, sb.scan_tmp, n, L, bits, pay);
        KCHK(c);
        skeys = keys[cur];
    }
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
    } else if (n <= 16 * nvox) {  // a run is a voxel, and a short one: a thread each
        voxel_means_thread<<<blocks_for(nvox), 256, 0, c->stream>>>(pay[0], pay[1], pay[2], run_start, nvox, op, dn ? on : nullptr,
                                                                    dcol ? oc : nullptr);
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
    c->last_voxel_path = -1;
    if (n < 0 || n > 0x7fffff00ll) return fail(c, MI_ICP_ERR_INVALID, "voxel_downsample: bad size");
    if (n == 0 || !(voxel > 0.0f)) return MI_ICP_OK;  // down_sample.cu:173-176
    if (!xyz || !out_xyz || (normals && !out_normals) || (colors && !out_colors))
        return fail(c, MI_ICP_ERR_INVALID, "voxel_downsample: null buffer");

    const float *dp, *dn, *dcol;
    TRY(to_device(c, xyz, (size_t)n * 3, mem_kind, c->stage[0], &dp));
    TRY(to_device(c, normals, (size_t)n * 3, mem_kind, c->stage[1], &dn));
    TRY(to_device(c, colors, (size_t)n * 3, mem_kind, c->stage[2], &dcol));

    // a dense grid: one move of every point (voxel_dense.h), started behind the bounds without waiting for them; the
    // bounds come back with its control words
    bool dense = false;
    float *dop = nullptr, *don = nullptr, *doc = nullptr;
    // (a context whose last call with this voxel size and a cloud of about this size was turned away by the plan -- a grid
    // of too many or too few cells -- does not try again: the attempt is seven launches that do nothing, ~25 us in front
    // of the general path.  Speed only; a stream of scans of one scene is the case in mind.)
    const bool turned_away = c->vx_refused_voxel == voxel && n >= c->vx_refused_n / 2 && n <= c->vx_refused_n * 2;
    if (!turned_away) TRY(voxel_dense_launch(c, dp, dn, dcol, n, voxel, out_xyz, out_normals, out_colors, mem_kind, &dense, &dop, &don, &doc));
    if (!dense) {
        float* bnd;
        TRY(compute_bounds(c, dp, n, &bnd));
        HIPCHK(c, hipMemcpyAsync(c->f_host, bnd, 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dense) {
        std::memcpy(c->f_host, c->u_host + kVxCtlBounds, 6 * sizeof(float));
        if (c->u_host[0] == 2u) {
            c->vx_refused_voxel = voxel;
            c->vx_refused_n = n;
        } else {
            c->vx_refused_n = 0;
        }
    }
    if (dense && c->u_host[0] == 0u) {  // (1: the cloud crowds into a few buckets, 2: not a grid for that path -- nothing was written)
        const int64_t nvox = (int64_t)c->u_host[2];
        if (mem_kind == MI_ICP_HOST) {
            TRY(from_device(c, (const float*)dop, out_xyz, (size_t)nvox * 3, mem_kind));
            if (dn) TRY(from_device(c, (const float*)don, out_normals, (size_t)nvox * 3, mem_kind));
            if (dcol) TRY(from_device(c, (const float*)doc, out_colors, (size_t)nvox * 3, mem_kind));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        *m = nvox;
        c->last_voxel_path = 1;
        return MI_ICP_OK;
    }
    VoxelGrid g;
    float ext = 0.0f;
    int bits[3];
    {
        const float* b = c->f_host;
        const float origin[3] = {b[0] - voxel * 0.5f, b[1] - voxel * 0.5f, b[2] - voxel * 0.5f};
        for (int d = 0; d < 3; ++d) ext = std::fmax(ext, (b[3 + d] + voxel * 0.5f) - origin[d]);
        if (voxel * (float)INT32_MAX < ext) return MI_ICP_OK;  // down_sample.cu:186-189
        c->last_voxel_path = 0;
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

    // (grids whose packed key needs more than 32 bits keep the first form below: 64-bit keys + indices, one gather)
    if (bits[0] + bits[1] + bits[2] <= 32)
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
    double *sums, *rows;  // the 32 totals; the rows of od_accumulate's larger grids (the ICP reduction's row buffer: transient there too)
    TRY(ensure(c, c->sys_dev, kSysSize, &sums));
    TRY(ensure(c, c->partial, (size_t)kSysSize * kOdMaxBlocks, &rows));
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
    a.rows = rows;
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
        return (int)std::min<int64_t>(kOdMaxBlocks, std::max<int64_t>(1, (n + kOdThreads - 1) / kOdThreads));
    };
    // rows left by an evaluation of level l for whoever consumes its sums (0: it added to the totals itself)
    auto rows_of = [&](int l) { const int g = grid_for(l); return g > kOdAtomicBlocks ? g : 0; };
    {   // NormalizeIntensity (:416-436) over the correspondences under odo_init
        TRY(set_T(init));
        od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[0], 0, rows, 0);
        level_args(0);
        od_accumulate<kOdMeans><<<grid_for(0), kOdThreads, 0, c->stream>>>(a);
        if (rows_of(0)) od_total<<<1, kOdStepThreads, 0, c->stream>>>(rows, rows_of(0), sums);
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
        od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[L - 1], 0, rows, 0);  // terms for the coarsest level; zeroes the sums
    }
    for (int level = L - 1; level >= 0; --level) {
        level_args(level);
        const int iters = option->iterations[L - level - 1];
        for (int iter = 0; iter < iters; ++iter) {
            // the next evaluation: this level again, the next finer one, or level 0 (information matrix)
            const int next = (iter + 1 < iters) ? level : std::max(level - 1, 0);
            if (weighted) {  // two passes: the weights' normalisation, then the weighted system
                od_accumulate<kOdWeightSum><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
                od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[level], 3, rows, rows_of(level));
                od_accumulate<kOdWeighted><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
                od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[next], 2, rows, rows_of(level));
                continue;
            }
            if (jacobian == MI_ICP_ODOMETRY_COLOR_TERM) od_accumulate<kOdColor><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
            else od_accumulate<kOdHybrid><<<grid_for(level), kOdThreads, 0, c->stream>>>(a);
            od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[next], 1, rows, rows_of(level));
        }
        if (iters <= 0 && level > 0) od_step<<<1, kOdStepThreads, 0, c->stream>>>(state, sums, cam[level - 1], 0, rows, 0);
    }
    KCHK(c);
    // CreateInformationMatrix (:349-394): I + sum G^T G over the final correspondences
    level_args(0);
    od_accumulate<kOdInformation><<<grid_for(0), kOdThreads, 0, c->stream>>>(a);
    if (rows_of(0)) od_total<<<1, kOdStepThreads, 0, c->stream>>>(rows, rows_of(0), sums);
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

}  // extern "C"
