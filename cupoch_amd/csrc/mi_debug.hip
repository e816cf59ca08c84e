// mi_debug.hip -- include/mi_icp_debug.h: test-only entry points (sort, scan, traversal census, tree / region / halo
// export, occupancy, the step's two solvers side by side)
// (one translation unit of libmi_icp.so; csrc/ctx.h lists them)
#include "ctx.h"
#include <vector>
#include "halo_format.h"
#include "traverse.h"
#include "wave_solver.h"
#include "eigen3.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

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
__host__ __device__ inline void eigen3_one(const float* A9, float* eval, float* evec, float* S) {
    M3 A;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A.m[r][c] = A9[r * 3 + c];
    float ev[3], e[3][3], s[3][3];
    fast_eigen3x3(A, ev, e);
    gicp_weight(A, s);
    for (int k = 0; k < 3; ++k) {
        if (eval) eval[k] = ev[k];
        for (int d = 0; d < 3; ++d) {
            if (evec) evec[d * 3 + k] = e[k][d];  // column k = eigenvector k
            if (S) S[k * 3 + d] = s[k][d];
        }
    }
}
__global__ __launch_bounds__(256) void eigen3_kernel(const float* A, int64_t n, float* eval, float* evec, float* S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eigen3_one(A + 9 * i, eval ? eval + 3 * i : nullptr, evec ? evec + 9 * i : nullptr, S ? S + 9 * i : nullptr);
}
}  // namespace mi

extern "C" {

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
    const size_t count = (size_t)c->nleaf * kHaloStored * kHaloLineFloats;
    if (!c->thalo.p) {  // no halos on this tree (MI_ICP_NO_CELLS / MI_ICP_NO_LINKS)
        std::memset(halos_out, 0, count * sizeof(float));
        return MI_ICP_OK;
    }
    HIPCHK(c, hipMemcpyAsync(halos_out, c->thalo.p, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MI_ICP_OK;
}

int mi_icp_debug_last_search_kind(const mi_icp_ctx* c) { return c ? c->last_search_kind : -1; }
int mi_icp_debug_last_voxel_path(const mi_icp_ctx* c) { return c ? c->last_voxel_path : -1; }

int mi_icp_debug_occupancy(int which) {
    if (which == 0 || which == 4) return occupancy_build(which);
    if (which >= 1 && which <= 3) return occupancy_loop(which);
    if (which >= 5 && which <= 8) return occupancy_geometry(which);
    return -1;
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

int mi_icp_debug_set_step_stamps(mi_icp_ctx* c, int enable) {
    TRY(check_ctx(c));
    c->stamps_on = enable != 0;  // (takes effect with the next mi_icp_icp_begin / mi_icp_registration_icp)
    return MI_ICP_OK;
}

int mi_icp_debug_get_step_stamps(mi_icp_ctx* c, uint64_t* out32, double* ticks_per_us) {
    TRY(check_ctx(c));
    if (!out32 || !c->stamps.p) return fail(c, MI_ICP_ERR_STATE, "debug_get_step_stamps: no stamped loop has run on this context");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out32, c->stamps.p, sizeof(uint64_t) * kStampWords, hipMemcpyDeviceToHost));
    if (ticks_per_us) {
        int khz = 0;
        *ticks_per_us = (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) ? khz / 1000.0 : 100.0;
    }
    return MI_ICP_OK;
}

int mi_icp_debug_loop_counters(mi_icp_ctx* c, int32_t* out4) {
    TRY(check_ctx(c));
    if (!out4 || !c->loop_host) return fail(c, MI_ICP_ERR_INVALID, "debug_loop_counters: bad arguments");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // (the pinned mirror of the loop's device state as of the last look: every loop entry point ends with one)
    out4[0] = c->loop_host->iterations;
    out4[1] = c->loop_host->passes;
    out4[2] = c->loop_host->relocations;
    out4[3] = c->relocate_armed ? 1 : 0;
    return MI_ICP_OK;
}

int mi_icp_debug_drop_seeds(mi_icp_ctx* c) {
    TRY(check_ctx(c));
    c->nn_valid = false;
    return MI_ICP_OK;
}

int mi_icp_debug_locate(mi_icp_ctx* c, const float* T, int32_t* leaf_out) {
    TRY(check_ctx(c));
    if (!leaf_out || c->ns <= 0 || c->nt <= 0) return fail(c, MI_ICP_ERR_INVALID, "debug_locate: bad state/arguments");
    if (!planes_available(c)) return fail(c, MI_ICP_ERR_INVALID, "debug_locate: this tree has no split planes");
    TRY(launch_locate_by_planes(c, make_xform(load_T(T)), nullptr, 0));
    c->nn_valid = true;  // (the seeds of the next seeded pass)
    c->n_user_pairs = -1;
    std::vector<int32_t> seeds((size_t)c->ns), perm((size_t)c->ns);
    HIPCHK(c, hipMemcpyAsync(seeds.data(), c->nn_idx.p, sizeof(int32_t) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(perm.data(), c->sperm.p, sizeof(int32_t) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < c->ns; ++i) leaf_out[perm[(size_t)i]] = seeds[(size_t)i] >> 3;
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

int mi_icp_debug_eigen3(int device, const float* A, int64_t n, float* eval, float* evec, float* S) {
    if (n < 0 || (n > 0 && !A)) return MI_ICP_ERR_INVALID;
    if (n == 0) return MI_ICP_OK;
    if (device < 0) {
        for (int64_t i = 0; i < n; ++i)
            eigen3_one(A + 9 * i, eval ? eval + 3 * i : nullptr, evec ? evec + 9 * i : nullptr, S ? S + 9 * i : nullptr);
        return MI_ICP_OK;
    }
    if (hipSetDevice(device) != hipSuccess) return MI_ICP_ERR_NO_DEVICE;
    float *dA = nullptr, *de = nullptr, *dv = nullptr, *dS = nullptr;
    int rc = MI_ICP_ERR_HIP;
    if (hipMalloc(&dA, (size_t)n * 36) == hipSuccess && hipMalloc(&de, (size_t)n * 12) == hipSuccess &&
        hipMalloc(&dv, (size_t)n * 36) == hipSuccess && hipMalloc(&dS, (size_t)n * 36) == hipSuccess &&
        hipMemcpy(dA, A, (size_t)n * 36, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(eigen3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dA, n, de, dv, dS);
        bool ok = hipDeviceSynchronize() == hipSuccess;
        if (ok && eval) ok = hipMemcpy(eval, de, (size_t)n * 12, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && evec) ok = hipMemcpy(evec, dv, (size_t)n * 36, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && S) ok = hipMemcpy(S, dS, (size_t)n * 36, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) rc = MI_ICP_OK;
    }
    (void)hipFree(dA);
    (void)hipFree(de);
    (void)hipFree(dv);
    (void)hipFree(dS);
    return rc;
}

}  // extern "C"
