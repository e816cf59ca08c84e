// mi_knn.hip -- k-nearest-neighbour work on the target tree: PointCloud::EstimateNormals, KDTreeFlann::SearchKNN /
// SearchRadius, Colored ICP's colour gradients and its registration entry (knn_normals.h)
// (one translation unit of libmi_icp.so; csrc/ctx.h lists them)
#include "ctx.h"
#include "knn_normals.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

namespace {

// The index rows of one k-NN launch (knn_normals.h KnnSlab): a quarter more rows per XCD than the XCD can hold waves of
// this kernel, flags cleared on the stream ahead of the launch.  `kernel`: the instantiation about to be launched.
template <class K>
int knn_slab(mi_icp_ctx* c, K kernel, int cap, KnnSlab* out) {
    static const int ncu = [] { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); return (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, knn_waves(cap) * 64, 0) != hipSuccess || occ <= 0) {
        (void)hipGetLastError();
        occ = std::min(32, (160 * 1024) / (cap * 64 * 4));  // one wave per workgroup; the lists' distances fill the LDS
    }
    const uint32_t per_xcc = (uint32_t)(((int64_t)occ * knn_waves(cap) * ((ncu + 7) / 8) * 5 + 3) / 4 + 8);
    TRY(ensure(c, c->knn_idx, (size_t)8 * per_xcc * cap * 64, &out->rows));
    TRY(ensure(c, c->knn_flags, (size_t)8 * per_xcc, &out->flags));
    HIPCHK(c, hipMemsetAsync(out->flags, 0, (size_t)8 * per_xcc * sizeof(uint32_t), c->stream));
    out->per_xcc = per_xcc;
    return MI_ICP_OK;
}

}  // namespace

extern "C" {

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
        KnnSlab slab;
        if (cap == kMaxKnn) TRY(knn_slab(a, knn_normals_kernel<0, kMaxKnn>, cap, &slab));
        else if (cap == kMaxKnnMid) TRY(knn_slab(a, knn_normals_kernel<0, kMaxKnnMid>, cap, &slab));
        else TRY(knn_slab(a, knn_normals_kernel<0, kMaxKnnBig>, cap, &slab));
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
    TRY(ensure(c, c->flags, 16, (unsigned long long**)&cnt));
    HIPCHK(c, hipMemsetAsync(cnt, 0, 16 * sizeof(unsigned long long), c->stream));
    const uint32_t npackets = (uint32_t)((nq + 63) / 64);
    const int cap = knn_capacity(knn), waves = knn_waves(cap);
    const uint32_t nblocks = (npackets + waves - 1) / waves;
    const uint32_t grid = ((nblocks + 7u) / 8u) * 8u;
    KnnSlab slab;
    if (cap == kMaxKnn) TRY(knn_slab(c, knn_search_kernel<kMaxKnn>, cap, &slab));
    else if (cap == kMaxKnnMid) TRY(knn_slab(c, knn_search_kernel<kMaxKnnMid>, cap, &slab));
    else TRY(knn_slab(c, knn_search_kernel<kMaxKnnBig>, cap, &slab));
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
    HIPCHK(c, hipMemcpyAsync(c->sys_host, cnt, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (found) *found = (int64_t) * reinterpret_cast<unsigned long long*>(c->sys_host);
#ifdef MI_KNN_CENSUS
    {   // (a census build: scripts/dev/knn_census.sh)
        const unsigned long long* u = reinterpret_cast<const unsigned long long*>(c->sys_host);
        const double* d = reinterpret_cast<const double*>(c->sys_host);
        const double p = (double)std::max<unsigned long long>(u[6], 1ull), ln = std::max(d[9], 1.0);
        std::fprintf(stderr, "knn census k=%d: per packet: leaves offered %.1f, candidates some lane accepted %.1f (%.1f %% of the offered), lane-accepts %.1f, "
                     "leaves after which a cube shrank %.1f, records %.1f; mean bound before the walk %.3g (packet max %.3g, min %.3g), after %.3g; lanes that walked alone %.2f\n",
                     knn, u[1] / p, u[2] / p, 100.0 * u[2] / std::max<double>(8.0 * u[1], 1.0), u[3] / p, u[4] / p, u[5] / p, d[7] / ln, d[10] / p, d[11] / p, d[8] / ln, u[12] / p);
    }
#endif
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
    KnnSlab slab;
    if (cap == kMaxKnn) TRY(knn_slab(c, knn_normals_kernel<1, kMaxKnn>, cap, &slab));
    else if (cap == kMaxKnnMid) TRY(knn_slab(c, knn_normals_kernel<1, kMaxKnnMid>, cap, &slab));
    else TRY(knn_slab(c, knn_normals_kernel<1, kMaxKnnBig>, cap, &slab));
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

}  // extern "C"
