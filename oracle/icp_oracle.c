/*
 * icp_oracle.c -- CPU restatement of cupoch's ICP registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cupoch_amd/ may call, link or
 * import this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line (relative to the cupoch tree,
 * v0.2.11.0) whose behaviour it restates.  Nothing here is copied: the
 * reference is thrust functors + Eigen + FLANN's CUDA kd-tree; this is plain
 * C99 with an exact kd-tree written for the purpose.
 *
 * Parity pinning (see oracle/README.md, DESIGN.md section "Oracle"):
 *   - kNN / radius / 1-NN        : pinned by the reference's golden vectors
 *                                  (src/tests/knn/kdtree_flann.cpp:47-135,
 *                                   src/tests/knn/lbvh_knn.cpp:47-86) and by
 *                                  the reference's own FLANN CPU kd-tree
 *                                  compiled into oracle/_ref.
 *   - Kabsch, Transform, VoxelDownSample, EstimateNormals
 *                                : pinned by src/tests/registration/kabsch.cpp
 *                                  and src/tests/geometry/pointcloud.cpp.
 *   - RegistrationICP loop, JtJ/Jtr accumulation, 6x6 LDLT solve, GICP,
 *     Colored ICP (colour gradients, two-row estimator), depth-image ->
 *     point-cloud factories, KinFu pose estimation
 *                                : PARITY UNPINNED by the reference's own
 *                                  tests (none exist); checked by
 *                                  self-consistency (recover a known T_gt)
 *                                  and, since round 6, against outside
 *                                  implementations: LAPACK through numpy for
 *                                  the eigen-solver / GICP weight / det /
 *                                  LDLT / colour-gradient fit, and a second
 *                                  loop on scipy's cKDTree (icp_numpy.py;
 *                                  tests/test_outside_checks.py).
 *
 * Arithmetic conventions (the reference is compiled with --use_fast_math, so
 * it defines no bit-exact order itself):
 *   - all per-point quantities are fp32, as in the reference;
 *   - squared distance is  d2 = fma(dz,dz, fma(dy,dy, dx*dx)), d = q - p;
 *   - R*p+t is  fma(R02,z, fma(R01,y, R00*x)) + t   (product, then add);
 *   - sums over correspondences accumulate in fp64 (the reference tree-sums in
 *     fp32; fp64 is what that sum approximates);
 *   - the 6x6 solve is fp32 LDLT with diagonal pivoting, as Eigen's ldlt().
 * The HIP engine uses the same per-point expressions, so nearest-neighbour
 * indices and d2 are comparable bit for bit.
 */
#define _POSIX_C_SOURCE 200809L
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

/* T is column-major 4x4, i.e. Eigen::Matrix4f::data(): T[c*4 + r]. */
#define TM(T, r, c) ((T)[(c) * 4 + (r)])

static inline float dist2f(const float *q, const float *p) {
    const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

static inline void rot3(const float *T, const float *p, float *o) {
    const float x = p[0], y = p[1], z = p[2];
    o[0] = fmaf(TM(T, 0, 2), z, fmaf(TM(T, 0, 1), y, TM(T, 0, 0) * x));
    o[1] = fmaf(TM(T, 1, 2), z, fmaf(TM(T, 1, 1), y, TM(T, 1, 0) * x));
    o[2] = fmaf(TM(T, 2, 2), z, fmaf(TM(T, 2, 1), y, TM(T, 2, 0) * x));
}

/* geometry_utils.cu:34-44  transform_points_functor: R*p + t */
static inline void xform_point(const float *T, const float *p, float *o) {
    float r[3];
    rot3(T, p, r);
    o[0] = r[0] + TM(T, 0, 3);
    o[1] = r[1] + TM(T, 1, 3);
    o[2] = r[2] + TM(T, 2, 3);
}

static void mat4_identity(float *T) {
    memset(T, 0, 16 * sizeof(float));
    T[0] = T[5] = T[10] = T[15] = 1.0f;
}

/* C = A * B, fp32, column-major (registration.cu:159 update * transformation) */
static void mat4_mul(const float *A, const float *B, float *C) {
    float out[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += TM(A, r, k) * TM(B, k, c);
            out[c * 4 + r] = s;
        }
    memcpy(C, out, sizeof(out));
}

/* Eigen's MatrixBase::isIdentity(prec = 1e-5f for float), used at
 * registration.cu:114,148.  Off-diagonal: |x| <= prec*1 ; diagonal:
 * |x-1| <= prec*min(|x|,1). */
static int mat4_is_identity(const float *T) {
    const float prec = 1e-5f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            const float v = TM(T, r, c);
            if (r == c) {
                const float m = fminf(fabsf(v), 1.0f);
                if (!(fabsf(v - 1.0f) <= prec * m)) return 0;
            } else {
                if (!(fabsf(v) <= prec)) return 0;
            }
        }
    return 1;
}

ORACLE_API int oracle_mat4_is_identity(const float *T) {
    return mat4_is_identity(T);
}

/* ------------------------------------------------------------------ */
/* PointCloud::Transform  (pointcloud.cu:293-299,                      */
/* geometry_utils.cu:34-52,257-265)                                    */
/* ------------------------------------------------------------------ */

ORACLE_API void oracle_transform_points(const float *T, float *pts, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float o[3];
        xform_point(T, pts + 3 * i, o);
        memcpy(pts + 3 * i, o, sizeof(o));
    }
}

/* normals: n <- R*n, no renormalisation (geometry_utils.cu:45-52) */
ORACLE_API void oracle_transform_normals(const float *T, float *nrm, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float o[3];
        rot3(T, nrm + 3 * i, o);
        memcpy(nrm + 3 * i, o, sizeof(o));
    }
}

/* covariances: C <- R*C*R^T, column-major 3x3 (geometry_utils.cu:257-265) */
static void rotate_cov(const float *T, const float *C, float *out) {
    float RC[9]; /* column-major: RC[c*3+r] */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            RC[c * 3 + r] = fmaf(TM(T, r, 2), C[c * 3 + 2],
                                 fmaf(TM(T, r, 1), C[c * 3 + 1],
                                      TM(T, r, 0) * C[c * 3 + 0]));
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            out[c * 3 + r] = fmaf(RC[2 * 3 + r], TM(T, c, 2),
                                  fmaf(RC[1 * 3 + r], TM(T, c, 1),
                                       RC[0 * 3 + r] * TM(T, c, 0)));
}

ORACLE_API void oracle_rotate_covariances(const float *T, float *covs, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float o[9];
        rotate_cov(T, covs + 9 * i, o);
        memcpy(covs + 9 * i, o, sizeof(o));
    }
}

/* ------------------------------------------------------------------ */
/* Exact kd-tree kNN / radius search.                                  */
/* Restates the *results* of knn::KDTreeFlann (kdtree_flann.inl:46-144) */
/* over flann::KDTreeCuda3dIndex: exact neighbours (checks=-1, eps=0), */
/* results sorted by ascending distance, radius accept test is strict  */
/* d2 < r2 with r2 = radius*radius in fp32 (kdtree_flann.inl:119-120;  */
/* util/cuda/result_set.h:385,401), unfilled slots idx=-1, d2=+inf     */
/* (result_set.h:447-459).  Ties (equal d2) are broken towards the      */
/* lower target index; the reference keeps the first-visited point,    */
/* which is traversal-order dependent.                                 */
/* ------------------------------------------------------------------ */

#define KD_LEAF 12

typedef struct {
    int lo, hi;      /* range in perm */
    int left, right; /* children or -1 */
    int dim;
    float split;
    float bmin[3], bmax[3];
} kd_node;

typedef struct {
    const float *pts;
    int n;
    int *perm;
    kd_node *nodes;
    int n_nodes, cap_nodes;
} kd_tree;

static int kd_new_node(kd_tree *t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

/* quickselect on perm[lo,hi) by coordinate dim so that perm[k] is in place */
static void kd_select(kd_tree *t, int lo, int hi, int k, int dim) {
    const float *P = t->pts;
    int *a = t->perm;
    int l = lo, r = hi - 1;
    while (l < r) {
        const float pv = P[3 * a[(l + r) / 2] + dim];
        int i = l, j = r;
        while (i <= j) {
            while (P[3 * a[i] + dim] < pv) ++i;
            while (P[3 * a[j] + dim] > pv) --j;
            if (i <= j) {
                int tmp = a[i];
                a[i] = a[j];
                a[j] = tmp;
                ++i;
                --j;
            }
        }
        if (k <= j)
            r = j;
        else if (k >= i)
            l = i;
        else
            break;
    }
}

static int kd_build_rec(kd_tree *t, int lo, int hi) {
    const int id = kd_new_node(t);
    float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; ++i) {
        const float *p = t->pts + 3 * t->perm[i];
        for (int d = 0; d < 3; ++d) {
            if (p[d] < bmin[d]) bmin[d] = p[d];
            if (p[d] > bmax[d]) bmax[d] = p[d];
        }
    }
    int dim = 0;
    float ext = bmax[0] - bmin[0];
    for (int d = 1; d < 3; ++d)
        if (bmax[d] - bmin[d] > ext) {
            ext = bmax[d] - bmin[d];
            dim = d;
        }
    kd_node nd;
    nd.lo = lo;
    nd.hi = hi;
    nd.left = nd.right = -1;
    nd.dim = dim;
    nd.split = 0.0f;
    memcpy(nd.bmin, bmin, sizeof(bmin));
    memcpy(nd.bmax, bmax, sizeof(bmax));
    if (hi - lo > KD_LEAF && ext > 0.0f) {
        const int mid = lo + (hi - lo) / 2;
        kd_select(t, lo, hi, mid, dim);
        nd.split = t->pts[3 * t->perm[mid] + dim];
        t->nodes[id] = nd;
        const int l = kd_build_rec(t, lo, mid);
        const int r = kd_build_rec(t, mid, hi);
        t->nodes[id].left = l;
        t->nodes[id].right = r;
    } else {
        t->nodes[id] = nd;
    }
    return id;
}

static kd_tree *kd_build(const float *pts, int n) {
    kd_tree *t = (kd_tree *)calloc(1, sizeof(kd_tree));
    t->pts = pts;
    t->n = n;
    t->perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) t->perm[i] = i;
    if (n > 0) kd_build_rec(t, 0, n);
    return t;
}

static void kd_free(kd_tree *t) {
    if (!t) return;
    free(t->perm);
    free(t->nodes);
    free(t);
}

typedef struct {
    int k, count;
    float r2;  /* strict upper bound on accepted d2 (INFINITY for plain kNN) */
    float *d2; /* ascending */
    int *idx;
} kd_result;

static inline float kd_worst(const kd_result *r) {
    return (r->count < r->k) ? r->r2 : r->d2[r->k - 1];
}

static inline void kd_offer(kd_result *r, float d2, int j) {
    if (!(d2 < r->r2)) return; /* strict radius test */
    if (r->count == r->k) {
        const float w = r->d2[r->k - 1];
        if (d2 > w || (d2 == w && j > r->idx[r->k - 1])) return;
    }
    int pos = (r->count < r->k) ? r->count : r->k - 1;
    while (pos > 0 && (r->d2[pos - 1] > d2 ||
                       (r->d2[pos - 1] == d2 && r->idx[pos - 1] > j))) {
        r->d2[pos] = r->d2[pos - 1];
        r->idx[pos] = r->idx[pos - 1];
        --pos;
    }
    r->d2[pos] = d2;
    r->idx[pos] = j;
    if (r->count < r->k) r->count++;
}

static inline float kd_box_d2(const kd_node *nd, const float *q) {
    float d[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = nd->bmin[a] - q[a], hi = q[a] - nd->bmax[a];
        d[a] = fmaxf(fmaxf(lo, hi), 0.0f);
    }
    return fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
}

static void kd_search_rec(const kd_tree *t, int id, const float *q, kd_result *r) {
    const kd_node *nd = &t->nodes[id];
    /* <= so that an equal-distance point with a lower index is still found */
    if (!(kd_box_d2(nd, q) <= kd_worst(r))) return;
    if (nd->left < 0) {
        for (int i = nd->lo; i < nd->hi; ++i) {
            const int j = t->perm[i];
            kd_offer(r, dist2f(q, t->pts + 3 * j), j);
        }
        return;
    }
    if (q[nd->dim] < nd->split) {
        kd_search_rec(t, nd->left, q, r);
        kd_search_rec(t, nd->right, q, r);
    } else {
        kd_search_rec(t, nd->right, q, r);
        kd_search_rec(t, nd->left, q, r);
    }
}

/* Generic search: k nearest with d2 < r2 (r2 = INFINITY -> plain kNN).
 * Output row-major [nq][k]; unfilled slots idx=-1, d2=+inf.
 * Returns the number of filled slots (KDTreeFlann::Search* return value). */
static int64_t kd_search_all(const kd_tree *t, const float *qry, int64_t nq, int k,
                             float r2, int *idx, float *d2) {
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : total)
    for (int64_t i = 0; i < nq; ++i) {
        kd_result r;
        r.k = k;
        r.count = 0;
        r.r2 = r2;
        r.d2 = d2 + i * k;
        r.idx = idx + i * k;
        if (t->n > 0) kd_search_rec(t, 0, qry + 3 * i, &r);
        for (int s = r.count; s < k; ++s) {
            r.idx[s] = -1;
            r.d2[s] = INFINITY;
        }
        total += r.count;
    }
    return total;
}

/* KDTreeFlann::SearchKNN (kdtree_flann.inl:46-94) */
ORACLE_API int64_t oracle_search_knn(const float *tgt, int64_t nt, const float *qry,
                                     int64_t nq, int k, int *idx, float *d2) {
    if (nt <= 0 || nq <= 0 || k < 0) return -1; /* kdtree_flann.cu:52-54 */
    if (k == 0) return 0;
    kd_tree *t = kd_build(tgt, (int)nt);
    const int64_t c = kd_search_all(t, qry, nq, k, INFINITY, idx, d2);
    kd_free(t);
    return c;
}

/* KDTreeFlann::SearchRadius (kdtree_flann.inl:96-122): the k = max_nn
 * nearest points with d2 < radius^2 (strict), r2 rounded to fp32. */
ORACLE_API int64_t oracle_search_radius(const float *tgt, int64_t nt, const float *qry,
                                        int64_t nq, float radius, int max_nn,
                                        int *idx, float *d2) {
    if (nt <= 0 || nq <= 0 || max_nn < 0) return -1; /* kdtree_flann.cu:72-73 */
    if (max_nn == 0) return 0;
    const float r2 = radius * radius;
    kd_tree *t = kd_build(tgt, (int)nt);
    const int64_t c = kd_search_all(t, qry, nq, max_nn, r2, idx, d2);
    kd_free(t);
    return c;
}

/* A tree kept across calls, for tests that query one large target many times
 * (a 10M-point build takes seconds).  The handle borrows `tgt`: the caller keeps
 * the array alive until oracle_tree_free. */
ORACLE_API void *oracle_tree_create(const float *tgt, int64_t nt) {
    return (void *)kd_build(tgt, (int)(nt > 0 ? nt : 0));
}

ORACLE_API int64_t oracle_tree_search_radius(const void *tree, const float *qry, int64_t nq,
                                             float radius, int max_nn, int *idx, float *d2) {
    const kd_tree *t = (const kd_tree *)tree;
    if (!t || t->n <= 0 || nq <= 0 || max_nn < 0) return -1;
    if (max_nn == 0) return 0;
    return kd_search_all(t, qry, nq, max_nn, radius * radius, idx, d2);
}

ORACLE_API int64_t oracle_tree_search_knn(const void *tree, const float *qry, int64_t nq, int k,
                                          int *idx, float *d2) {
    const kd_tree *t = (const kd_tree *)tree;
    if (!t || t->n <= 0 || nq <= 0 || k < 0) return -1;
    if (k == 0) return 0;
    return kd_search_all(t, qry, nq, k, INFINITY, idx, d2);
}

ORACLE_API void oracle_tree_free(void *tree) { kd_free((kd_tree *)tree); }

/* Brute-force variant (O(nq*nt)), used to validate the kd-tree itself. */
ORACLE_API int64_t oracle_search_bruteforce(const float *tgt, int64_t nt,
                                            const float *qry, int64_t nq, int k,
                                            float radius_or_neg, int *idx, float *d2) {
    const float r2 = (radius_or_neg < 0.0f) ? INFINITY : radius_or_neg * radius_or_neg;
    int64_t total = 0;
    for (int64_t i = 0; i < nq; ++i) {
        kd_result r;
        r.k = k;
        r.count = 0;
        r.r2 = r2;
        r.d2 = d2 + i * k;
        r.idx = idx + i * k;
        for (int64_t j = 0; j < nt; ++j)
            kd_offer(&r, dist2f(qry + 3 * i, tgt + 3 * j), (int)j);
        for (int s = r.count; s < k; ++s) {
            r.idx[s] = -1;
            r.d2[s] = INFINITY;
        }
        total += r.count;
    }
    return total;
}

/* ------------------------------------------------------------------ */
/* Host linear algebra restating what the reference gets from Eigen    */
/* (third_party/eigen is an empty submodule in the reference checkout; */
/*  call sites: utility/eigen.cu:92,103 ; registration/kabsch.cu:108). */
/* ------------------------------------------------------------------ */

/* fp32 determinant via LU with partial pivoting (Eigen: PartialPivLU for
 * fixed sizes > 4); product of pivots taken in fp32 so that it overflows to
 * +-inf where the reference's does. */
static float det6f(const float *A_colmajor) {
    float a[36];
    memcpy(a, A_colmajor, sizeof(a));
    float det = 1.0f;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabsf(a[k * 6 + k]);
        for (int r = k + 1; r < 6; ++r)
            if (fabsf(a[k * 6 + r]) > best) {
                best = fabsf(a[k * 6 + r]);
                p = r;
            }
        if (best == 0.0f) return 0.0f;
        if (p != k) {
            for (int c = 0; c < 6; ++c) {
                float tmp = a[c * 6 + k];
                a[c * 6 + k] = a[c * 6 + p];
                a[c * 6 + p] = tmp;
            }
            det = -det;
        }
        const float piv = a[k * 6 + k];
        det *= piv;
        for (int r = k + 1; r < 6; ++r) {
            const float f = a[k * 6 + r] / piv;
            for (int c = k + 1; c < 6; ++c) a[c * 6 + r] -= f * a[c * 6 + k];
        }
    }
    return det;
}

/* fp32 LDLT with symmetric diagonal pivoting (what Eigen's ldlt() is),
 * solves A x = b for symmetric A (6x6, column-major). */
static void ldlt6_solve(const float *A_colmajor, const float *b, float *x) {
    float a[6][6];
    int perm[6];
    for (int r = 0; r < 6; ++r) {
        perm[r] = r;
        for (int c = 0; c < 6; ++c) a[r][c] = A_colmajor[c * 6 + r];
    }
    float L[6][6] = {{0}};
    float D[6];
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabsf(a[k][k]);
        for (int i = k + 1; i < 6; ++i)
            if (fabsf(a[i][i]) > best) {
                best = fabsf(a[i][i]);
                p = i;
            }
        if (p != k) { /* symmetric row+column swap, and the computed part of L */
            for (int c = 0; c < 6; ++c) {
                float t = a[k][c];
                a[k][c] = a[p][c];
                a[p][c] = t;
            }
            for (int r = 0; r < 6; ++r) {
                float t = a[r][k];
                a[r][k] = a[r][p];
                a[r][p] = t;
            }
            for (int c = 0; c < k; ++c) {
                float t = L[k][c];
                L[k][c] = L[p][c];
                L[p][c] = t;
            }
            int ti = perm[k];
            perm[k] = perm[p];
            perm[p] = ti;
        }
        D[k] = a[k][k];
        L[k][k] = 1.0f;
        if (D[k] != 0.0f) {
            for (int i = k + 1; i < 6; ++i) L[i][k] = a[i][k] / D[k];
            for (int i = k + 1; i < 6; ++i)
                for (int j = k + 1; j < 6; ++j) a[i][j] -= L[i][k] * D[k] * L[j][k];
        }
    }
    float y[6], z[6];
    for (int i = 0; i < 6; ++i) {
        float s = b[perm[i]];
        for (int j = 0; j < i; ++j) s -= L[i][j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0f) ? y[i] / D[i] : 0.0f;
    for (int i = 5; i >= 0; --i) {
        float s = y[i];
        for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 6; ++i) x[perm[i]] = z[i];
}

/* utility::TransformVector6fToMatrix4f (utility/eigen.cu:28-50):
 * x = [w(3); t(3)], R = Rodrigues(w), theta == 0 -> R = I. */
ORACLE_API void oracle_vector6_to_matrix4(const float *x, float *T) {
    mat4_identity(T);
    TM(T, 0, 3) = x[3];
    TM(T, 1, 3) = x[4];
    TM(T, 2, 3) = x[5];
    const float th = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (th == 0.0f) return;
    const float w0 = x[0] / th, w1 = x[1] / th, w2 = x[2] / th;
    const float c = cosf(th), s = sinf(th);
    TM(T, 0, 0) = c + w0 * w0 * (1 - c);
    TM(T, 0, 1) = w0 * w1 * (1 - c) - w2 * s;
    TM(T, 0, 2) = w1 * s + w0 * w2 * (1 - c);
    TM(T, 1, 0) = w2 * s + w0 * w1 * (1 - c);
    TM(T, 1, 1) = c + w1 * w1 * (1 - c);
    TM(T, 1, 2) = -w0 * s + w1 * w2 * (1 - c);
    TM(T, 2, 0) = -w1 * s + w0 * w2 * (1 - c);
    TM(T, 2, 1) = w0 * s + w1 * w2 * (1 - c);
    TM(T, 2, 2) = c + w2 * w2 * (1 - c);
}

/* utility::SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.cu:107-122)
 * over SolveLinearSystemPSD<6> (utility/eigen.cu:76-105): solves JTJ x = -JTr,
 * det check only when det_thresh > 0.  sys[] is the accumulated system in the
 * layout documented at oracle_compute_system().  Returns 1 on success; on
 * failure T = Identity (transformation_estimation.cu:221). */
ORACLE_API int oracle_solve_system(const double *sys, float det_thresh, float *T) {
    float A[36], b[6], x[6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) {
            A[j * 6 + i] = (float)sys[k];
            A[i * 6 + j] = (float)sys[k];
        }
    for (int i = 0; i < 6; ++i) b[i] = -(float)sys[21 + i];
    if (det_thresh > 0.0f) {
        const float det = det6f(A);
        if (fabsf(det) < det_thresh || isnan(det) || isinf(det)) {
            mat4_identity(T);
            return 0;
        }
    }
    ldlt6_solve(A, b, x);
    oracle_vector6_to_matrix4(x, T);
    return 1;
}

/* 3x3 SVD A = U S V^T by one-sided Jacobi in fp64 (the reference calls
 * Eigen::JacobiSVD<Matrix3f>, kabsch.cu:108-109).  Row-major 3x3 in/out. */
static void svd3(const double A[9], double U[9], double S[3], double V[9]) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[r][c] = A[r * 3 + c];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += a[r][p] * a[r][p];
                    beta += a[r][q] * a[r][q];
                    gamma += a[r][p] * a[r][q];
                }
                off = fmax(off, fabs(gamma) / (sqrt(alpha * beta) + 1e-300));
                if (fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = ((zeta >= 0) ? 1.0 : -1.0) /
                                 (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double x = a[r][p], y = a[r][q];
                    a[r][p] = c * x - s * y;
                    a[r][q] = s * x + c * y;
                    const double vx = v[r][p], vy = v[r][q];
                    v[r][p] = c * vx - s * vy;
                    v[r][q] = s * vx + c * vy;
                }
            }
        if (off < 1e-15) break;
    }
    int order[3] = {0, 1, 2};
    double sv[3];
    for (int c = 0; c < 3; ++c)
        sv[c] = sqrt(a[0][c] * a[0][c] + a[1][c] * a[1][c] + a[2][c] * a[2][c]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[order[j]] > sv[order[i]]) {
                int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    /* A singular value that is ZERO up to rounding (an exactly planar or collinear source: a row of the cross-covariance
     * is zero and its third value comes out ~1e-17 of the largest) has no column of U: a / sv would be a unit vector of
     * noise inside the span of the others, U singular, det(U*V) = 0 and the "rotation" of rank 2.  Eigen's two-sided
     * JacobiSVD (kabsch.cu:108, ComputeFullU | ComputeFullV) builds U and V from rotations and always returns full
     * orthogonal factors, hence a proper rotation; so such a column is completed below instead. */
    const double sv_floor = 1e-10 * sv[order[0]];
    double u[3][3];
    for (int k = 0; k < 3; ++k) {
        const int c = order[k];
        S[k] = sv[c];
        for (int r = 0; r < 3; ++r) {
            V[r * 3 + k] = v[r][c];
            u[r][k] = (sv[c] > 1e-300 && sv[c] > sv_floor) ? a[r][c] / sv[c] : 0.0;
        }
    }
    /* complete U to an orthonormal basis when rank deficient */
    for (int k = 0; k < 3; ++k) {
        double nrm = sqrt(u[0][k] * u[0][k] + u[1][k] * u[1][k] + u[2][k] * u[2][k]);
        if (nrm < 0.5) {
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            double n1 = sqrt(u[0][k1] * u[0][k1] + u[1][k1] * u[1][k1] + u[2][k1] * u[2][k1]);
            double n2 = sqrt(u[0][k2] * u[0][k2] + u[1][k2] * u[1][k2] + u[2][k2] * u[2][k2]);
            if (n1 > 0.5 && n2 > 0.5) {
                u[0][k] = u[1][k1] * u[2][k2] - u[2][k1] * u[1][k2];
                u[1][k] = u[2][k1] * u[0][k2] - u[0][k1] * u[2][k2];
                u[2][k] = u[0][k1] * u[1][k2] - u[1][k1] * u[0][k2];
            } else {
                /* rank <= 1: pick any vector orthogonal to the valid column(s) */
                const int kv = (n1 > 0.5) ? k1 : ((n2 > 0.5) ? k2 : -1);
                double e[3] = {1, 0, 0};
                if (kv >= 0 && fabs(u[0][kv]) > 0.9) {
                    e[0] = 0;
                    e[1] = 1;
                }
                if (kv >= 0) {
                    const double d = e[0] * u[0][kv] + e[1] * u[1][kv] + e[2] * u[2][kv];
                    for (int r = 0; r < 3; ++r) e[r] -= d * u[r][kv];
                }
                const double en = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                for (int r = 0; r < 3; ++r) u[r][k] = e[r] / en;
            }
        }
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) U[r * 3 + c] = u[r][c];
}

static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* Kabsch finish (kabsch.cu:105-118): given centroid sums and the
 * cross-covariance sum already divided as the reference divides them,
 * hh = sum (ps-cs)(pt-ct)^T / n_model,  R = V * diag(1,1,det(U*V)) * U^T,
 * t = ct - R*cs.  hh is row-major 3x3 [source row][target col]. */
static void kabsch_finish(const double hh[9], const double cs[3], const double ct[3],
                          float *T) {
    double U[9], S[3], V[9], UV[9], R[9];
    svd3(hh, U, S, V);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[r * 3 + k] * V[k * 3 + c];
            UV[r * 3 + c] = s;
        }
    const double d = det3(UV); /* kabsch.cu:111 det(U*V) */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k)
                s += V[r * 3 + k] * ((k == 2) ? d : 1.0) * U[c * 3 + k];
            R[r * 3 + c] = s;
        }
    mat4_identity(T);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) TM(T, r, c) = (float)R[r * 3 + c];
        TM(T, r, 3) = (float)(ct[r] - (R[r * 3 + 0] * cs[0] + R[r * 3 + 1] * cs[1] +
                                       R[r * 3 + 2] * cs[2]));
    }
}

/* ------------------------------------------------------------------ */
/* FastEigen3x3 / SqrtMatrix3x3 (utility/eigenvalue.inl:28-177), fp32. */
/* A is symmetric, addressed A(r,c) = a[r][c].  NOTE the reference     */
/* returns the eigenvalues of A / A.maxCoeff() (the scaled matrix) on  */
/* the general branch and of A itself on the diagonal branch; this is  */
/* restated as is (it scales GICP's W per correspondence).             */
/* ------------------------------------------------------------------ */

/* eigenvalue.inl:28 defines signf(x) = x / fabs(x), which is NaN for x == 0.
 * That happens whenever (Ct+Cs)^-1 has an (almost) repeated eigenvalue and
 * m00 or m01 rounds to exactly 0 -- e.g. GICP on a cloud registered against a
 * rigidly moved copy of itself -- and one NaN poisons the whole 6x6 system.
 * DELIBERATE DEVIATION (documented in DESIGN.md): sign(0) := +1.  Any other
 * input gives the reference's value bit for bit. */
static inline float signf_ref(float x) { return copysignf(1.0f, x); }

static void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static float dot3(const float *a, const float *b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* eigenvalue.inl:30-49 */
static void eigvec0(float A[3][3], float eval0, float *out) {
    float row0[3] = {A[0][0] - eval0, A[0][1], A[0][2]};
    float row1[3] = {A[0][1], A[1][1] - eval0, A[1][2]};
    float row2[3] = {A[0][2], A[1][2], A[2][2] - eval0};
    float rxr[3][3], d[3];
    cross3(row0, row1, rxr[0]);
    cross3(row0, row2, rxr[1]);
    cross3(row1, row2, rxr[2]);
    for (int i = 0; i < 3; ++i) d[i] = dot3(rxr[i], rxr[i]);
    int imax = 0;
    if (d[1] > d[imax]) imax = 1;
    if (d[2] > d[imax]) imax = 2;
    const float s = sqrtf(d[imax]);
    for (int i = 0; i < 3; ++i) out[i] = rxr[imax][i] / s;
}

/* eigenvalue.inl:51-91 */
static void eigvec1(float A[3][3], const float *evec0, float eval1, float *out) {
    const float max_evec0_abs = fmaxf(fabsf(evec0[0]), fabsf(evec0[1]));
    const float inv_length =
            1.0f / sqrtf(max_evec0_abs * max_evec0_abs + evec0[2] * evec0[2]);
    float U[3], V[3];
    if (fabsf(evec0[0]) > fabsf(evec0[1])) {
        U[0] = -evec0[2];
        U[1] = 0;
        U[2] = evec0[0];
    } else {
        U[0] = 0;
        U[1] = evec0[2];
        U[2] = -evec0[1];
    }
    for (int i = 0; i < 3; ++i) U[i] *= inv_length;
    cross3(evec0, U, V);
    float AU[3] = {A[0][0] * U[0] + A[0][1] * U[1] + A[0][2] * U[2],
                   A[0][1] * U[0] + A[1][1] * U[1] + A[1][2] * U[2],
                   A[0][2] * U[0] + A[1][2] * U[1] + A[2][2] * U[2]};
    float AV[3] = {A[0][0] * V[0] + A[0][1] * V[1] + A[0][2] * V[2],
                   A[0][1] * V[0] + A[1][1] * V[1] + A[1][2] * V[2],
                   A[0][2] * V[0] + A[1][2] * V[1] + A[2][2] * V[2]};
    const float m00 = dot3(U, AU) - eval1;
    const float m01 = dot3(U, AV);
    const float m11 = dot3(V, AV) - eval1;
    const float absM00 = fabsf(m00), absM01 = fabsf(m01), absM11 = fabsf(m11);
    const float max_abs_comp0 = fmaxf(absM00, absM11);
    const float max_abs_comp = fmaxf(max_abs_comp0, absM01);
    float coef2 = fminf(max_abs_comp0, absM01) / fmaxf(max_abs_comp, 1.0e-6f);
    const float coef1 = 1.0f / sqrtf(1.0f + coef2 * coef2);
    float cu, cv;
    if (absM00 >= absM11) {
        coef2 *= coef1 * signf_ref(m00) * signf_ref(m01);
        if (max_abs_comp0 >= absM01) {
            cu = coef2;
            cv = coef1;
        } else {
            cu = coef1;
            cv = coef2;
        }
    } else {
        coef2 *= coef1 * signf_ref(m11) * signf_ref(m01);
        if (max_abs_comp0 >= absM01) {
            cu = coef1;
            cv = coef2;
        } else {
            cu = coef2;
            cv = coef1;
        }
    }
    for (int i = 0; i < 3; ++i) out[i] = cu * U[i] - cv * V[i];
}

/* eigenvalue.inl:93-154.  evec columns: evec[r][c]. */
static void fast_eigen3x3(float A[3][3], float eval[3], float evec[3][3]) {
    float max_coeff = A[0][0];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (A[r][c] > max_coeff) max_coeff = A[r][c]; /* signed max, :100 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) evec[r][c] = (r == c) ? 1.0f : 0.0f;
    if (max_coeff == 0) {
        eval[0] = eval[1] = eval[2] = 0.0f;
        return;
    }
    float S[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[r][c] = A[r][c] / max_coeff;
    const float norm = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    if (norm > 0) {
        const float q = (S[0][0] + S[1][1] + S[2][2]) / 3;
        const float b00 = S[0][0] - q, b11 = S[1][1] - q, b22 = S[2][2] - q;
        const float p = sqrtf((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        const float c00 = b11 * b22 - S[1][2] * S[1][2];
        const float c01 = S[0][1] * b22 - S[1][2] * S[0][2];
        const float c02 = S[0][1] * S[1][2] - b11 * S[0][2];
        const float det = (b00 * c00 - S[0][1] * c01 + S[0][2] * c02) / (p * p * p);
        float half_det = det * 0.5f;
        half_det = fminf(fmaxf(half_det, -1.0f), 1.0f);
        const float angle = acosf(half_det) / 3.0f;
        const float two_thirds_pi = 2.09439510239319549f;
        const float beta2 = cosf(angle) * 2;
        const float beta0 = cosf(angle + two_thirds_pi) * 2;
        const float beta1 = -(beta0 + beta2);
        eval[0] = q + p * beta0;
        eval[1] = q + p * beta1;
        eval[2] = q + p * beta2;
        float e0[3], e1[3], e2[3];
        if (half_det >= 0) {
            eigvec0(S, eval[2], e2);
            eigvec1(S, e2, eval[1], e1);
            cross3(e1, e2, e0);
        } else {
            eigvec0(S, eval[0], e0);
            eigvec1(S, e0, eval[1], e1);
            cross3(e0, e1, e2);
        }
        for (int r = 0; r < 3; ++r) {
            evec[r][0] = e0[r];
            evec[r][1] = e1[r];
            evec[r][2] = e2[r];
        }
    } else {
        eval[0] = A[0][0];
        eval[1] = A[1][1];
        eval[2] = A[2][2];
    }
}

/* SqrtMatrix3x3 (eigenvalue.inl:172-177): V * diag(sqrt(eval)) * V^T */
static void sqrt_matrix3x3(float A[3][3], float W[3][3]) {
    float eval[3], evec[3][3];
    fast_eigen3x3(A, eval, evec);
    float s[3] = {sqrtf(eval[0]), sqrtf(eval[1]), sqrtf(eval[2])};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            W[r][c] = evec[r][0] * s[0] * evec[c][0] + evec[r][1] * s[1] * evec[c][1] +
                      evec[r][2] * s[2] * evec[c][2];
}

/* Eigen 3x3 inverse by cofactors (generalized_icp.cu:91 (Ct+Cs).inverse()) */
static void inverse3(float M[3][3], float I[3][3]) {
    const float c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1];
    const float c10 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
    const float c20 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
    const float det = M[0][0] * c00 + M[0][1] * c10 + M[0][2] * c20;
    const float inv = 1.0f / det;
    I[0][0] = c00 * inv;
    I[1][0] = c10 * inv;
    I[2][0] = c20 * inv;
    I[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * inv;
    I[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * inv;
    I[2][1] = (M[2][0] * M[0][1] - M[0][0] * M[2][1]) * inv;
    I[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * inv;
    I[1][2] = (M[1][0] * M[0][2] - M[0][0] * M[1][2]) * inv;
    I[2][2] = (M[0][0] * M[1][1] - M[1][0] * M[0][1]) * inv;
}

/* GICP weight W = sqrt((Ct+Cs)^-1) (generalized_icp.cu:91-92).
 * covariances are column-major 3x3 (Eigen::Matrix3f). */
static void gicp_weight(const float *Cs, const float *Ct, float W[3][3]) {
    float M[3][3], Mi[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = Ct[c * 3 + r] + Cs[c * 3 + r];
    inverse3(M, Mi);
    sqrt_matrix3x3(Mi, W);
}

ORACLE_API void oracle_gicp_weight(const float *Cs, const float *Ct, float *W_rowmajor) {
    float W[3][3];
    gicp_weight(Cs, Ct, W);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W_rowmajor[r * 3 + c] = W[r][c];
}

/* Entry points for tests/test_outside_checks.py: the pieces above one at a time, so that
 * numpy's / scipy's own routines (LAPACK in fp64) can be held against them. */
ORACLE_API void oracle_fast_eigen3x3(const float *A_rowmajor, int64_t n, float *eval, float *evec_rowmajor) {
    for (int64_t i = 0; i < n; ++i) {
        float A[3][3], ev[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) A[r][c] = A_rowmajor[9 * i + r * 3 + c];
        fast_eigen3x3(A, eval + 3 * i, ev);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) evec_rowmajor[9 * i + r * 3 + c] = ev[r][c]; /* column c = eigenvector c */
    }
}

ORACLE_API float oracle_det6(const float *A_colmajor) { return det6f(A_colmajor); }

ORACLE_API void oracle_ldlt6_solve(const float *A_colmajor, const float *b, float *x) {
    ldlt6_solve(A_colmajor, b, x);
}

/* ------------------------------------------------------------------ */
/* Accumulated linear system.                                          */
/* Layout of sys[32] (fp64):                                           */
/*   [0..20]  upper triangle of JtJ, row-major (00,01,..,05,11,..,55)  */
/*   [21..26] Jtr                                                      */
/*   [27]     sum r^2   (third tuple member, eigen.inl:34-46)          */
/*   [28]     sum d2    (Euclidean NN distance^2 of the pairs used)    */
/*   [29]     count                                                    */
/*   [30..31] reserved (0)                                             */
/* For point-to-point the same 32 slots hold the Kabsch sums instead:  */
/*   [0..2] sum ps, [3..5] sum pt, [6..14] sum ps*pt^T (row-major),    */
/*   [27] sum |ps-pt|^2, [28] sum d2, [29] count.                      */
/* ------------------------------------------------------------------ */

enum { EST_P2P = 1, EST_PT2PL = 2, EST_SYM = 3, EST_COLORED = 4, EST_GICP = 5 };

/* Inputs of TransformationEstimationForColoredICP (colored_icp.cu:150-216): per-point
 * intensities (mean of r,g,b), the target's colour gradient, lambda_geometric.
 * Kept in a context struct so that the other estimators' signatures stay as they are. */
typedef struct {
    const float *src_int;   /* is  per source point */
    const float *tgt_int;   /* it  per target point */
    const float *tgt_grad;  /* dit per target point (3 floats) */
    float lambda_geometric;
} colored_ctx;
static colored_ctx g_colored = {0, 0, 0, 0.968f};

ORACLE_API void oracle_set_colored_context(const float *src_int, const float *tgt_int,
                                           const float *tgt_grad, float lambda_geometric) {
    g_colored.src_int = src_int;
    g_colored.tgt_int = tgt_int;
    g_colored.tgt_grad = tgt_grad;
    g_colored.lambda_geometric = (lambda_geometric < 0 || lambda_geometric > 1.0f) ? 0.968f : lambda_geometric;
}

/* the two rows of colored_icp.cu:183-215 */
static void colored_rows(const float *vs, const float *vt, const float *nt, float is, float it,
                         const float *dit, float slg, float slp, float J0[6], float *r0, float J1[6],
                         float *r1) {
    const float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
    const float dn = dot3(d, nt);
    float c[3];
    cross3(vs, nt, c);
    for (int a = 0; a < 3; ++a) {
        J0[a] = slg * c[a];
        J0[3 + a] = slg * nt[a];
    }
    *r0 = slg * dn;
    const float vs_proj[3] = {vs[0] - dn * nt[0], vs[1] - dn * nt[1], vs[2] - dn * nt[2]};
    const float e[3] = {vs_proj[0] - vt[0], vs_proj[1] - vt[1], vs_proj[2] - vt[2]};
    const float is0_proj = dot3(dit, e) + it;
    /* M = I - nt nt^T ; ditM = -dit^T M */
    float ditM[3];
    for (int col = 0; col < 3; ++col) {
        float s = 0.0f;
        for (int row = 0; row < 3; ++row) {
            const float m = ((row == col) ? 1.0f : 0.0f) - nt[row] * nt[col];
            s += dit[row] * m;
        }
        ditM[col] = -s;
    }
    cross3(vs, ditM, c);
    for (int a = 0; a < 3; ++a) {
        J1[a] = slp * c[a];
        J1[3 + a] = slp * ditM[a];
    }
    *r1 = slp * (is - is0_proj);
}

static inline void accum_row(double *sys, const float *J, float r) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) sys[k] += (double)J[i] * (double)J[j];
    for (int i = 0; i < 6; ++i) sys[21 + i] += (double)J[i] * (double)r;
    sys[27] += (double)r * (double)r;
}

/* ComputeJTJandJTr over the estimator's functor:
 *   point-to-plane  transformation_estimation.cu:34-56  (r=(vs-vt).nt, J=[vs x nt; nt])
 *   symmetric       transformation_estimation.cu:58-90  (n=ns+nt, r=(vs-vt).n, J=[(vs+vt) x n; n])
 *   GICP            generalized_icp.cu:63-105           (3 rows, J=W*[-skew(vs) I], r=W*d)
 *   summed as eigen.inl:34-70,93-145.
 * point-to-point accumulates the Kabsch sums of kabsch.cu:42-104 with the
 * single-pass identity sum (ps-cs)(pt-ct)^T = sum ps pt^T - n_model cs ct^T -
 * ... evaluated at finish time (see oracle_kabsch_from_sums).
 * corres = int32 pairs (source idx, target idx). */
ORACLE_API void oracle_compute_system(int est, const float *src, const float *src_nrm,
                                      const float *src_cov, const float *tgt,
                                      const float *tgt_nrm, const float *tgt_cov,
                                      const int32_t *corres, int64_t c, double *sys_out) {
    /* Summed in fixed chunks of 65536 correspondences (each chunk sequentially, the chunk
     * totals in order), so that the result does not depend on the thread count. */
    enum { CHUNK = 65536 };
    const int64_t nchunk = (c + CHUNK - 1) / CHUNK;
    double *part = (double *)calloc((size_t)(nchunk > 0 ? nchunk : 1) * 32, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
    double *sys = part + ch * 32;
    const int64_t k_end = (ch + 1) * CHUNK < c ? (ch + 1) * CHUNK : c;
    for (int64_t k = ch * CHUNK; k < k_end; ++k) {
        const int i = corres[2 * k], j = corres[2 * k + 1];
        const float *vs = src + 3 * (int64_t)i;
        const float *vt = tgt + 3 * (int64_t)j;
        sys[28] += (double)dist2f(vs, vt);
        sys[29] += 1.0;
        if (est == EST_PT2PL) {
            const float *nt = tgt_nrm + 3 * (int64_t)j;
            float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
            float J[6];
            const float r = dot3(d, nt);
            cross3(vs, nt, J);
            J[3] = nt[0];
            J[4] = nt[1];
            J[5] = nt[2];
            accum_row(sys, J, r);
        } else if (est == EST_SYM) {
            const float *ns = src_nrm + 3 * (int64_t)i;
            const float *nt = tgt_nrm + 3 * (int64_t)j;
            float n[3] = {ns[0] + nt[0], ns[1] + nt[1], ns[2] + nt[2]};
            float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
            float s[3] = {vs[0] + vt[0], vs[1] + vt[1], vs[2] + vt[2]};
            float J[6];
            const float r = dot3(d, n);
            cross3(s, n, J);
            J[3] = n[0];
            J[4] = n[1];
            J[5] = n[2];
            accum_row(sys, J, r);
        } else if (est == EST_COLORED) {
            const float slg = sqrtf(g_colored.lambda_geometric);
            const float slp = sqrtf(1.0f - g_colored.lambda_geometric);
            float J0[6], J1[6], r0, r1;
            colored_rows(vs, vt, tgt_nrm + 3 * (int64_t)j, g_colored.src_int[i], g_colored.tgt_int[j],
                         g_colored.tgt_grad + 3 * (int64_t)j, slg, slp, J0, &r0, J1, &r1);
            accum_row(sys, J0, r0);
            accum_row(sys, J1, r1);
        } else if (est == EST_GICP) {
            float W[3][3];
            gicp_weight(src_cov + 9 * (int64_t)i, tgt_cov + 9 * (int64_t)j, W);
            float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
            /* -skew(vs) = [[0, z, -y], [-z, 0, x], [y, -x, 0]] */
            const float A[3][3] = {{0, vs[2], -vs[1]}, {-vs[2], 0, vs[0]}, {vs[1], -vs[0], 0}};
            for (int row = 0; row < 3; ++row) {
                float J[6];
                for (int col = 0; col < 3; ++col) {
                    J[col] = W[row][0] * A[0][col] + W[row][1] * A[1][col] +
                             W[row][2] * A[2][col];
                    J[3 + col] = W[row][col];
                }
                const float r = W[row][0] * d[0] + W[row][1] * d[1] + W[row][2] * d[2];
                accum_row(sys, J, r);
            }
        } else { /* EST_P2P */
            for (int a = 0; a < 3; ++a) {
                sys[a] += (double)vs[a];
                sys[3 + a] += (double)vt[a];
                for (int b = 0; b < 3; ++b) sys[6 + a * 3 + b] += (double)vs[a] * (double)vt[b];
            }
            sys[27] += (double)dist2f(vs, vt);
        }
    }
    }
    memset(sys_out, 0, 32 * sizeof(double));
    for (int64_t ch = 0; ch < nchunk; ++ch)
        for (int e = 0; e < 32; ++e) sys_out[e] += part[ch * 32 + e];
    free(part);
}

/* Kabsch (kabsch.cu:42-120) from the accumulated sums.  The reference divides
 * the centroid sums and H by model.size() (ALL source points, kabsch.cu:76,107)
 * rather than by the correspondence count -- restated as is. */
ORACLE_API void oracle_kabsch_from_sums(const double *sys, int64_t n_model, float *T) {
    const double c = sys[29];
    const double inv = 1.0 / (double)n_model;
    double cs[3], ct[3], hh[9];
    for (int a = 0; a < 3; ++a) {
        cs[a] = sys[a] * inv;
        ct[a] = sys[3 + a] * inv;
    }
    /* sum (ps-cs)(pt-ct)^T = S_st - cs*St^T - Ss*ct^T + c*cs*ct^T */
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            hh[a * 3 + b] = (sys[6 + a * 3 + b] - cs[a] * sys[3 + b] - sys[a] * ct[b] +
                             c * cs[a] * ct[b]) *
                            inv;
    kabsch_finish(hh, cs, ct, T);
}

/* ComputeRMSE of each estimator (transformation_estimation.cu:109-135,
 * 144-193,224-287 ; generalized_icp.cu:107-150).  GICP returns
 * sqrt(mean d^T W d). */
ORACLE_API float oracle_compute_rmse(int est, const float *src, const float *src_nrm,
                                     const float *src_cov, const float *tgt,
                                     const float *tgt_nrm, const float *tgt_cov,
                                     const int32_t *corres, int64_t c) {
    if (c <= 0) return 0.0f;
    double err = 0.0;
    for (int64_t k = 0; k < c; ++k) {
        const int i = corres[2 * k], j = corres[2 * k + 1];
        const float *vs = src + 3 * (int64_t)i;
        const float *vt = tgt + 3 * (int64_t)j;
        float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
        if (est == EST_COLORED) {
            const float slg = sqrtf(g_colored.lambda_geometric);
            const float slp = sqrtf(1.0f - g_colored.lambda_geometric);
            float J0[6], J1[6], r0, r1;
            colored_rows(vs, vt, tgt_nrm + 3 * (int64_t)j, g_colored.src_int[i], g_colored.tgt_int[j],
                         g_colored.tgt_grad + 3 * (int64_t)j, slg, slp, J0, &r0, J1, &r1);
            err += (double)(r0 * r0 + r1 * r1);
        } else if (est == EST_PT2PL) {
            const float r = dot3(d, tgt_nrm + 3 * (int64_t)j);
            err += (double)(r * r);
        } else if (est == EST_SYM) {
            const float *ns = src_nrm + 3 * (int64_t)i;
            const float *nt = tgt_nrm + 3 * (int64_t)j;
            float n[3] = {ns[0] + nt[0], ns[1] + nt[1], ns[2] + nt[2]};
            const float e = dot3(d, n);
            const float e2 = e * e; /* ComputeErrorUsingNormals returns e^2 ... */
            err += (double)(e2 * e2); /* ... and the lambda squares it again (:283-286) */
        } else if (est == EST_GICP) {
            float W[3][3];
            gicp_weight(src_cov + 9 * (int64_t)i, tgt_cov + 9 * (int64_t)j, W);
            float Wd[3];
            for (int r = 0; r < 3; ++r) Wd[r] = W[r][0] * d[0] + W[r][1] * d[1] + W[r][2] * d[2];
            err += (double)dot3(d, Wd);
        } else {
            err += (double)dot3(d, d);
        }
    }
    if (est == EST_COLORED) return (float)err; /* the reference returns the plain sum (colored_icp.cu:302-306) */
    return sqrtf((float)err / (float)c);
}

/* ------------------------------------------------------------------ */
/* GICP covariance initialisation from normals                         */
/* (generalized_icp.cu:18-30,52-59): C = Rx * diag(eps,1,1) * Rx^T      */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_covariances_from_normals(const float *nrm, int64_t n, float eps,
                                                float *covs) {
    for (int64_t i = 0; i < n; ++i) {
        const float *x = nrm + 3 * i;
        float Rx[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        /* v = e1 x x = (0, -x2, x1) ; c = x0 */
        const float v[3] = {0.0f, -x[2], x[1]};
        const float c = x[0];
        if (!(c < -0.99f)) {
            const float sv[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
            const float factor = 1.0f / (1.0f + c);
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) {
                    const float sv2 = sv[r][0] * sv[0][cc] + sv[r][1] * sv[1][cc] +
                                      sv[r][2] * sv[2][cc];
                    Rx[r][cc] = ((r == cc) ? 1.0f : 0.0f) + sv[r][cc] + sv2 * factor;
                }
        }
        const float D[3] = {eps, 1.0f, 1.0f};
        float *C = covs + 9 * i;
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)
                C[cc * 3 + r] = Rx[r][0] * D[0] * Rx[cc][0] + Rx[r][1] * D[1] * Rx[cc][1] +
                                Rx[r][2] * D[2] * Rx[cc][2];
    }
}

/* ------------------------------------------------------------------ */
/* GetRegistrationResultAndCorrespondences (registration.cu:33-80)     */
/* ------------------------------------------------------------------ */
typedef struct {
    float transformation[16]; /* column-major */
    float fitness;
    float inlier_rmse;
    int64_t n_corres;
    int32_t iterations; /* ICP iterations executed (solves) */
} oracle_result;

static void eval_correspondences(const kd_tree *tree, const float *src_pts, int64_t ns,
                                 float max_dist, int32_t *corres, int *tmp_idx,
                                 float *tmp_d2, oracle_result *res) {
    res->fitness = 0.0f;
    res->inlier_rmse = 0.0f;
    res->n_corres = 0;
    if (max_dist <= 0.0f) return; /* registration.cu:40-42 */
    kd_search_all(tree, src_pts, ns, 1, max_dist * max_dist, tmp_idx, tmp_d2);
    double error2 = 0.0;
    int64_t c = 0;
    for (int64_t i = 0; i < ns; ++i) {
        if (tmp_idx[i] < 0) continue;
        error2 += (double)tmp_d2[i];
        corres[2 * c] = (int32_t)i;
        corres[2 * c + 1] = tmp_idx[i];
        ++c;
    }
    res->n_corres = c;
    if (c > 0) {
        res->fitness = (float)c / (float)ns;
        res->inlier_rmse = sqrtf((float)error2 / (float)c);
    }
}

/* EvaluateRegistration (registration.cu:106-119) */
ORACLE_API int oracle_evaluate_registration(const float *src, int64_t ns, const float *tgt,
                                            int64_t nt, float max_dist, const float *T,
                                            int32_t *corres_out, oracle_result *res) {
    float *pts = (float *)malloc(sizeof(float) * 3 * (size_t)(ns > 0 ? ns : 1));
    memcpy(pts, src, sizeof(float) * 3 * (size_t)ns);
    if (!mat4_is_identity(T)) oracle_transform_points(T, pts, ns);
    kd_tree *tree = kd_build(tgt, (int)nt);
    int *ti = (int *)malloc(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
    float *td = (float *)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1));
    memcpy(res->transformation, T, 16 * sizeof(float));
    res->iterations = 0;
    eval_correspondences(tree, pts, ns, max_dist, corres_out, ti, td, res);
    kd_free(tree);
    free(ti);
    free(td);
    free(pts);
    return 0;
}

/* DESIGN.md deviation 1, restated: the engine applies the COMPOSED transformation to the pristine source on
 * load instead of transforming a copy incrementally (registration.cu:160).  oracle_set_composed(1) makes the
 * loop below do the same -- A <- update * A in fp32 (A = init, or exactly I for an init that isIdentity(),
 * registration.cu:148-150), points / normals / covariances re-derived from the caller's arrays under A -- so
 * that the engine can be held to its own stated form at rounding level on ANY input (tests/test_gpu_fuzz.py),
 * next to the reference's incremental form at the parity tolerance.  Default 0: the reference's form. */
static int g_composed = 0;
ORACLE_API void oracle_set_composed(int on) { g_composed = on != 0; }

/* RegistrationICP (registration.cu:121-172).
 * est: 1 p2p, 2 pt2pl, 3 symmetric, 5 GICP (covariances must be supplied;
 * RegistrationGeneralizedICP's initialisation is oracle_covariances_from_normals).
 * The source copy is transformed incrementally in fp32 every iteration
 * (registration.cu:160), normals by R, covariances by R C R^T. */
ORACLE_API int oracle_registration_icp(
        const float *src, const float *src_nrm, const float *src_cov, int64_t ns,
        const float *tgt, const float *tgt_nrm, const float *tgt_cov, int64_t nt,
        float max_dist, const float *init, int est, float det_thresh,
        float relative_fitness, float relative_rmse, int max_iteration,
        int32_t *corres_out, oracle_result *res) {
    const size_t nsz = (size_t)(ns > 0 ? ns : 1);
    float *pts = (float *)malloc(sizeof(float) * 3 * nsz);
    float *nrm = src_nrm ? (float *)malloc(sizeof(float) * 3 * nsz) : NULL;
    float *cov = src_cov ? (float *)malloc(sizeof(float) * 9 * nsz) : NULL;
    memcpy(pts, src, sizeof(float) * 3 * (size_t)ns);
    if (nrm) memcpy(nrm, src_nrm, sizeof(float) * 3 * (size_t)ns);
    if (cov) memcpy(cov, src_cov, sizeof(float) * 9 * (size_t)ns);
    int *ti = (int *)malloc(sizeof(int) * nsz);
    float *td = (float *)malloc(sizeof(float) * nsz);

    float T[16];
    memcpy(T, init, sizeof(T));
    kd_tree *tree = kd_build(tgt, (int)nt);
    float A[16]; /* composed form: what the pristine source is seen through */
    mat4_identity(A);
    if (!mat4_is_identity(init)) { /* registration.cu:148-150 */
        memcpy(A, init, sizeof(A));
        oracle_transform_points(init, pts, ns);
        if (nrm) oracle_transform_normals(init, nrm, ns);
        if (cov) oracle_rotate_covariances(init, cov, ns);
    }
    oracle_result cur;
    memcpy(cur.transformation, T, sizeof(T));
    cur.iterations = 0;
    eval_correspondences(tree, pts, ns, max_dist, corres_out, ti, td, &cur);

    int it = 0;
    for (; it < max_iteration; ++it) {
        float update[16];
        mat4_identity(update);
        const int64_t c = cur.n_corres;
        /* estimation.ComputeTransformation(pcd, target, corres), :157 */
        if (c > 0) {
            double sys[32];
            if (est == EST_P2P) {
                oracle_compute_system(est, pts, nrm, cov, tgt, tgt_nrm, tgt_cov, corres_out, c, sys);
                oracle_kabsch_from_sums(sys, ns, update);
            } else if (est == EST_PT2PL) {
                if (tgt_nrm) { /* transformation_estimation.cu:199-200 */
                    oracle_compute_system(est, pts, nrm, cov, tgt, tgt_nrm, tgt_cov, corres_out, c, sys);
                    oracle_solve_system(sys, det_thresh, update);
                }
            } else if (est == EST_SYM) {
                if (tgt_nrm && nrm) { /* :293-294 */
                    oracle_compute_system(est, pts, nrm, cov, tgt, tgt_nrm, tgt_cov, corres_out, c, sys);
                    float half[16];
                    if (oracle_solve_system(sys, det_thresh, half)) {
                        /* R = R_half^2 in fp64, translation kept (:312-345) */
                        mat4_identity(update);
                        for (int r = 0; r < 3; ++r) {
                            for (int cc = 0; cc < 3; ++cc) {
                                double s = 0;
                                for (int k = 0; k < 3; ++k)
                                    s += (double)TM(half, r, k) * (double)TM(half, k, cc);
                                TM(update, r, cc) = (float)s;
                            }
                            TM(update, r, 3) = TM(half, r, 3);
                        }
                    }
                }
            } else if (est == EST_COLORED) {
                /* colored_icp.cu:222-224: needs target normals + colours on both clouds */
                if (tgt_nrm && g_colored.src_int && g_colored.tgt_int && g_colored.tgt_grad) {
                    oracle_compute_system(est, pts, nrm, cov, tgt, tgt_nrm, tgt_cov, corres_out, c, sys);
                    oracle_solve_system(sys, det_thresh, update);
                }
            } else if (est == EST_GICP) {
                if (tgt_cov && cov) { /* generalized_icp.cu:156-159 */
                    oracle_compute_system(est, pts, nrm, cov, tgt, tgt_nrm, tgt_cov, corres_out, c, sys);
                    oracle_solve_system(sys, -1.0f, update); /* no det check, :180 */
                }
            }
        }
        mat4_mul(update, T, T); /* :159 */
        if (g_composed) {
            mat4_mul(update, A, A);
            memcpy(pts, src, sizeof(float) * 3 * (size_t)ns);
            oracle_transform_points(A, pts, ns);
            if (nrm) {
                memcpy(nrm, src_nrm, sizeof(float) * 3 * (size_t)ns);
                oracle_transform_normals(A, nrm, ns);
            }
            if (cov) {
                memcpy(cov, src_cov, sizeof(float) * 9 * (size_t)ns);
                oracle_rotate_covariances(A, cov, ns);
            }
        } else {
            oracle_transform_points(update, pts, ns); /* :160 */
            if (nrm) oracle_transform_normals(update, nrm, ns);
            if (cov) oracle_rotate_covariances(update, cov, ns);
        }
        const oracle_result backup = cur; /* :161 */
        memcpy(cur.transformation, T, sizeof(T));
        eval_correspondences(tree, pts, ns, max_dist, corres_out, ti, td, &cur); /* :162 */
        if (fabsf(backup.fitness - cur.fitness) < relative_fitness &&
            fabsf(backup.inlier_rmse - cur.inlier_rmse) < relative_rmse) { /* :165-170 */
            ++it;
            break;
        }
    }
    cur.iterations = it;
    memcpy(cur.transformation, T, sizeof(T));
    *res = cur;
    kd_free(tree);
    free(pts);
    free(nrm);
    free(cov);
    free(ti);
    free(td);
    return 0;
}

/* ------------------------------------------------------------------ */
/* PointCloud::VoxelDownSample (down_sample.cu:64-90,170-273)          */
/* key = floor((p - (min_bound - voxel/2)) / voxel), output ordered by  */
/* lexicographic (x,y,z) key (helper.h:114-121); points / colors       */
/* averaged, normals averaged then normalised.  Returns the number of  */
/* voxels, 0 for voxel_size <= 0 or a too-small voxel (:173-189).       */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t k[3];
    int32_t idx;
} vox_key;

static int vox_cmp(const void *a, const void *b) {
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    for (int d = 0; d < 3; ++d)
        if (x->k[d] != y->k[d]) return (x->k[d] < y->k[d]) ? -1 : 1;
    return (x->idx < y->idx) ? -1 : (x->idx > y->idx);
}

ORACLE_API int64_t oracle_voxel_downsample(const float *pts, const float *nrm,
                                           const float *col, int64_t n, float voxel,
                                           float *out_pts, float *out_nrm,
                                           float *out_col) {
    if (voxel <= 0.0f || n <= 0) return 0;
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (int64_t i = 1; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], pts[3 * i + d]);
            mx[d] = fmaxf(mx[d], pts[3 * i + d]);
        }
    float origin[3], ext = 0.0f;
    for (int d = 0; d < 3; ++d) {
        origin[d] = mn[d] - voxel * 0.5f;
        const float hi = mx[d] + voxel * 0.5f;
        ext = fmaxf(ext, hi - origin[d]);
    }
    if (voxel * (float)INT32_MAX < ext) return 0;
    vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        for (int d = 0; d < 3; ++d)
            keys[i].k[d] = (int32_t)floorf((pts[3 * i + d] - origin[d]) / voxel);
        keys[i].idx = (int32_t)i;
    }
    qsort(keys, (size_t)n, sizeof(vox_key), vox_cmp);
    int64_t m = 0;
    for (int64_t s = 0; s < n;) {
        int64_t e = s + 1;
        while (e < n && keys[e].k[0] == keys[s].k[0] && keys[e].k[1] == keys[s].k[1] &&
               keys[e].k[2] == keys[s].k[2])
            ++e;
        double ap[3] = {0, 0, 0}, an[3] = {0, 0, 0}, ac[3] = {0, 0, 0};
        for (int64_t t = s; t < e; ++t) {
            const int64_t i = keys[t].idx;
            for (int d = 0; d < 3; ++d) {
                ap[d] += pts[3 * i + d];
                if (nrm) an[d] += nrm[3 * i + d];
                if (col) ac[d] += col[3 * i + d];
            }
        }
        const double cnt = (double)(e - s);
        for (int d = 0; d < 3; ++d) out_pts[3 * m + d] = (float)(ap[d] / cnt);
        if (nrm) {
            float v[3] = {(float)(an[0] / cnt), (float)(an[1] / cnt), (float)(an[2] / cnt)};
            const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int d = 0; d < 3; ++d) out_nrm[3 * m + d] = v[d] / l;
        }
        if (col)
            for (int d = 0; d < 3; ++d) out_col[3 * m + d] = (float)(ac[d] / cnt);
        ++m;
        s = e;
    }
    free(keys);
    return m;
}

/* ------------------------------------------------------------------ */
/* PointCloud::CreateFromDepthImage / CreateFromRGBDImage               */
/* (pointcloud_factory.cu:43-110,117-220,286-376), in the reference's   */
/* own order: a structured cloud of one point per pixel (+inf where the */
/* depth is rejected), normals from that cloud's 4-neighbourhood        */
/* (:161-199), then RemoveNoneFinitePoints (pointcloud.cu:40-54).       */
/* depth is float; the uint16 conversion (image.cu:339-348: divide by   */
/* (int)scale, >= (int)trunc -> 0) is oracle_depth_u16_to_float.        */
/* pose = extrinsic^-1, column-major.  rgbd = 0: :43-82 (stride, d <= 0 */
/* rejected); rgbd = 1: :117-160 (cutoff, colours, optional normals).   */
/* The last row's lower neighbour is past the end of the reference's    */
/* buffer (its bounds test is i >= height, :173); taken as zero here,   */
/* the value the functor gives any non-finite neighbour.                */
/* color_kind: 0 none, 1 uint8 x 3, 2 float x 1.  Returns the count.    */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_depth_u16_to_float(const uint16_t *in, int64_t n, float depth_scale,
                                          float depth_trunc, float *out) {
    const int iscale = (int)depth_scale, itrunc = (int)depth_trunc;
    for (int64_t i = 0; i < n; ++i) {
        float f = (float)in[i];
        f /= (float)iscale;
        if (f >= (float)itrunc) f = 0.0f;
        out[i] = f;
    }
}

static int all_finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

ORACLE_API int64_t oracle_create_from_depth(const float *depth, const void *color, int color_kind,
                                            int width, int height, const float *intrinsic4,
                                            const float *pose, float depth_cutoff, int stride,
                                            int rgbd, int compute_normals, int valid_only,
                                            float *out_pts, float *out_nrm, float *out_col) {
    const float fx = intrinsic4[0], fy = intrinsic4[1], cx = intrinsic4[2], cy = intrinsic4[3];
    const int sw = width / stride, sh = height / stride;
    const int64_t count = (int64_t)sw * sh;
    if (count <= 0) return 0;
    float *P = (float *)malloc(sizeof(float) * 3 * (size_t)count);
    float *Cc = color ? (float *)malloc(sizeof(float) * 3 * (size_t)count) : NULL;
    float *Nn = compute_normals ? (float *)malloc(sizeof(float) * 3 * (size_t)count) : NULL;
    for (int64_t idx = 0; idx < count; ++idx) {
        const int row = (int)(idx / sw) * stride, col = (int)(idx % sw) * stride;
        const int64_t pix = (int64_t)row * width + col;
        const float d = depth[pix];
        const int ok = rgbd ? (d > 0.0f && (depth_cutoff <= 0.0f || depth_cutoff > d)) : !(d <= 0.0f);
        float *p = P + 3 * idx;
        if (!ok) {
            p[0] = p[1] = p[2] = INFINITY;
            if (Cc) Cc[3 * idx] = Cc[3 * idx + 1] = Cc[3 * idx + 2] = INFINITY;
            continue;
        }
        const float z = d;
        const float x = ((float)col - cx) * z / fx;
        const float y = ((float)row - cy) * z / fy;
        for (int r = 0; r < 3; ++r)
            p[r] = TM(pose, r, 0) * x + TM(pose, r, 1) * y + TM(pose, r, 2) * z + TM(pose, r, 3);
        if (Cc) {
            if (color_kind == 1) {
                const uint8_t *pc = (const uint8_t *)color + pix * 3;
                for (int k = 0; k < 3; ++k) Cc[3 * idx + k] = (float)pc[k] / 255.0f;
            } else {
                const float v = ((const float *)color)[pix] / 1.0f;
                Cc[3 * idx] = Cc[3 * idx + 1] = Cc[3 * idx + 2] = v;
            }
        }
    }
    if (Nn) {
        for (int64_t idx = 0; idx < count; ++idx) {
            const int i = (int)(idx / width), j = (int)(idx % width);
            float *n = Nn + 3 * idx;
            n[0] = n[1] = n[2] = 0.0f;
            if (i < 1 || i >= height || j < 1 || j >= width) continue;
            const float zero[3] = {0.0f, 0.0f, 0.0f};
            const float *l = P + 3 * (idx - 1);
            const float *r = (idx + 1 < count) ? P + 3 * (idx + 1) : zero;
            const float *u = P + 3 * (idx - width);
            const float *lo = (idx + width < count) ? P + 3 * (idx + width) : zero;
            if (!all_finite3(l)) l = zero;
            if (!all_finite3(r)) r = zero;
            if (!all_finite3(u)) u = zero;
            if (!all_finite3(lo)) lo = zero;
            const float hor[3] = {l[0] - r[0], l[1] - r[1], l[2] - r[2]};
            const float ver[3] = {u[0] - lo[0], u[1] - lo[1], u[2] - lo[2]};
            float c[3];
            cross3(hor, ver, c);
            const float norm = sqrtf(dot3(c, c));
            if (norm == 0.0f) continue;
            for (int k = 0; k < 3; ++k) n[k] = c[k] / norm;
            if (n[2] > 0.0f)
                for (int k = 0; k < 3; ++k) n[k] *= -1.0f;
        }
    }
    int64_t m = 0;
    for (int64_t idx = 0; idx < count; ++idx) {
        if (valid_only && !all_finite3(P + 3 * idx)) continue;
        memcpy(out_pts + 3 * m, P + 3 * idx, 3 * sizeof(float));
        if (Nn) memcpy(out_nrm + 3 * m, Nn + 3 * idx, 3 * sizeof(float));
        if (Cc) memcpy(out_col + 3 * m, Cc + 3 * idx, 3 * sizeof(float));
        ++m;
    }
    free(P);
    free(Cc);
    free(Nn);
    return m;
}

/* ------------------------------------------------------------------ */
/* PointCloud::EstimateNormals(KNN k) (estimate_normals.cu:38-127,     */
/* geometry_functor.h:35-55): neighbours include the point itself,     */
/* covariance from raw second moments in fp32, smallest-eigenvalue      */
/* eigenvector via FastEigen3x3MinMaxVec, (0,0,1) when count < 3 or the */
/* result has zero norm.  The reference sums the 9 cumulants in fp32    */
/* (reduce_by_key); restated in fp32 in neighbour order.               */
/* ------------------------------------------------------------------ */
static void normal_from_cumulants(const float *cum, int count, float *out) {
    out[0] = 0.0f;
    out[1] = 0.0f;
    out[2] = 1.0f;
    if (count < 3) return;
    float c[9];
    for (int i = 0; i < 9; ++i) c[i] = cum[i] / (float)count;
    float A[3][3];
    A[0][0] = c[3] - c[0] * c[0];
    A[1][1] = c[6] - c[1] * c[1];
    A[2][2] = c[8] - c[2] * c[2];
    A[0][1] = A[1][0] = c[4] - c[0] * c[1];
    A[0][2] = A[2][0] = c[5] - c[0] * c[2];
    A[1][2] = A[2][1] = c[7] - c[1] * c[2];
    float eval[3], evec[3][3];
    fast_eigen3x3(A, eval, evec);
    int mi = 0;
    if (eval[1] < eval[mi]) mi = 1;
    if (eval[2] < eval[mi]) mi = 2;
    float nrm[3] = {evec[0][mi], evec[1][mi], evec[2][mi]};
    const float l = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    if (l == 0.0f || isnan(l)) return;
    memcpy(out, nrm, sizeof(nrm));
}

static int estimate_normals_impl(const float *pts, int64_t n, int k, float r2, float *out);

ORACLE_API int oracle_estimate_normals_knn(const float *pts, int64_t n, int k, float *out) {
    return estimate_normals_impl(pts, n, k, INFINITY, out);
}

/* KDTreeSearchParamRadius(radius, max_nn): the max_nn nearest with d2 < radius^2
 * (estimate_normals.cu:93-101 over kdtree_flann.inl:96-122) */
ORACLE_API int oracle_estimate_normals_radius(const float *pts, int64_t n, float radius, int max_nn,
                                              float *out) {
    return estimate_normals_impl(pts, n, max_nn, radius * radius, out);
}

static int estimate_normals_impl(const float *pts, int64_t n, int k, float r2, float *out) {
    if (k <= 0) {
        for (int64_t i = 0; i < n; ++i) {
            out[3 * i] = 0;
            out[3 * i + 1] = 0;
            out[3 * i + 2] = 1;
        }
        return 0;
    }
    kd_tree *t = kd_build(pts, (int)n);
#pragma omp parallel
    {
        int *idx = (int *)malloc(sizeof(int) * (size_t)k);
        float *d2 = (float *)malloc(sizeof(float) * (size_t)k);
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            kd_result r = {k, 0, r2, d2, idx};
            kd_search_rec(t, 0, pts + 3 * i, &r);
            float cum[9] = {0};
            for (int s = 0; s < r.count; ++s) {
                const float *p = pts + 3 * (int64_t)idx[s];
                cum[0] += p[0];
                cum[1] += p[1];
                cum[2] += p[2];
                cum[3] += p[0] * p[0];
                cum[4] += p[0] * p[1];
                cum[5] += p[0] * p[2];
                cum[6] += p[1] * p[1];
                cum[7] += p[1] * p[2];
                cum[8] += p[2] * p[2];
            }
            normal_from_cumulants(cum, r.count, out + 3 * i);
        }
        free(idx);
        free(d2);
    }
    kd_free(t);
    return 0;
}

/* InitializePointCloudForColoredICP (colored_icp.cu:74-148): per target point the
 * least-squares gradient of the intensity over the tangent plane, from the max_nn
 * nearest neighbours within radius (the first -- the point itself -- is skipped),
 * fewer than 4 neighbours -> 0.  intensity[i] = (r+g+b)/3. */
ORACLE_API int oracle_color_gradients(const float *pts, const float *nrm, const float *intensity,
                                      int64_t n, float radius, int max_nn, float *grad_out) {
    kd_tree *t = kd_build(pts, (int)n);
    const float r2 = radius * radius;
#pragma omp parallel
    {
        int *idx = (int *)malloc(sizeof(int) * (size_t)(max_nn > 0 ? max_nn : 1));
        float *d2 = (float *)malloc(sizeof(float) * (size_t)(max_nn > 0 ? max_nn : 1));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            float *g = grad_out + 3 * i;
            g[0] = g[1] = g[2] = 0.0f;
            if (max_nn <= 0) continue;
            kd_result r = {max_nn, 0, r2, d2, idx};
            kd_search_rec(t, 0, pts + 3 * i, &r);
            const float *vt = pts + 3 * i, *nt = nrm + 3 * i;
            const float it = intensity[i];
            float AtA[3][3] = {{0}}, Atb[3] = {0, 0, 0};
            int nn = 0;
            for (int s = 1; s < r.count; ++s) {
                const float *va = pts + 3 * (int64_t)idx[s];
                const float da[3] = {va[0] - vt[0], va[1] - vt[1], va[2] - vt[2]};
                const float h = dot3(da, nt);
                const float v[3] = {va[0] - h * nt[0] - vt[0], va[1] - h * nt[1] - vt[1],
                                    va[2] - h * nt[2] - vt[2]};
                const float di = intensity[idx[s]] - it;
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 3; ++b) AtA[a][b] += v[a] * v[b];
                    Atb[a] += di * v[a];
                }
                ++nn;
            }
            if (nn < 4) continue;
            const float w = (float)((nn - 1) * (nn - 1));
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) AtA[a][b] += w * nt[a] * nt[b];
                AtA[a][a] += 1.0e-6f;
            }
            float inv[3][3];
            inverse3(AtA, inv);
            for (int a = 0; a < 3; ++a) g[a] = inv[a][0] * Atb[0] + inv[a][1] * Atb[1] + inv[a][2] * Atb[2];
        }
        free(idx);
        free(d2);
    }
    kd_free(t);
    return 0;
}

/* Kabsch without correspondences (kabsch.cu:122-...): all points paired by
 * index; pinned by src/tests/registration/kabsch.cpp:35-55. */
ORACLE_API void oracle_kabsch(const float *model, const float *target, int64_t n, float *T) {
    double sys[32];
    memset(sys, 0, sizeof(sys));
    for (int64_t i = 0; i < n; ++i) {
        const float *vs = model + 3 * i, *vt = target + 3 * i;
        for (int a = 0; a < 3; ++a) {
            sys[a] += vs[a];
            sys[3 + a] += vt[a];
            for (int b = 0; b < 3; ++b) sys[6 + a * 3 + b] += (double)vs[a] * (double)vt[b];
        }
    }
    sys[29] = (double)n;
    oracle_kabsch_from_sums(sys, n, T);
}


/* ------------------------------------------------------------------ */
/* cpu_baseline leg of bench.py: wall-clock of ONE point-to-plane ICP   */
/* iteration (radius 1-NN for every sampled source point, 6x6          */
/* accumulation, solve) over the first n_sample source points against  */
/* the full target; tree build timed separately.  OpenMP over queries. */
/* ------------------------------------------------------------------ */
#include <time.h>
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

ORACLE_API int oracle_bench_iteration(const float *src, int64_t ns, const float *tgt,
                                      const float *tgt_nrm, int64_t nt, float max_dist,
                                      int64_t n_sample, int repeats, double *build_s,
                                      double *iter_s, double *fitness, int64_t n_single,
                                      double *iter_single_s) {
    if (n_sample > ns) n_sample = ns;
    double t0 = now_s();
    kd_tree *tree = kd_build(tgt, (int)nt);
    *build_s = now_s() - t0;
    int *ti = (int *)malloc(sizeof(int) * (size_t)n_sample);
    float *td = (float *)malloc(sizeof(float) * (size_t)n_sample);
    int32_t *cor = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n_sample);
    double best = 1e300;
    oracle_result res;
    for (int r = 0; r < repeats; ++r) {
        t0 = now_s();
        eval_correspondences(tree, src, n_sample, max_dist, cor, ti, td, &res);
        double sys[32];
        float T[16];
        oracle_compute_system(EST_PT2PL, src, NULL, NULL, tgt, tgt_nrm, NULL, cor, res.n_corres, sys);
        oracle_solve_system(sys, -1.0f, T);
        const double dt = now_s() - t0;
        if (dt < best) best = dt;
    }
    *iter_s = best;
    *fitness = res.fitness;
    /* the same iteration on ONE thread over the first n_single points (the reference's README
     * quotes its CPU comparison single-threaded, README.md:124) */
    if (iter_single_s) {
        *iter_single_s = 0.0;
        if (n_single > n_sample) n_single = n_sample;
        if (n_single > 0) {
#ifdef _OPENMP
            const int saved = omp_get_max_threads();
            omp_set_num_threads(1);
#endif
            oracle_result r1;
            t0 = now_s();
            eval_correspondences(tree, src, n_single, max_dist, cor, ti, td, &r1);
            double sys[32];
            float T[16];
            oracle_compute_system(EST_PT2PL, src, NULL, NULL, tgt, tgt_nrm, NULL, cor, r1.n_corres, sys);
            oracle_solve_system(sys, -1.0f, T);
            *iter_single_s = now_s() - t0;
#ifdef _OPENMP
            omp_set_num_threads(saved);
#endif
        }
    }
    kd_free(tree);
    free(ti);
    free(td);
    free(cor);
    return 0;
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
