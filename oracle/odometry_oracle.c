/*
 * odometry_oracle.c -- CPU restatement of cupoch's RGB-D odometry
 * (odometry::ComputeRGBDOdometry, src/cupoch/odometry/odometry.cu), the other in-repo caller
 * of utility::ComputeJTJandJTr / SolveJacobianSystemAndObtainExtrinsicMatrix next to the ICP
 * estimators.
 *
 * TEST INFRASTRUCTURE ONLY (see icp_oracle.c): nothing under cupoch_amd/ may call or link it.
 * Every function cites the reference file:line it restates, in the reference's own structure
 * (materialised filtered images and pyramids, a compacted correspondence list, one Jacobian
 * functor call per correspondence) -- the HIP engine fuses most of this.
 *
 * Pinning: the two Jacobian functors are checked against the reference's golden vectors
 * (src/tests/odometry/rgbdodometry_jacobian_from_{color,hybrid}_term.cpp, inputs regenerated
 * through the reference's unit_test::Raw in oracle/_ref).  The image filters, the
 * correspondence rule, the multi-scale loop and the information matrix have NO reference test:
 * PARITY UNPINNED, checked by self-consistency (recover a known camera motion).
 *
 * Arithmetic: fp32 per pixel in the reference's order (compiled with -ffp-contract=off), sums
 * over correspondences in fp64 (the reference tree-sums in fp32).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

int oracle_solve_system(const double *sys, float det_thresh, float *T); /* icp_oracle.c */

/* ---- images ---------------------------------------------------------------------------- */

/* Image::FilterHorizontal (geometry/image.cu:176-205): clamp-to-edge 1-D correlation */
static void filter_horizontal(const float *src, int w, int h, const float *k, int half, float *dst) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float temp = 0;
            for (int i = -half; i <= half; ++i) {
                int xs = x + i;
                xs = xs < 0 ? 0 : (xs > w - 1 ? w - 1 : xs);
                temp += src[y * w + xs] * k[i + half];
            }
            dst[y * w + x] = temp;
        }
}

static void transpose(const float *src, int w, int h, float *dst) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[x * h + y] = src[y * w + x];
}

/* Image::Filter(type) (image.cu:30-75,522-570): horizontal pass with the first kernel,
 * transpose, horizontal pass with the second, transpose back.
 * type 0 Gaussian3 (1/4,1/2,1/4 both ways), 1 Sobel3Dx ((-1,0,1) along x, (1,2,1) along y),
 * 2 Sobel3Dy ((1,2,1) along x, (-1,0,1) along y). */
ORACLE_API void oracle_od_filter(const float *src, int w, int h, int type, float *dst) {
    static const float g3[3] = {0.25f, 0.5f, 0.25f}, s1[3] = {-1.0f, 0.0f, 1.0f}, s2[3] = {1.0f, 2.0f, 1.0f};
    const float *kx = type == 0 ? g3 : (type == 1 ? s1 : s2);
    const float *ky = type == 0 ? g3 : (type == 1 ? s2 : s1);
    float *t1 = (float *)malloc(sizeof(float) * (size_t)w * h), *t2 = (float *)malloc(sizeof(float) * (size_t)w * h);
    filter_horizontal(src, w, h, kx, 1, t1);
    transpose(t1, w, h, t2);
    filter_horizontal(t2, h, w, ky, 1, t1);
    transpose(t1, h, w, dst);
    free(t1);
    free(t2);
}

/* Image::Downsample, float images (image.cu:121-145,454-486): 2x2 mean, floor(w/2) x floor(h/2) */
ORACLE_API void oracle_od_downsample(const float *src, int w, int h, float *dst) {
    const int hw = w / 2, hh = h / 2;
    for (int y = 0; y < hh; ++y)
        for (int x = 0; x < hw; ++x) {
            const float p1 = src[(y * 2) * w + x * 2], p2 = src[(y * 2) * w + x * 2 + 1];
            const float p3 = src[(y * 2 + 1) * w + x * 2], p4 = src[(y * 2 + 1) * w + x * 2 + 1];
            dst[y * hw + x] = (p1 + p2 + p3 + p4) / 4.0f;
        }
}

/* PreprocessDepth (odometry.cu:444-474) */
ORACLE_API void oracle_od_preprocess_depth(float *depth, int64_t n, float min_depth, float max_depth) {
    for (int64_t i = 0; i < n; ++i)
        if (depth[i] < min_depth || depth[i] > max_depth || depth[i] <= 0) depth[i] = NAN;
}

/* ---- correspondences (odometry.cu:153-270) ---------------------------------------------- */

/* Eigen 3x3 inverse by cofactors, fp32, row-major in/out */
static void inv3(const float *M, float *I) {
    const float c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const float det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0f / det;
    I[0] = c00 * id;
    I[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id;
    I[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id;
    I[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

static void mul3(const float *A, const float *B, float *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}

/* K (row-major 3x3) and the column-major 4x4 extrinsic -> KRK^-1 (row-major) and Kt (:225-229) */
ORACLE_API void oracle_od_projection(const float *K, const float *E, float *KRKinv, float *Kt) {
    float Kinv[9], R[9], KR[9];
    inv3(K, Kinv);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = E[c * 4 + r];
    mul3(K, R, KR);
    mul3(KR, Kinv, KRKinv);
    for (int r = 0; r < 3; ++r) Kt[r] = (K[r * 3] * E[12] + K[r * 3 + 1] * E[13]) + K[r * 3 + 2] * E[14];
}

/* The rule of compute_correspondence_map (:182-203) for source pixel (u_s, v_s): returns 1 and
 * the target pixel when the warped pixel lands inside the image on a valid depth within
 * max_depth_diff of the warped depth. */
static int correspond(const float *KRKinv, const float *Kt, const float *depth_s, const float *depth_t, int w,
                      int h, float max_depth_diff, int u_s, int v_s, int *u_t, int *v_t) {
    const float d_s = depth_s[v_s * w + u_s];
    if (isnan(d_s)) return 0;
    float uv[3];
    for (int r = 0; r < 3; ++r) { /* d_s * KRK_inv * (u, v, 1) + Kt: the matrix is scaled first */
        const float m0 = d_s * KRKinv[r * 3], m1 = d_s * KRKinv[r * 3 + 1], m2 = d_s * KRKinv[r * 3 + 2];
        uv[r] = ((m0 * (float)u_s + m1 * (float)v_s) + m2 * 1.0f) + Kt[r];
    }
    const float tz = uv[2];
    const int ut = (int)(uv[0] / tz + 0.5), vt = (int)(uv[1] / tz + 0.5); /* (double + 0.5, truncated) */
    if (!(ut >= 0 && ut < w && vt >= 0 && vt < h)) return 0;
    const float d_t = depth_t[vt * w + ut];
    if (isnan(d_t) || !(fabsf(tz - d_t) <= max_depth_diff)) return 0;
    *u_t = ut;
    *v_t = vt;
    return 1;
}

/* ComputeCorrespondence (:219-270): (u_s, v_s, u_t, v_t) of every source pixel that has one, in
 * pixel order (the map is indexed by the SOURCE pixel, so no two entries compete). */
ORACLE_API int64_t oracle_od_correspondence(const float *K, const float *E, const float *depth_s,
                                            const float *depth_t, int w, int h, float max_depth_diff,
                                            int32_t *corr4) {
    float KRKinv[9], Kt[3];
    oracle_od_projection(K, E, KRKinv, Kt);
    int64_t n = 0;
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {
            int ut, vt;
            if (correspond(KRKinv, Kt, depth_s, depth_t, w, h, max_depth_diff, u, v, &ut, &vt)) {
                corr4[4 * n] = u;
                corr4[4 * n + 1] = v;
                corr4[4 * n + 2] = ut;
                corr4[4 * n + 3] = vt;
                ++n;
            }
        }
    return n;
}

/* ---- Jacobians (odometry/rgbdodometry_jacobian.inl) ------------------------------------- */

#define SOBEL_SCALE 0.125f
#define LAMBDA_HYBRID_DEPTH 0.968f

/* RGBDOdometryJacobianFromColorTerm::ComputeJacobianAndResidual (.inl:41-94) and
 * ...FromHybridTerm (.inl:96-172).  Images are w-wide float; source_xyz has 3 channels;
 * intrinsic row-major 3x3 (only fx, fy are read), extrinsic column-major 4x4.  hybrid = 0:
 * row 1 is zero. */
ORACLE_API void oracle_od_jacobian(int hybrid, int row, const int32_t *corr4, const float *source_color,
                                   const float *target_color, const float *target_depth, const float *source_xyz,
                                   const float *dx_color, const float *dx_depth, const float *dy_color,
                                   const float *dy_depth, int w, const float *K, const float *E, float *J0,
                                   float *r0, float *J1, float *r1) {
    const int u_s = corr4[4 * row], v_s = corr4[4 * row + 1], u_t = corr4[4 * row + 2], v_t = corr4[4 * row + 3];
    const float fx = K[0], fy = K[4];
    const float diff_photo = target_color[v_t * w + u_t] - source_color[v_s * w + u_s];
    const float dIdx = SOBEL_SCALE * dx_color[v_t * w + u_t], dIdy = SOBEL_SCALE * dy_color[v_t * w + u_t];
    const float *p = source_xyz + 3 * (v_s * w + u_s);
    float pt[3];
    for (int r = 0; r < 3; ++r) pt[r] = ((E[r] * p[0] + E[4 + r] * p[1]) + E[8 + r] * p[2]) + E[12 + r];
    const float invz = (float)(1. / pt[2]);
    const float c0 = dIdx * fx * invz, c1 = dIdy * fy * invz;
    const float c2 = -(c0 * pt[0] + c1 * pt[1]) * invz;
    if (!hybrid) {
        J0[0] = -pt[2] * c1 + pt[1] * c2;
        J0[1] = pt[2] * c0 - pt[0] * c2;
        J0[2] = -pt[1] * c0 + pt[0] * c1;
        J0[3] = c0;
        J0[4] = c1;
        J0[5] = c2;
        *r0 = diff_photo;
        for (int k = 0; k < 6; ++k) J1[k] = 0.0f;
        *r1 = 0.0f;
        return;
    }
    const float sl_dep = (float)sqrt(LAMBDA_HYBRID_DEPTH), sl_img = (float)sqrt(1.0 - LAMBDA_HYBRID_DEPTH);
    float dDdx = SOBEL_SCALE * dx_depth[v_t * w + u_t], dDdy = SOBEL_SCALE * dy_depth[v_t * w + u_t];
    if (isnan(dDdx)) dDdx = 0;
    if (isnan(dDdy)) dDdy = 0;
    const float diff_geo = target_depth[v_t * w + u_t] - pt[2];
    const float d0 = dDdx * fx * invz, d1 = dDdy * fy * invz;
    const float d2 = -(d0 * pt[0] + d1 * pt[1]) * invz;
    J0[0] = sl_img * (-pt[2] * c1 + pt[1] * c2);
    J0[1] = sl_img * (pt[2] * c0 - pt[0] * c2);
    J0[2] = sl_img * (-pt[1] * c0 + pt[0] * c1);
    J0[3] = sl_img * c0;
    J0[4] = sl_img * c1;
    J0[5] = sl_img * c2;
    *r0 = sl_img * diff_photo;
    J1[0] = sl_dep * ((-pt[2] * d1 + pt[1] * d2) - pt[1]);
    J1[1] = sl_dep * ((pt[2] * d0 - pt[0] * d2) + pt[0]);
    J1[2] = sl_dep * (-pt[1] * d0 + pt[0] * d1);
    J1[3] = sl_dep * d0;
    J1[4] = sl_dep * d1;
    J1[5] = sl_dep * (d2 - 1.0f);
    *r1 = sl_dep * diff_geo;
}

/* ConvertDepthImageToXYZImage (odometry.cu:273-330) */
static void depth_to_xyz(const float *depth, int w, int h, const float *K, float *xyz) {
    const float inv_fx = (float)(1.0 / K[0]), inv_fy = (float)(1.0 / K[4]), ox = K[2], oy = K[5];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float z = depth[y * w + x];
            xyz[3 * (y * w + x)] = ((float)x - ox) * z * inv_fx;
            xyz[3 * (y * w + x) + 1] = ((float)y - oy) * z * inv_fy;
            xyz[3 * (y * w + x) + 2] = z;
        }
}

static void accum(double *sys, const float *J, float r) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) sys[k] += (double)J[i] * (double)J[j];
    for (int i = 0; i < 6; ++i) sys[21 + i] += (double)J[i] * (double)r;
    sys[27] += (double)r * (double)r;
}

static void mul4(const float *A, const float *B, float *C) { /* column-major 4x4 */
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
            t[c * 4 + r] = s;
        }
    memcpy(C, t, sizeof(t));
}

/* ---- the pipeline ----------------------------------------------------------------------- */

/* odometry::ComputeRGBDOdometry (odometry.cu:498-528 InitializeRGBDOdometry, :584-631
 * DoSingleIteration, :708-764 ComputeMultiscale, :371-394 CreateInformationMatrix,
 * :833-879 ComputeRGBDOdometryT).  color / depth: float images w x h; intrinsic4 = fx, fy, cx,
 * cy; odo_init, trans_out: column-major 4x4; iterations[l]: coarsest level first
 * (iteration_number_per_pyramid_level_); info_out: row-major 6x6.  Returns is_success
 * (failure: identity transformation and identity information, :876-878). */
/* utility::TransformMatrix4fToVector6f (utility/eigen.cu:52-65): Eigen::Quaternionf from the
 * rotation block (Eigen's trace / largest-diagonal branches), angle * axis, translation. */
ORACLE_API void oracle_matrix4_to_vector6(const float *T, float *out) {
    float m[3][3], q[4]; /* q = x, y, z, w */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m[r][c] = T[c * 4 + r];
    float tr = m[0][0] + m[1][1] + m[2][2];
    if (tr > 0.0f) {
        float s = sqrtf(tr + 1.0f);
        q[3] = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m[2][1] - m[1][2]) * s;
        q[1] = (m[0][2] - m[2][0]) * s;
        q[2] = (m[1][0] - m[0][1]) * s;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = sqrtf(m[i][i] - m[j][j] - m[k][k] + 1.0f);
        q[i] = 0.5f * s;
        s = 0.5f / s;
        q[3] = (m[k][j] - m[j][k]) * s;
        q[j] = (m[j][i] + m[i][j]) * s;
        q[k] = (m[k][i] + m[i][k]) * s;
    }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float angle = 0, axis[3] = {0, 0, 1.0f};
    if (n > 0) {
        angle = (float)(2.0 * atan2f(n, q[3]));
        for (int a = 0; a < 3; ++a) axis[a] = q[a] / n;
    }
    for (int a = 0; a < 3; ++a) {
        out[a] = angle * axis[a];
        out[3 + a] = T[12 + a];
    }
}

/* odometry::ComputeRGBDOdometry and ComputeWeightedRGBDOdometry (odometry.cu:498-528
 * InitializeRGBDOdometry, :584-631 DoSingleIteration, :633-706 DoSingleIterationWeighted over
 * utility/eigen.inl:147-195 ComputeWeightedJTJandJTr, :708-831 ComputeMultiscale[Weighted],
 * :371-394 CreateInformationMatrix, :833-879 ComputeRGBDOdometryT).  color / depth: float images
 * w x h; intrinsic4 = fx, fy, cx, cy; odo_init, trans_out: column-major 4x4; iterations[l]:
 * coarsest level first (iteration_number_per_pyramid_level_); info_out: row-major 6x6.
 * weighted != 0 (always the hybrid term, :937-941): the t-distribution weights -- w_sum = sum over
 * correspondences of r2 (nu+1) / (nu + r2 / sigma2) with r2 the correspondence's squared residual,
 * weight = (nu+1) / (nu + r2 / w_sum), sigma2 <- w_sum -- plus the motion prior
 * inv_sigma_diag . (prev_twist - current velocity); twist_out = the velocity at the end.
 * Returns is_success (failure: identity transformation and identity information, :876-878). */
static int od_core(const float *src_color, const float *src_depth, const float *tgt_color, const float *tgt_depth,
                   int w, int h, const float *intrinsic4, const float *odo_init, int hybrid, const int *iterations,
                   int num_levels, float max_depth_diff, float min_depth, float max_depth, int weighted, float nu,
                   float sigma2_init, const float *inv_sigma_diag, const float *prev_twist, float *trans_out,
                   float *twist_out, double *info_out) {
    const size_t n0 = (size_t)w * h;
    float *col[2][8], *dep[2][8];
    int lw[8], lh[8];
    /* InitializeRGBDOdometry */
    for (int s = 0; s < 2; ++s) {
        col[s][0] = (float *)malloc(sizeof(float) * n0);
        dep[s][0] = (float *)malloc(sizeof(float) * n0);
        oracle_od_filter(s ? tgt_color : src_color, w, h, 0, col[s][0]);
        float *d = (float *)malloc(sizeof(float) * n0);
        memcpy(d, s ? tgt_depth : src_depth, sizeof(float) * n0);
        oracle_od_preprocess_depth(d, (int64_t)n0, min_depth, max_depth);
        oracle_od_filter(d, w, h, 0, dep[s][0]);
        free(d);
    }
    float K0[9] = {intrinsic4[0], 0, intrinsic4[2], 0, intrinsic4[1], intrinsic4[3], 0, 0, 1};
    int32_t *corr = (int32_t *)malloc(sizeof(int32_t) * 4 * n0);
    {   /* NormalizeIntensity (:416-436) */
        const int64_t nc = oracle_od_correspondence(K0, odo_init, dep[0][0], dep[1][0], w, h, max_depth_diff, corr);
        double ms = 0.0, mt = 0.0;
        for (int64_t i = 0; i < nc; ++i) {
            ms += col[0][0][corr[4 * i + 1] * w + corr[4 * i]];
            mt += col[1][0][corr[4 * i + 3] * w + corr[4 * i + 2]];
        }
        const float mean_s = (float)ms / (float)nc, mean_t = (float)mt / (float)nc;
        const float sc_s = (float)(0.5 / mean_s), sc_t = (float)(0.5 / mean_t);
        for (size_t i = 0; i < n0; ++i) { /* Image::LinearTransform: scale * f + offset */
            col[0][0][i] = sc_s * col[0][0][i] + 0.0f;
            col[1][0][i] = sc_t * col[1][0][i] + 0.0f;
        }
    }
    /* pyramids: colour Gaussian3 + Downsample, depth Downsample only (rgbdimage.cu:96-112,
     * image_factory.cu:251-278) */
    lw[0] = w;
    lh[0] = h;
    for (int l = 1; l < num_levels; ++l) {
        lw[l] = lw[l - 1] / 2;
        lh[l] = lh[l - 1] / 2;
        for (int s = 0; s < 2; ++s) {
            float *b = (float *)malloc(sizeof(float) * (size_t)lw[l - 1] * lh[l - 1]);
            oracle_od_filter(col[s][l - 1], lw[l - 1], lh[l - 1], 0, b);
            col[s][l] = (float *)malloc(sizeof(float) * (size_t)lw[l] * lh[l]);
            oracle_od_downsample(b, lw[l - 1], lh[l - 1], col[s][l]);
            free(b);
            dep[s][l] = (float *)malloc(sizeof(float) * (size_t)lw[l] * lh[l]);
            oracle_od_downsample(dep[s][l - 1], lw[l - 1], lh[l - 1], dep[s][l]);
        }
    }
    float T[16];
    {   /* extrinsic_initial.isZero() ? Identity : extrinsic_initial (:722-724) */
        int zero = 1;
        for (int i = 0; i < 16; ++i) zero &= (odo_init[i] == 0.0f);
        memcpy(T, odo_init, sizeof(T));
        if (zero) {
            memset(T, 0, sizeof(T));
            T[0] = T[5] = T[10] = T[15] = 1.0f;
        }
    }
    int ok = 1;
    float sigma2 = sigma2_init;
    float vel[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; /* curr_vel (:785) */
    float Kl[8][9];
    for (int l = 0; l < num_levels; ++l) { /* CreateCameraMatrixPyramid (:332-347) */
        for (int i = 0; i < 9; ++i) Kl[l][i] = l == 0 ? K0[i] : (float)(0.5 * Kl[l - 1][i]);
        Kl[l][8] = 1.0f;
    }
    for (int level = num_levels - 1; level >= 0 && ok; --level) {
        const int W = lw[level], H = lh[level];
        const size_t n = (size_t)W * H;
        float *xyz = (float *)malloc(sizeof(float) * 3 * n);
        float *dxc = (float *)malloc(sizeof(float) * n), *dyc = (float *)malloc(sizeof(float) * n);
        float *dxd = (float *)malloc(sizeof(float) * n), *dyd = (float *)malloc(sizeof(float) * n);
        depth_to_xyz(dep[0][level], W, H, Kl[level], xyz);
        oracle_od_filter(col[1][level], W, H, 1, dxc);
        oracle_od_filter(dep[1][level], W, H, 1, dxd);
        oracle_od_filter(col[1][level], W, H, 2, dyc);
        oracle_od_filter(dep[1][level], W, H, 2, dyd);
        for (int iter = 0; iter < iterations[num_levels - level - 1] && ok; ++iter) {
            const int64_t nc =
                    oracle_od_correspondence(Kl[level], T, dep[0][level], dep[1][level], W, H, max_depth_diff, corr);
            double sys[32];
            memset(sys, 0, sizeof(sys));
            float w_sum = 0.0f;
            if (weighted) { /* first pass: sum of fw_reduce over the correspondences' r2 */
                double acc = 0.0;
                for (int64_t i = 0; i < nc; ++i) {
                    float J0[6], J1[6], r0, r1;
                    oracle_od_jacobian(hybrid, (int)i, corr, col[0][level], col[1][level], dep[1][level], xyz, dxc,
                                       dxd, dyc, dyd, W, Kl[level], T, J0, &r0, J1, &r1);
                    const float r2 = r0 * r0 + r1 * r1;
                    acc += (double)(float)(r2 * (nu + 1.0) / (nu + r2 / sigma2));
                }
                w_sum = (float)acc;
            }
            for (int64_t i = 0; i < nc; ++i) {
                float J0[6], J1[6], r0, r1;
                oracle_od_jacobian(hybrid, (int)i, corr, col[0][level], col[1][level], dep[1][level], xyz, dxc, dxd,
                                   dyc, dyd, W, Kl[level], T, J0, &r0, J1, &r1);
                if (!weighted) {
                    accum(sys, J0, r0);
                    accum(sys, J1, r1);
                } else {
                    const float r2 = r0 * r0 + r1 * r1;
                    const double wt = (double)((nu + 1) / (nu + r2 / w_sum));
                    double one[32];
                    memset(one, 0, sizeof(one));
                    accum(one, J0, r0);
                    accum(one, J1, r1);
                    for (int k = 0; k < 28; ++k) sys[k] += wt * one[k];
                }
            }
            sys[29] = (double)nc;
            if (weighted) {
                float cv[6];
                oracle_matrix4_to_vector6(vel, cv);
                const int diag[6] = {0, 6, 11, 15, 18, 20}; /* positions of JTJ(i,i) in the packed upper triangle */
                for (int a = 0; a < 6; ++a) {
                    sys[diag[a]] = (double)((float)sys[diag[a]] + inv_sigma_diag[a]);
                    sys[21 + a] = (double)((float)sys[21 + a] - inv_sigma_diag[a] * (prev_twist[a] - cv[a]));
                }
                sigma2 = w_sum;
            }
            float upd[16];
            ok = oracle_solve_system(sys, -1.0f, upd); /* det_thresh default -1: always "solved" (utility/eigen.cu:76-122) */
            if (ok && weighted) mul4(upd, vel, vel);
            if (ok) mul4(upd, T, T);
        }
        free(xyz);
        free(dxc);
        free(dyc);
        free(dxd);
        free(dyd);
    }
    memset(trans_out, 0, sizeof(float) * 16);
    trans_out[0] = trans_out[5] = trans_out[10] = trans_out[15] = 1.0f;
    for (int i = 0; i < 36; ++i) info_out[i] = (i % 7 == 0) ? 1.0 : 0.0;
    if (twist_out) {
        for (int a = 0; a < 6; ++a) twist_out[a] = 0.0f;
        if (ok && weighted) oracle_matrix4_to_vector6(vel, twist_out);
    }
    if (ok) {
        memcpy(trans_out, T, sizeof(T));
        /* CreateInformationMatrix (:349-394): I + sum over correspondences of G^T G */
        const int64_t nc = oracle_od_correspondence(K0, T, dep[0][0], dep[1][0], w, h, max_depth_diff, corr);
        float *xyz_t = (float *)malloc(sizeof(float) * 3 * n0);
        depth_to_xyz(dep[1][0], w, h, K0, xyz_t);
        for (int64_t i = 0; i < nc; ++i) {
            const float *q = xyz_t + 3 * ((size_t)corr[4 * i + 3] * w + corr[4 * i + 2]);
            const float x = q[0], y = q[1], z = q[2];
            const float g[3][6] = {{0.0f, z, -y, 1.0f, 0.0f, 0.0f}, {-z, 0.0f, x, 0.0f, 1.0f, 0.0f}, {y, -x, 0.0f, 0.0f, 0.0f, 1.0f}};
            for (int a = 0; a < 3; ++a)
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) info_out[r * 6 + c] += (double)g[a][r] * (double)g[a][c];
        }
        free(xyz_t);
    }
    for (int l = 0; l < num_levels; ++l)
        for (int s = 0; s < 2; ++s) {
            free(col[s][l]);
            free(dep[s][l]);
        }
    free(corr);
    return ok;
}

ORACLE_API int oracle_od_compute(const float *src_color, const float *src_depth, const float *tgt_color,
                                 const float *tgt_depth, int w, int h, const float *intrinsic4,
                                 const float *odo_init, int hybrid, const int *iterations, int num_levels,
                                 float max_depth_diff, float min_depth, float max_depth, float *trans_out,
                                 double *info_out) {
    return od_core(src_color, src_depth, tgt_color, tgt_depth, w, h, intrinsic4, odo_init, hybrid, iterations, num_levels,
                   max_depth_diff, min_depth, max_depth, 0, 0.0f, 0.0f, NULL, NULL, trans_out, NULL, info_out);
}

ORACLE_API int oracle_od_compute_weighted(const float *src_color, const float *src_depth, const float *tgt_color,
                                          const float *tgt_depth, int w, int h, const float *intrinsic4,
                                          const float *odo_init, const float *prev_twist, const int *iterations,
                                          int num_levels, float max_depth_diff, float min_depth, float max_depth,
                                          float nu, float sigma2_init, const float *inv_sigma_diag, float *trans_out,
                                          float *twist_out, double *info_out) {
    return od_core(src_color, src_depth, tgt_color, tgt_depth, w, h, intrinsic4, odo_init, 1, iterations, num_levels,
                   max_depth_diff, min_depth, max_depth, 1, nu, sigma2_init, inv_sigma_diag, prev_twist, trans_out,
                   twist_out, info_out);
}
