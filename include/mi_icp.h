/*
 * mi_icp.h -- C ABI of libmi_icp.so, the MI355X (gfx950) ICP registration
 * engine that sits behind cupoch's registration / geometry C++ surface.
 *
 * Every entry point names the reference interface it replaces
 * (paths relative to the cupoch tree, v0.2.11.0).  The reference-side
 * bindings (C++ classes in namespace cupoch, Python ctypes) are shown in
 * INTEGRATION.md; cupoch_amd/cpp and the Python modules under cupoch_amd implement them.
 *
 * Conventions
 *   - every function returns an int status: 0 = MI_ICP_OK, negative = error;
 *     nothing throws or exit()s across this boundary (the reference prints and
 *     exit(0)s on device errors, utility/platform.cu:60-67);
 *     mi_icp_last_error() returns the text of the last failure on a context.
 *   - points / normals / colors are AoS float[n][3] with a 12-byte stride
 *     (the memory layout of device_vector<Eigen::Vector3f>,
 *     geometry/pointcloud.h:259-262); covariances are float[n][9],
 *     column-major 3x3 (Eigen::Matrix3f); 4x4 transforms are float[16]
 *     column-major, i.e. exactly Eigen::Matrix4f::data();
 *     correspondences are int32 pairs (source_idx, target_idx) =
 *     device_vector<Eigen::Vector2i> (registration/transformation_estimation.h:36).
 *   - every buffer argument carries a mem_kind: MI_ICP_HOST (pageable or
 *     pinned host memory, copied by the engine) or MI_ICP_DEVICE (a HIP device
 *     pointer on the context's GPU, read in place).  The caller owns all
 *     buffers it passes; the context owns its internal SoA copies, LBVH and
 *     scratch arena.
 *   - one context per GPU; a context is not thread-safe, independent contexts
 *     may be driven from different host threads.
 *   - all work is enqueued on the context's stream (mi_icp_set_stream; the
 *     default is the null stream).  Functions that return scalars or host
 *     buffers synchronise that stream before returning.
 */
#ifndef MI_ICP_H_
#define MI_ICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ICP_API __attribute__((visibility("default")))

enum {
    MI_ICP_OK = 0,
    MI_ICP_ERR_INVALID = -1,   /* bad argument (null, negative size, bad enum) */
    MI_ICP_ERR_STATE = -2,     /* call order: no target / source / normals set */
    MI_ICP_ERR_HIP = -3,       /* a HIP runtime call failed */
    MI_ICP_ERR_COMM = -4,      /* RCCL unavailable or failed */
    MI_ICP_ERR_NO_DEVICE = -5  /* no usable gfx950 device */
};

enum { MI_ICP_HOST = 0, MI_ICP_DEVICE = 1 };

/* registration::TransformationEstimationType
 * (registration/transformation_estimation.h:38-45), same values */
enum {
    MI_ICP_EST_POINT_TO_POINT = 1,
    MI_ICP_EST_POINT_TO_PLANE = 2,
    MI_ICP_EST_SYMMETRIC = 3,
    MI_ICP_EST_COLORED = 4, /* TransformationEstimationType::ColoredICP */
    MI_ICP_EST_GENERALIZED = 5
};

typedef struct mi_icp_ctx mi_icp_ctx;

/* registration::ICPConvergenceCriteria (registration/registration.h:35-49)
 * + the estimator's scalar parameter. */
typedef struct {
    float relative_fitness; /* default 1e-6; compared as an ABSOLUTE difference */
    float relative_rmse;    /* default 1e-6; ditto (registration.cu:165-170) */
    int32_t max_iteration;  /* default 30 */
    float det_thresh;       /* PointToPlane / Symmetric det check, default 1e-6;
                               <= 0 disables it (utility/eigen.cu:114) */
} mi_icp_params;

/* registration::RegistrationResult (registration/registration.h:51-67);
 * the correspondence set itself is fetched with mi_icp_get_correspondences. */
typedef struct {
    float transformation[16]; /* column-major */
    float fitness;
    float inlier_rmse;
    int64_t n_correspondences;
    int32_t iterations;   /* solves executed */
    int32_t nn_passes;    /* nearest-neighbour passes executed (iterations+1) */
} mi_icp_result;

/* ---- context / errors  (replaces utility::InitializeAllocator + GetStream,
 *      utility/device_vector.h:78-106, utility/platform.cu:38-67) ---------- */
MI_ICP_API int mi_icp_create(int device, mi_icp_ctx** out);
MI_ICP_API void mi_icp_destroy(mi_icp_ctx* ctx);
MI_ICP_API const char* mi_icp_last_error(const mi_icp_ctx* ctx);
MI_ICP_API const char* mi_icp_version(void);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
MI_ICP_API int mi_icp_set_stream(mi_icp_ctx* ctx, void* hip_stream);
MI_ICP_API int mi_icp_synchronize(mi_icp_ctx* ctx);

/* ---- clouds -----------------------------------------------------------
 * mi_icp_set_target replaces knn::KDTreeFlann::KDTreeFlann(target.points_) /
 * SetRawData (knn/kdtree_flann.inl:124-144) and FLANN's
 * CudaKdTreeBuilder::buildTree (third_party/flann/algorithms/
 * kdtree_cuda_builder.h:401-700): partitions the target into kd cells and
 * builds the implicit 8-ary tree over them.  normals / covs may be NULL.
 * Synchronous (the tree's size is read back once); at most ~3e8 points.
 * The leaves' HALOS (what a loop's seeded searches use once the matches are no
 * longer exact; csrc/leaf_halo.h) are built on demand, on a private low-priority
 * stream: when a registration loop's searches ask for them (ahead of the loop's first
 * pass, on the loop's own stream, on a context whose loops have asked before), and --
 * for a target below 2M points on a context that has registered before -- right behind
 * the tree, next to whatever the caller enqueues next.
 * Device memory a context keeps per target point: ~75 B of tree (leaf lines, regions,
 * records; 1.67 slots per point at 10M), + 40 B with normals, + 60 B with covariances;
 * the halos add 1 KB per LEAF (~215 B per point) once built, and their build 0.5 KB per
 * leaf of candidate scratch that is released at the end of the registration call that
 * built them (targets whose scratch is below 64 MB keep it: frame-to-frame callers).
 * mi_icp_set_source replaces `geometry::PointCloud pcd = source`
 * (registration/registration.cu:147): the engine keeps a Morton-sorted SoA
 * copy and never mutates the caller's cloud.  Stream-ordered. */
MI_ICP_API int mi_icp_set_target(mi_icp_ctx* ctx, const float* xyz, const float* normals,
                                 const float* covs, int64_t n, int mem_kind);
MI_ICP_API int mi_icp_set_source(mi_icp_ctx* ctx, const float* xyz, const float* normals,
                                 const float* covs, int64_t n, int mem_kind);

/* ---- nearest neighbours -------------------------------------------------
 * knn::KDTreeFlann::SearchRadius(source.points_, r, max_nn = 1, indices,
 * dists) as used by GetRegistrationResultAndCorrespondences
 * (registration/registration.cu:33-80, knn/kdtree_flann.inl:96-122):
 * for every source point transformed by T, the target point with the
 * smallest d2 subject to the strict test d2 < float(r*r); no match ->
 * idx -1, d2 +inf.  idx_out / d2_out are in ORIGINAL source order and hold
 * ORIGINAL target indices; either may be NULL.  stats[3] (optional) receives
 * {count, sum d2, n_source}.  T == NULL means identity.
 * The result also becomes the context's current correspondence set.  Among
 * target points at exactly the same distance the one with the lowest position
 * in the tree's order is returned (FLANN returns the first one it visits).  A
 * search on the same pair of clouds as the previous one starts from its matches;
 * that only makes it faster. */
MI_ICP_API int mi_icp_search_radius_1nn(mi_icp_ctx* ctx, const float* T, float radius,
                                        int32_t* idx_out, float* d2_out, int mem_kind,
                                        double* stats);

/* ---- correspondences ----------------------------------------------------
 * get: RegistrationResult::correspondence_set_ (registration.cu:54-69):
 * pairs (i, j), ascending in source index i (stable compaction).
 * `capacity` is in pairs; *count receives the number of pairs available.
 * set: supplies an explicit CorrespondenceSet for the
 * TransformationEstimation::ComputeTransformation / ComputeRMSE entry points
 * below (registration/transformation_estimation.h:50-65). */
MI_ICP_API int mi_icp_get_correspondences(mi_icp_ctx* ctx, int32_t* pairs, int64_t capacity,
                                          int64_t* count, int mem_kind);
MI_ICP_API int mi_icp_set_correspondences(mi_icp_ctx* ctx, const int32_t* pairs,
                                          int64_t count, int mem_kind);

/* ---- estimation ---------------------------------------------------------
 * mi_icp_compute_system replaces utility::ComputeJTJandJTr
 * (utility/eigen.inl:84-145) over the estimator's Jacobian functor
 * (registration/transformation_estimation.cu:34-90,
 *  registration/generalized_icp.cu:63-105) or, for point-to-point, the three
 * reductions of registration/kabsch.cu:42-104.  The source is taken under T
 * (points R*p+t, normals R*n, covariances R*C*R^T).  out[32] (fp64):
 *   [0..20] upper triangle of JtJ row-major, [21..26] Jtr, [27] sum r^2,
 *   [28] sum d2, [29] count;
 *   point-to-point: [0..2] sum ps, [3..5] sum pt, [6..14] sum ps*pt^T
 *   (row-major), [27] sum |ps-pt|^2, [28] sum d2, [29] count.
 * mi_icp_compute_transformation = TransformationEstimation*::
 * ComputeTransformation (transformation_estimation.cu:137-142,195-222,
 * 289-350; generalized_icp.cu:152-183) incl. SolveJacobianSystemAndObtain-
 * ExtrinsicMatrix (utility/eigen.cu:107-122) / Kabsch (kabsch.cu:105-118);
 * solver failure -> identity.  mi_icp_compute_rmse = ::ComputeRMSE. */
MI_ICP_API int mi_icp_compute_system(mi_icp_ctx* ctx, int est_type, const float* T,
                                     double* out32);
MI_ICP_API int mi_icp_compute_transformation(mi_icp_ctx* ctx, int est_type, const float* T,
                                             float det_thresh, float* update16);
MI_ICP_API int mi_icp_compute_rmse(mi_icp_ctx* ctx, int est_type, const float* T,
                                   float* rmse);
/* host-only helpers, exposed for parity tests:
 * utility::SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.cu:107-122),
 * utility::TransformVector6fToMatrix4f (utility/eigen.cu:28-50). */
MI_ICP_API int mi_icp_solve_system(const double* sys32, float det_thresh, float* T16);
MI_ICP_API int mi_icp_kabsch_from_sums(const double* sys32, int64_t n_model, float* T16);
MI_ICP_API void mi_icp_vector6_to_matrix4(const float* x6, float* T16);

/* host-only: the LZF byte format of PCD's "DATA binary_compressed" (io/file_format/file_pcd.cu:218,461,690
 * call liblzf's lzf_decompress / lzf_compress).  Return the number of bytes produced, 0 on a
 * corrupt stream or when out_capacity is too small (compress: 2 x in_len always suffices). */
MI_ICP_API int64_t mi_icp_lzf_decompress(const void* in, int64_t in_len, void* out, int64_t out_capacity);
MI_ICP_API int64_t mi_icp_lzf_compress(const void* in, int64_t in_len, void* out, int64_t out_capacity);

/* ---- the registration loop ---------------------------------------------
 * registration::EvaluateRegistration (registration.cu:106-119) and
 * registration::RegistrationICP (registration.cu:121-172) with the built-in
 * estimators.  init == NULL means identity.  GICP expects covariances on both
 * clouds (RegistrationGeneralizedICP's InitializePointCloudForGeneralizedICP,
 * generalized_icp.cu:37-61, is mi_icp_covariances_from_normals). */
MI_ICP_API int mi_icp_evaluate_registration(mi_icp_ctx* ctx, float max_distance,
                                            const float* T, mi_icp_result* out);
MI_ICP_API int mi_icp_registration_icp(mi_icp_ctx* ctx, int est_type, float max_distance,
                                       const float* init, const mi_icp_params* params,
                                       mi_icp_result* out);

/* Stepping form of the same loop (no reference counterpart: the reference only
 * offers the whole call).  begin = the setup + first correspondence pass
 * (registration.cu:144-152); iterate(n) = n executions of the loop body
 * (registration.cu:155-163) without the convergence test.  Used by per-frame
 * callers that budget iterations themselves and by bench.py, which times
 * exactly K iterations. */
MI_ICP_API int mi_icp_icp_begin(mi_icp_ctx* ctx, int est_type, float max_distance,
                                const float* init, float det_thresh, mi_icp_result* out);
MI_ICP_API int mi_icp_icp_iterate(mi_icp_ctx* ctx, int n_iterations, mi_icp_result* out);

/* ---- geometry -----------------------------------------------------------
 * PointCloud::Transform (geometry/pointcloud.cu:293-299): in place on the
 * caller's arrays; any of the three may be NULL. */
MI_ICP_API int mi_icp_transform(mi_icp_ctx* ctx, const float* T, float* xyz, float* normals,
                                float* covs, int64_t n, int mem_kind);
/* GeometryBase3D::GetMinBound / GetMaxBound / GetCenter of a cloud (geometry/geometry_base.h:47-52,
 * geometry/pointcloud.cu:205-215; utility::ComputeMinBound / ComputeMaxBound / ComputeCenter,
 * utility/eigen.inl:208-232).  Any of the outputs (host float[3]) may be NULL; an empty cloud
 * gives zero vectors.  The centre is the fp64 sum divided by n, rounded once (the reference
 * sums in fp32). */
MI_ICP_API int mi_icp_compute_bounds(mi_icp_ctx* ctx, const float* xyz, int64_t n, int mem_kind,
                                     float* min3, float* max3, float* center3);
/* GeometryBase3D::Translate / Scale / Rotate (geometry/geometry_base.h:58-90,
 * geometry/pointcloud.cu:225-242, geometry_utils.cu:150-270), in place:
 *   points  <- (R (p - center)) * scale + center + translate
 *   normals <- R n          covariances <- R C R^T           (only when R9 is given)
 * with exactly the reference functors' operations for the terms present: R9 (column-major 3x3,
 * Eigen::Matrix3f::data()), center3, translate3 may be NULL, use_scale = 0 skips the scaling.
 *   Translate(t, relative)  = affine(NULL, 0, 0, NULL, relative ? t : t - GetCenter(), points)
 *   Scale(s, center)        = affine(NULL, s, 1, center ? GetCenter() : NULL, NULL, points)
 *   Rotate(R, center)       = affine(R, 0, 0, center ? GetCenter() : NULL, NULL, points, normals, covs) */
MI_ICP_API int mi_icp_affine(mi_icp_ctx* ctx, const float* R9, float scale, int use_scale,
                             const float* center3, const float* translate3, float* xyz, float* normals,
                             float* covs, int64_t n, int mem_kind);
/* PointCloud::VoxelDownSample (geometry/down_sample.cu:170-273): outputs in
 * lexicographic voxel order; out arrays must hold n entries; *m receives the
 * voxel count (0 for voxel_size <= 0 or a too-small voxel, as the reference
 * returns an empty cloud).  normals / colors and their outputs may be NULL.
 * A voxel's values are added in fp64 in a fixed order -- the same result on
 * every run; in the input's order on dense grids (csrc/voxel_dense.h) and on
 * fine ones (a point or two per voxel), where the means equal the CPU oracle's
 * bit for bit (the reference's thrust::reduce_by_key adds in fp32 in an order
 * of its own choosing).  The entry point synchronises the context's stream. */
MI_ICP_API int mi_icp_voxel_downsample(mi_icp_ctx* ctx, const float* xyz, const float* normals,
                                       const float* colors, int64_t n, float voxel_size,
                                       float* out_xyz, float* out_normals, float* out_colors,
                                       int64_t* m, int mem_kind);
/* PointCloud::CreateFromDepthImage and PointCloud::CreateFromRGBDImage
 * (geometry/pointcloud_factory.cu:43-110,117-220,286-376) incl. the
 * RemoveNoneFinitePoints pass that follows (geometry/pointcloud.cu:40-54,360-385):
 * the depth-image side of the path's tracker callers (kinfu/kinfu.cpp:87-104).
 *   depth       [height][width], MI_ICP_DEPTH_F32 or MI_ICP_DEPTH_U16 (then value /
 *               (int)depth_scale, values >= (int)depth_trunc dropped, image.cu:339-348;
 *               both are truncated to int as the reference does)
 *   color       NULL, MI_ICP_COLOR_U8X3 [h][w][3] (scaled by 1/255) or
 *               MI_ICP_COLOR_F32X1 [h][w] (replicated to 3 channels)
 *   intrinsic4  fx, fy, cx, cy;   extrinsic: 4x4 column-major or NULL (identity);
 *               points are mapped by extrinsic^-1
 *   rgbd = 0    CreateFromDepthImage: pixel (row*stride, col*stride) of a
 *               (width/stride) x (height/stride) grid, depth <= 0 dropped, no
 *               colours or normals; non-finite points always removed
 *   rgbd = 1    CreateFromRGBDImage: stride must be 1; a pixel is kept when
 *               depth > 0 and (depth_cutoff <= 0 or depth < depth_cutoff);
 *               compute_normals: cross product of the 4-neighbourhood differences,
 *               flipped to z <= 0 (:161-199); valid_only = 0 keeps one point per
 *               pixel, rejected ones +inf.
 * Outputs must hold (width/stride)*(height/stride) points; *m = points written, in
 * pixel order. */
#define MI_ICP_DEPTH_F32 0
#define MI_ICP_DEPTH_U16 1
#define MI_ICP_COLOR_NONE 0
#define MI_ICP_COLOR_U8X3 1
#define MI_ICP_COLOR_F32X1 2
MI_ICP_API int mi_icp_create_from_depth(mi_icp_ctx* ctx, const void* depth, int depth_type,
                                        const void* color, int color_type, int width, int height,
                                        const float* intrinsic4, const float* extrinsic,
                                        float depth_scale, float depth_trunc, float depth_cutoff,
                                        int stride, int rgbd, int compute_normals, int valid_only,
                                        float* out_xyz, float* out_normals, float* out_colors,
                                        int64_t* m, int mem_kind);
/* odometry::ComputeRGBDOdometry (odometry/odometry.cu:833-943, odometry.h:43-53): the
 * transformation that maps the source RGB-D frame onto the target frame, and the 6x6
 * information matrix of the result.  The other in-repo caller of ComputeJTJandJTr /
 * SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.h:79-115) besides the ICP
 * estimators.
 *   colors / depths  [height][width] float32 (intensity in [0,1], depth in the scene's unit);
 *   intrinsic4       fx, fy, cx, cy;   odo_init: 4x4 column-major or NULL (identity);
 *   jacobian         MI_ICP_ODOMETRY_COLOR_TERM (rgbdodometry_jacobian.inl:41-94) or
 *                    MI_ICP_ODOMETRY_HYBRID_TERM (:96-172, the reference's default);
 *   option           OdometryOption (odometry_option.h:30-62); iterations[] coarsest level
 *                    first, as iteration_number_per_pyramid_level_.
 * Outputs: *success (0: the reference's failure result -- identity transformation and
 * identity information), transformation16 column-major, information36 row-major. */
#define MI_ICP_ODOMETRY_COLOR_TERM 0
#define MI_ICP_ODOMETRY_HYBRID_TERM 1
#define MI_ICP_ODOMETRY_MAX_LEVELS 8
typedef struct mi_icp_odometry_option {
    int32_t num_levels;                               /* 3 */
    int32_t iterations[MI_ICP_ODOMETRY_MAX_LEVELS];   /* {20, 10, 5} */
    float max_depth_diff;                             /* 0.03 */
    float min_depth;                                  /* 0.0 */
    float max_depth;                                  /* 4.0 */
    /* ComputeWeightedRGBDOdometry only */
    float nu;                                         /* 5.0 */
    float sigma2_init;                                /* 1.0 */
    float inv_sigma_mat_diag[6];                      /* 0 */
} mi_icp_odometry_option;
MI_ICP_API int mi_icp_compute_rgbd_odometry(mi_icp_ctx* ctx, const float* source_color,
                                            const float* source_depth, const float* target_color,
                                            const float* target_depth, int width, int height,
                                            const float* intrinsic4, const float* odo_init,
                                            int jacobian, const mi_icp_odometry_option* option,
                                            int* success, float* transformation16,
                                            double* information36, int mem_kind);
/* odometry::ComputeWeightedRGBDOdometry (odometry/odometry.cu:633-706,766-831,927-941): the same
 * with t-distribution weights on the correspondences (always the hybrid term) and a motion prior
 * inv_sigma_mat_diag . (prev_twist - velocity so far); twist6 receives the velocity of this call
 * (angle * axis, translation: utility::TransformMatrix4fToVector6f).  prev_twist6 may be NULL (0). */
MI_ICP_API int mi_icp_compute_weighted_rgbd_odometry(mi_icp_ctx* ctx, const float* source_color,
                                                     const float* source_depth, const float* target_color,
                                                     const float* target_depth, int width, int height,
                                                     const float* intrinsic4, const float* odo_init,
                                                     const float* prev_twist6,
                                                     const mi_icp_odometry_option* option, int* success,
                                                     float* transformation16, float* twist6,
                                                     double* information36, int mem_kind);
/* InitializePointCloudForGeneralizedICP's normals -> covariances
 * (registration/generalized_icp.cu:18-30,52-59). */
MI_ICP_API int mi_icp_covariances_from_normals(mi_icp_ctx* ctx, const float* normals,
                                               int64_t n, float epsilon, float* covs,
                                               int mem_kind);
/* PointCloud::EstimateNormals(KDTreeSearchParamKNN(knn))
 * (geometry/estimate_normals.cu:82-127), knn <= 100 (knn::NUM_MAX_NN,
 * knn/kdtree_search_param.h:26; lists of up to 32 neighbours take the faster kernel).
 * Workspace: the k-NN kernels (this call, mi_icp_search_knn, the colour gradients) keep
 * their candidates' INDICES in a device slab sized by the waves the GPU can hold at once,
 * not by the queries: ~50 MB at any list length and any number of points (a wave claims a
 * row of its XCD's pool when it starts and returns it when it ends), allocated by the first
 * such call of a context and reused by the later ones; EstimateNormals builds the cloud's
 * tree in a private scratch context of its own (the caller's target / source / loop
 * state survive the call). */
MI_ICP_API int mi_icp_estimate_normals_knn(mi_icp_ctx* ctx, const float* xyz, int64_t n,
                                           int knn, float* normals, int mem_kind);
/* PointCloud::EstimateNormals(KDTreeSearchParamRadius(radius, max_nn)): the max_nn
 * (<= 100) nearest points with d2 < radius^2, as KDTreeFlann::SearchRadius
 * feeds it (geometry/estimate_normals.cu:93-101, knn/kdtree_flann.inl:96-122);
 * fewer than 3 neighbours -> (0,0,1). */
MI_ICP_API int mi_icp_estimate_normals_radius(mi_icp_ctx* ctx, const float* xyz, int64_t n,
                                              float radius, int max_nn, float* normals,
                                              int mem_kind);

/* ---- knn::KDTreeFlann as a search object (knn/kdtree_flann.h:43-124) ---------
 * SearchKNN / SearchRadius (knn/kdtree_flann.inl:46-122) of arbitrary queries
 * float[nq][3] against the cloud given to mi_icp_set_target: per query the knn
 * (<= 100 = knn::NUM_MAX_NN) nearest target points -- with d2 < radius^2 when radius > 0, i.e.
 * SearchRadius(radius, max_nn = knn) -- ascending in distance (ties ascending in
 * index).  idx_out / d2_out are [nq][knn] row-major in the caller's query order,
 * original target indices, padded with -1 / +inf.  *found (optional) = number of
 * neighbours over all queries (the reference's return value for one query).
 * The queries are staged in the context's SOURCE slot: a cloud set with
 * mi_icp_set_source is replaced. */
MI_ICP_API int mi_icp_search_knn(mi_icp_ctx* ctx, const float* queries, int64_t nq, int knn,
                                 float radius, int32_t* idx_out, float* d2_out, int64_t* found,
                                 int mem_kind);

/* ---- Colored ICP (registration/colored_icp.cu) -----------------------------
 * Colours are float[n][3] RGB in the order of the cloud last given to
 * mi_icp_set_target / mi_icp_set_source (call these afterwards; a new
 * set_target / set_source drops them).  Only the intensity (r+g+b)/3 is kept,
 * as the reference's functors only use that (colored_icp.cu:91,195-200).  The
 * target needs normals (colored_icp.cu:222-224). */
MI_ICP_API int mi_icp_set_target_colors(mi_icp_ctx* ctx, const float* rgb, int mem_kind);
MI_ICP_API int mi_icp_set_source_colors(mi_icp_ctx* ctx, const float* rgb, int mem_kind);
/* TransformationEstimationForColoredICP::lambda_geometric_ (default 0.968;
 * values outside [0,1] fall back to it, colored_icp.cu:47-51).  Used by every
 * MI_ICP_EST_COLORED evaluation. */
MI_ICP_API int mi_icp_set_lambda_geometric(mi_icp_ctx* ctx, float lambda_geometric);
/* InitializePointCloudForColoredICP (colored_icp.cu:108-148): per target point
 * the intensity gradient in the tangent plane over the max_nn (<= 100) nearest
 * points within `radius` (the nearest -- the point itself -- excluded; fewer
 * than 4 others -> 0).  Kept on the device for MI_ICP_EST_COLORED;
 * gradients_out (float[nt][3], target's original order) may be NULL. */
MI_ICP_API int mi_icp_compute_color_gradients(mi_icp_ctx* ctx, float radius, int max_nn,
                                              float* gradients_out, int mem_kind);
/* registration::RegistrationColoredICP (colored_icp.cu:329-341) =
 * compute_color_gradients(2 * max_distance, 30) + registration_icp(COLORED).
 * params->det_thresh is the estimator's det_thresh (default 1e-6).
 * NOTE the reference's ComputeRMSE for this estimator returns the plain sum of
 * squared residuals (colored_icp.cu:302-306); mi_icp_compute_rmse(COLORED)
 * does the same. */
MI_ICP_API int mi_icp_registration_colored_icp(mi_icp_ctx* ctx, float max_distance,
                                               const float* init, const mi_icp_params* params,
                                               float lambda_geometric, mi_icp_result* out);

/* ---- multi-GPU (new: the reference is single-GPU) -------------------------
 * One context per rank/GPU, each holding the full target and its own shard of
 * the source.  After mi_icp_comm_init every accumulated system (32 doubles) is
 * summed across ranks before the solve, in a fixed order, so all ranks take
 * bit-identical steps.  How:
 *  - ranks of ONE node (<= 16) exchange through a MAILBOX in POSIX shared memory
 *    that every rank's GPU maps (csrc/mailbox.h): the reduction's finishing block
 *    posts its sums, waits for the peers' and goes on to the step -- an iteration
 *    stays two launches, no collective is launched;
 *  - otherwise (MI_ICP_NO_MAILBOX=1, more ranks, mailbox set-up failed) one
 *    ncclAllReduce(double, 32) on the context's stream, then the step kernel.
 * mi_icp_comm_init: RCCL communicator from a shared ncclUniqueId + the mailbox (named after
 * the id); ncclCommInitRank blocks until every rank has joined -- the call waits MI_ICP_COMM_INIT_MS (default 120000;
 * <= 0: for ever) for it and fails with MI_ICP_ERR_COMM past that.  mi_icp_comm_init_local: the mailbox alone, no RCCL -- all ranks pass the same
 * job_name (letters, digits, '_', '-'), one node only.  mi_icp_comm_kind: 0 none, 1 RCCL
 * all-reduce, 2 mailbox, 3 mailbox with device inboxes (every rank keeps an inbox in fine-grained device memory
 * that its peers open through HIP IPC and write their posts into -- GPU to GPU, polls stay local; any rank failing
 * to set that up keeps all of them on the host-memory box; USED when MI_ICP_MAILBOX=device is set ON RANK 0
 * (published through the box: the peers' own environment is not consulted) or
 * mi_icp_comm_autotune below measured them faster; MI_ICP_MAILBOX=host on rank 0: not even set up).  Set-up: rank 0 makes the box and waits (MI_ICP_MAIL_ATTACH_MS, default 30 s)
 * until every other rank has mapped and registered it; mi_icp_comm_init then lets the ranks agree over the
 * RCCL communicator whether ALL of them have it (else none uses it).  A peer that does not post within ~10 s
 * fails the call with MI_ICP_ERR_COMM; the communicator is void from then on (the ranks' exchange counters
 * are apart): every further call that would exchange fails the same way until mi_icp_comm_destroy /
 * mi_icp_comm_init[_local] have made a new one. */
MI_ICP_API int mi_icp_comm_unique_id(char* id128);
MI_ICP_API int mi_icp_comm_init(mi_icp_ctx* ctx, const char* id128, int nranks, int rank);
MI_ICP_API int mi_icp_comm_init_local(mi_icp_ctx* ctx, const char* job_name, int nranks, int rank);
MI_ICP_API int mi_icp_comm_kind(const mi_icp_ctx* ctx);
MI_ICP_API int mi_icp_comm_destroy(mi_icp_ctx* ctx);
/* Which way the ranks exchange is MEASURED, not assumed (collective: every rank calls it, right after
 * mi_icp_comm_init[_local] and before any registration).  Each available path -- [0] the box's host-memory words,
 * [1] device inboxes over HIP IPC (set up next to the box unless MI_ICP_MAILBOX=host), [2] the in-library
 * ncclAllReduce followed by a one-block kernel, as the loop's step kernel follows it -- performs `exchanges`
 * (<= 0: 200) all-reduces of a KNOWN vector that changes with every exchange; every total is checked exactly on the
 * device, the run is timed with events on the context's stream, and the ranks' CPUs gather the figures through the
 * shared-memory box (through the communicator when there is no box).  lat_us3[p]: microseconds per exchange, the
 * maximum over the ranks; -1: path not available, -2: it failed its self-test (timed out or summed wrongly) on some
 * rank -- such a path is skipped, never fatal, and the ranks' exchange counters are re-aligned behind it.  The
 * fastest path that passed everywhere becomes the one the loops use (identical on every rank: all decide on the same
 * gathered figures).  info4 = {chosen path (1 host words, 2 device inboxes, 3 RCCL; 0: single rank), ncclCommCount
 * of the communicator (0: none), exchanges timed per path, 1 if every path that was tried passed}.
 * MI_ICP_ERR_COMM when no path passed or the ranks did not meet (the communicator is void then). */
MI_ICP_API int mi_icp_comm_autotune(mi_icp_ctx* ctx, int exchanges, double* lat_us3, int* info4);
/* total source size over all ranks (fitness denominator, registration.cu:76) */
MI_ICP_API int mi_icp_set_global_source_count(mi_icp_ctx* ctx, int64_t n_total);
/* The order in which a source cloud is cut into per-rank shards: order_out[s] = original index
 * of the s-th point along the engine's space-filling (Morton) order, so that rank r of R takes
 * order_out[r*n/R .. (r+1)*n/R) -- one compact region of the target tree per GPU.  Computed on
 * the device (bounds, keys, radix sort); xyz and order_out on the side named by mem_kind. */
MI_ICP_API int mi_icp_spatial_order(mi_icp_ctx* ctx, const float* xyz, int64_t n, uint32_t* order_out,
                                    int mem_kind);

/* ---- per-iteration report (registration.cu:155-156: utility::LogDebug("ICP Iteration #{:d}: Fitness {:.4f},
 * RMSE {:.4f}", i, ...) at the top of every iteration) ----------------------------------------------
 * The loop runs on the device, several iterations per host look; with a callback set every update records
 * the evaluation it starts from, and the callback is called -- on the calling thread, in iteration order,
 * from inside mi_icp_registration_icp / mi_icp_icp_begin / mi_icp_icp_iterate / mi_icp_registration_colored_icp
 * -- once per iteration with the iteration's number (from 0), fitness and inlier RMSE: the values the reference
 * logs.  NULL removes it.  Costs one small device-to-host copy per look; nothing when unset. */
typedef void (*mi_icp_iteration_fn)(void* user, int iteration, float fitness, float inlier_rmse);
MI_ICP_API int mi_icp_set_iteration_callback(mi_icp_ctx* ctx, mi_icp_iteration_fn fn, void* user);

/* ---- instrumentation ----------------------------------------------------
 * enable != 0: every nearest-neighbour and reduction launch is bracketed by
 * hipEvents on the context's stream.  out[8] = {nn_ms_total, nn_launches,
 * reduce_ms_total, reduce_launches, build_ms_target, build_ms_source, halo builds
 * started by registration loops on this context (always counted), 0}. */
MI_ICP_API int mi_icp_set_profiling(mi_icp_ctx* ctx, int enable);
MI_ICP_API int mi_icp_get_profile(mi_icp_ctx* ctx, double* out8);

#ifdef __cplusplus
}
#endif
#endif /* MI_ICP_H_ */
