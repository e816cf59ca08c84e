// cupoch_amd.cpp -- the C++ surface of cupoch's ICP path
// (namespace cupoch::{geometry,registration,utility}; headers under
// cupoch_amd/cpp/include/cupoch) implemented over libmi_icp.so's C ABI.
// Same names, defaults and error behaviour as the reference
// (registration/registration.cu:106-172, transformation_estimation.cu,
// generalized_icp.cu:37-61,185-198, geometry/pointcloud.cu:293-299,
// down_sample.cu:170-273, estimate_normals.cu:82-127); no kernel lives here.
#include <typeinfo>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>

#include "cupoch/cupoch.h"
#include "cupoch/utility/console.h"
#include "mi_icp.h"

namespace cupoch {

// ---------------------------------------------------------------- utility
namespace utility {

static void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        // the reference prints and exit(0)s (utility/platform.cu:60-67); throwing is the
        // closest well-behaved equivalent for a library
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
}
void* device_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    void* p = nullptr;
    hip_check(hipMalloc(&p, bytes), "hipMalloc");
    return p;
}
void device_free(void* p) {
    if (p) (void)hipFree(p);
}
void copy_h2d(void* d, const void* s, size_t n) { hip_check(hipMemcpy(d, s, n, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
void copy_d2h(void* d, const void* s, size_t n) { hip_check(hipMemcpy(d, s, n, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
void copy_d2d(void* d, const void* s, size_t n) { hip_check(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice), "hipMemcpy D2D"); }

Eigen::Matrix4f TransformVector6fToMatrix4f(const Eigen::Vector6f& input) {
    Eigen::Matrix4f out;
    mi_icp_vector6_to_matrix4(input.data(), out.data());
    return out;
}

Eigen::Matrix4f InverseTransform(const Eigen::Matrix4f& input) {
    Eigen::Matrix4f inv = Eigen::Matrix4f::Identity();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv(r, c) = input(c, r);
    for (int r = 0; r < 3; ++r)
        inv(r, 3) = -(inv(r, 0) * input(0, 3) + inv(r, 1) * input(1, 3) + inv(r, 2) * input(2, 3));
    return inv;
}

std::pair<bool, Eigen::Matrix4f> SolveJacobianSystemAndObtainExtrinsicMatrix(
        const Eigen::Matrix6f& JTJ, const Eigen::Vector6f& JTr, float det_thresh) {
    double sys[32] = {0};
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) sys[k++] = JTJ(i, j);
    for (int i = 0; i < 6; ++i) sys[21 + i] = JTr(i);
    Eigen::Matrix4f T;
    const int ok = mi_icp_solve_system(sys, det_thresh, T.data());
    return {ok > 0, T};
}

namespace {
VerbosityLevel g_verbosity = VerbosityLevel::Info;   // the reference's default (spdlog: info)
}
void SetVerbosityLevel(VerbosityLevel level) { g_verbosity = level; }
VerbosityLevel GetVerbosityLevel() { return g_verbosity; }

}  // namespace utility

// ---------------------------------------------------------------- engine
namespace {

void LogError(const char* msg) { std::fprintf(stderr, "[cupoch_amd] Error: %s\n", msg); }    // console.h:54-56: logs, continues
void LogWarning(const char* msg) { std::fprintf(stderr, "[cupoch_amd] Warning: %s\n", msg); }

mi_icp_ctx* Engine() {
    static mi_icp_ctx* ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (mi_icp_create(dev, &ctx) != MI_ICP_OK)
            throw std::runtime_error("mi_icp_create failed: no MI355X device available (there is no CPU fallback)");
    });
    return ctx;
}

void Check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("mi_icp: ") + mi_icp_last_error(Engine()));
}

const float* Ptr(const utility::device_vector<Eigen::Vector3f>& v) { return v.empty() ? nullptr : v.data()->data(); }
const float* Ptr(const utility::device_vector<Eigen::Matrix3f>& v) { return v.empty() ? nullptr : v.data()->data(); }

// bumped whenever the engine's clouds are replaced: lets the generic ICP loop notice that a
// user estimator has used the engine (and that its own target tree / seeds are gone)
static unsigned long long g_load_generation = 0;

void LoadClouds(const geometry::PointCloud& source, const geometry::PointCloud& target) {
    mi_icp_ctx* c = Engine();
    ++g_load_generation;
    Check(mi_icp_set_target(c, Ptr(target.points_), target.HasNormals() ? Ptr(target.normals_) : nullptr,
                            target.HasCovariances() ? Ptr(target.covariances_) : nullptr,
                            (int64_t)target.points_.size(), MI_ICP_DEVICE));
    Check(mi_icp_set_source(c, Ptr(source.points_), source.HasNormals() ? Ptr(source.normals_) : nullptr,
                            source.HasCovariances() ? Ptr(source.covariances_) : nullptr,
                            (int64_t)source.points_.size(), MI_ICP_DEVICE));
}

registration::RegistrationResult MakeResult(const mi_icp_result& r) {
    registration::RegistrationResult out;
    std::memcpy(out.transformation_.data(), r.transformation, sizeof(float) * 16);
    out.fitness_ = r.fitness;
    out.inlier_rmse_ = r.inlier_rmse;
    int64_t count = 0;
    Check(mi_icp_get_correspondences(Engine(), nullptr, 0, &count, MI_ICP_DEVICE));
    out.correspondence_set_.resize((size_t)count);
    if (count > 0)
        Check(mi_icp_get_correspondences(Engine(), out.correspondence_set_.data()->data(), count, &count,
                                         MI_ICP_DEVICE));
    return out;
}

// estimator entry points on explicit correspondence sets
int EstType(registration::TransformationEstimationType t) { return (int)t; }

Eigen::Matrix4f ComputeWith(int est, float det_thresh, const geometry::PointCloud& source,
                            const geometry::PointCloud& target, const registration::CorrespondenceSet& corres) {
    if (corres.empty()) return Eigen::Matrix4f::Identity();
    LoadClouds(source, target);
    Check(mi_icp_set_correspondences(Engine(), corres.data()->data(), (int64_t)corres.size(), MI_ICP_DEVICE));
    Eigen::Matrix4f T;
    Check(mi_icp_compute_transformation(Engine(), est, nullptr, det_thresh, T.data()));
    return T;
}

float RmseWith(int est, const geometry::PointCloud& source, const geometry::PointCloud& target,
               const registration::CorrespondenceSet& corres) {
    if (corres.empty()) return 0.0f;
    LoadClouds(source, target);
    Check(mi_icp_set_correspondences(Engine(), corres.data()->data(), (int64_t)corres.size(), MI_ICP_DEVICE));
    float rmse = 0.0f;
    Check(mi_icp_compute_rmse(Engine(), est, nullptr, &rmse));
    return rmse;
}

}  // namespace

// ---------------------------------------------------------------- geometry
namespace geometry {

PointCloud& PointCloud::Transform(const Eigen::Matrix4f& transformation) {
    const size_t n = points_.size();
    Check(mi_icp_transform(Engine(), transformation.data(),
                           points_.empty() ? nullptr : points_.data()->data(),
                           normals_.size() == n && n ? normals_.data()->data() : nullptr,
                           covariances_.size() == n && n ? covariances_.data()->data() : nullptr,
                           (int64_t)n, MI_ICP_DEVICE));
    return *this;
}

std::shared_ptr<PointCloud> PointCloud::VoxelDownSample(float voxel_size) const {
    auto out = std::make_shared<PointCloud>();
    if (voxel_size <= 0.0f) {
        LogWarning("[VoxelDownSample] voxel_size <= 0.");  // down_sample.cu:173-176
        return out;
    }
    const size_t n = points_.size();
    if (n == 0) return out;
    const bool hn = HasNormals(), hc = HasColors();
    out->points_.resize(n);
    if (hn) out->normals_.resize(n);
    if (hc) out->colors_.resize(n);
    int64_t m = 0;
    Check(mi_icp_voxel_downsample(Engine(), Ptr(points_), hn ? Ptr(normals_) : nullptr,
                                  hc ? Ptr(colors_) : nullptr, (int64_t)n, voxel_size,
                                  out->points_.data()->data(), hn ? out->normals_.data()->data() : nullptr,
                                  hc ? out->colors_.data()->data() : nullptr, &m, MI_ICP_DEVICE));
    if (m == 0) LogWarning("[VoxelDownSample] voxel_size is too small.");
    out->points_.resize((size_t)m);
    if (hn) out->normals_.resize((size_t)m);
    if (hc) out->colors_.resize((size_t)m);
    return out;
}

bool PointCloud::EstimateNormals(const knn::KDTreeSearchParam& search_param) {
    normals_.resize(points_.size());
    if (points_.empty()) return true;
    switch (search_param.GetSearchType()) {
        case knn::KDTreeSearchParam::SearchType::Knn:
            Check(mi_icp_estimate_normals_knn(Engine(), Ptr(points_), (int64_t)points_.size(),
                                              ((const knn::KDTreeSearchParamKNN&)search_param).knn_,
                                              normals_.data()->data(), MI_ICP_DEVICE));
            return true;
        case knn::KDTreeSearchParam::SearchType::Radius: {
            const auto& p = (const knn::KDTreeSearchParamRadius&)search_param;
            Check(mi_icp_estimate_normals_radius(Engine(), Ptr(points_), (int64_t)points_.size(), p.radius_,
                                                 p.max_nn_, normals_.data()->data(), MI_ICP_DEVICE));
            return true;
        }
        default:
            LogError("Unknown search param type.");  // estimate_normals.cu:102-104
            return false;
    }
}

// GeometryBase3D on a cloud (pointcloud.cu:205-242): device reductions / one in-place kernel each
static void Bounds(const utility::device_vector<Eigen::Vector3f>& pts, Eigen::Vector3f* mn, Eigen::Vector3f* mx,
                   Eigen::Vector3f* center) {
    Check(mi_icp_compute_bounds(Engine(), Ptr(pts), (int64_t)pts.size(), MI_ICP_DEVICE, mn ? mn->data() : nullptr,
                                mx ? mx->data() : nullptr, center ? center->data() : nullptr));
}
Eigen::Vector3f PointCloud::GetMinBound() const {
    Eigen::Vector3f b;
    Bounds(points_, &b, nullptr, nullptr);
    return b;
}
Eigen::Vector3f PointCloud::GetMaxBound() const {
    Eigen::Vector3f b;
    Bounds(points_, nullptr, &b, nullptr);
    return b;
}
Eigen::Vector3f PointCloud::GetCenter() const {
    Eigen::Vector3f b;
    Bounds(points_, nullptr, nullptr, &b);
    return b;
}
AxisAlignedBoundingBox3 PointCloud::GetAxisAlignedBoundingBox() const {  // AxisAlignedBoundingBox<3>::CreateFromPoints
    Eigen::Vector3f mn, mx;
    Bounds(points_, &mn, &mx, nullptr);
    return AxisAlignedBoundingBox3(mn, mx);
}
PointCloud& PointCloud::Translate(const Eigen::Vector3f& translation, bool relative) {
    Eigen::Vector3f t = translation;
    if (!relative) t -= GetCenter();                       // geometry_utils.cu:155-158
    Check(mi_icp_affine(Engine(), nullptr, 0.0f, 0, nullptr, t.data(), points_.empty() ? nullptr : points_.data()->data(),
                        nullptr, nullptr, (int64_t)points_.size(), MI_ICP_DEVICE));
    return *this;
}
PointCloud& PointCloud::Scale(const float scale, bool center) {
    Eigen::Vector3f c = Eigen::Vector3f::Zero();
    const bool use_c = center && !points_.empty();          // geometry_utils.cu:170-173
    if (use_c) c = GetCenter();
    Check(mi_icp_affine(Engine(), nullptr, scale, 1, use_c ? c.data() : nullptr, nullptr,
                        points_.empty() ? nullptr : points_.data()->data(), nullptr, nullptr, (int64_t)points_.size(),
                        MI_ICP_DEVICE));
    return *this;
}
PointCloud& PointCloud::Rotate(const Eigen::Matrix3f& R, bool center) {
    Eigen::Vector3f c = Eigen::Vector3f::Zero();
    const bool use_c = center && !points_.empty();          // geometry_utils.cu:211-214
    if (use_c) c = GetCenter();
    const size_t n = points_.size();
    Check(mi_icp_affine(Engine(), R.data(), 0.0f, 0, use_c ? c.data() : nullptr, nullptr,
                        n ? points_.data()->data() : nullptr, normals_.size() == n && n ? normals_.data()->data() : nullptr,
                        covariances_.size() == n && n ? covariances_.data()->data() : nullptr, (int64_t)n,
                        MI_ICP_DEVICE));
    return *this;
}

// geometry::AxisAlignedBoundingBox<3> (geometry/boundingvolume.cu:300-354)
AxisAlignedBoundingBox3 AxisAlignedBoundingBox3::GetAxisAlignedBoundingBox() const { return *this; }
AxisAlignedBoundingBox3& AxisAlignedBoundingBox3::Transform(const Eigen::Matrix4f&) {
    LogError("A general transform of a AxisAlignedBoundingBox would not be axis aligned anymore, convert it to a "
             "OrientedBoundingBox first");
    return *this;
}
AxisAlignedBoundingBox3& AxisAlignedBoundingBox3::Translate(const Eigen::Vector3f& translation, bool relative) {
    if (relative) {
        min_bound_ += translation;
        max_bound_ += translation;
    } else {
        const Eigen::Vector3f half_extent = GetHalfExtent();
        min_bound_ = translation - half_extent;
        max_bound_ = translation + half_extent;
    }
    return *this;
}
AxisAlignedBoundingBox3& AxisAlignedBoundingBox3::Scale(const float scale, bool center) {
    if (center) {
        const Eigen::Vector3f c = GetCenter();
        min_bound_ = c + scale * (min_bound_ - c);
        max_bound_ = c + scale * (max_bound_ - c);
    } else {
        min_bound_ *= scale;
        max_bound_ *= scale;
    }
    return *this;
}
AxisAlignedBoundingBox3& AxisAlignedBoundingBox3::Rotate(const Eigen::Matrix3f&, bool) {
    LogError("A rotation of a AxisAlignedBoundingBox would not be axis aligned anymore, convert it to an "
             "OrientedBoundingBox first");
    return *this;
}

static std::shared_ptr<PointCloud> FromDepth(const Image& depth, const Image* color, int color_type,
                                             const camera::PinholeCameraIntrinsic& intrinsic,
                                             const Eigen::Matrix4f& extrinsic, float depth_scale, float depth_trunc,
                                             float depth_cutoff, int stride, bool rgbd, bool compute_normals,
                                             bool valid_only) {
    auto out = std::make_shared<PointCloud>();
    if (stride < 1 || depth.width_ <= 0 || depth.height_ <= 0) return out;
    const size_t count = (size_t)(depth.width_ / stride) * (size_t)(depth.height_ / stride);
    if (count == 0) return out;
    out->points_.resize(count);
    if (color) out->colors_.resize(count);
    if (compute_normals) out->normals_.resize(count);
    const float k4[4] = {intrinsic.fx_, intrinsic.fy_, intrinsic.cx_, intrinsic.cy_};
    int64_t m = 0;
    Check(mi_icp_create_from_depth(Engine(), depth.data_.data(),
                                   depth.bytes_per_channel_ == 2 ? MI_ICP_DEPTH_U16 : MI_ICP_DEPTH_F32,
                                   color ? color->data_.data() : nullptr, color_type, depth.width_, depth.height_, k4,
                                   extrinsic.data(), depth_scale, depth_trunc, depth_cutoff, stride, rgbd ? 1 : 0,
                                   compute_normals ? 1 : 0, valid_only ? 1 : 0, out->points_.data()->data(),
                                   compute_normals ? out->normals_.data()->data() : nullptr,
                                   color ? out->colors_.data()->data() : nullptr, &m, MI_ICP_DEVICE));
    out->points_.resize((size_t)m);
    if (color) out->colors_.resize((size_t)m);
    if (compute_normals) out->normals_.resize((size_t)m);
    return out;
}

std::shared_ptr<PointCloud> PointCloud::CreateFromDepthImage(const Image& depth,
                                                             const camera::PinholeCameraIntrinsic& intrinsic,
                                                             const Eigen::Matrix4f& extrinsic, float depth_scale,
                                                             float depth_trunc, int stride) {
    if (depth.num_of_channels_ == 1 && (depth.bytes_per_channel_ == 2 || depth.bytes_per_channel_ == 4))
        return FromDepth(depth, nullptr, MI_ICP_COLOR_NONE, intrinsic, extrinsic, depth_scale, depth_trunc, -1.0f,
                         stride, false, false, true);
    LogError("[PointCloud::CreateFromDepthImage] Unsupported image format.");  // pointcloud_factory.cu:348-350
    return std::make_shared<PointCloud>();
}

std::shared_ptr<PointCloud> PointCloud::CreateFromRGBDImage(const RGBDImage& image,
                                                            const camera::PinholeCameraIntrinsic& intrinsic,
                                                            const Eigen::Matrix4f& extrinsic,
                                                            bool project_valid_depth_only, float depth_cutoff,
                                                            bool compute_normals) {
    const Image& c = image.color_;
    const bool depth_ok = image.depth_.num_of_channels_ == 1 && image.depth_.bytes_per_channel_ == 4;
    int color_type = -1;
    if (c.data_.empty()) color_type = MI_ICP_COLOR_NONE;
    else if (c.bytes_per_channel_ == 1 && c.num_of_channels_ == 3) color_type = MI_ICP_COLOR_U8X3;
    else if (c.bytes_per_channel_ == 4 && c.num_of_channels_ == 1) color_type = MI_ICP_COLOR_F32X1;
    if (!depth_ok || color_type < 0 ||
        (color_type != MI_ICP_COLOR_NONE && (c.width_ != image.depth_.width_ || c.height_ != image.depth_.height_))) {
        LogError("[PointCloud::CreateFromRGBDImage] Unsupported image format.");  // pointcloud_factory.cu:373-375
        return std::make_shared<PointCloud>();
    }
    return FromDepth(image.depth_, color_type == MI_ICP_COLOR_NONE ? nullptr : &c, color_type, intrinsic, extrinsic,
                     1000.0f, 1000.0f, depth_cutoff, 1, true, compute_normals, project_valid_depth_only);
}

}  // namespace geometry

// ---------------------------------------------------------------- registration
namespace registration {

float TransformationEstimationPointToPoint::ComputeRMSE(const geometry::PointCloud& s, const geometry::PointCloud& t,
                                                        const CorrespondenceSet& c) const {
    return RmseWith(MI_ICP_EST_POINT_TO_POINT, s, t, c);
}
Eigen::Matrix4f TransformationEstimationPointToPoint::ComputeTransformation(const geometry::PointCloud& s,
                                                                            const geometry::PointCloud& t,
                                                                            const CorrespondenceSet& c) const {
    return ComputeWith(MI_ICP_EST_POINT_TO_POINT, -1.0f, s, t, c);
}
float TransformationEstimationPointToPlane::ComputeRMSE(const geometry::PointCloud& s, const geometry::PointCloud& t,
                                                        const CorrespondenceSet& c) const {
    if (!t.HasNormals()) return 0.0f;
    return RmseWith(MI_ICP_EST_POINT_TO_PLANE, s, t, c);
}
Eigen::Matrix4f TransformationEstimationPointToPlane::ComputeTransformation(const geometry::PointCloud& s,
                                                                            const geometry::PointCloud& t,
                                                                            const CorrespondenceSet& c) const {
    if (!t.HasNormals()) return Eigen::Matrix4f::Identity();
    return ComputeWith(MI_ICP_EST_POINT_TO_PLANE, det_thresh_, s, t, c);
}
float TransformationEstimationSymmetricMethod::ComputeRMSE(const geometry::PointCloud& s, const geometry::PointCloud& t,
                                                           const CorrespondenceSet& c) const {
    if (!s.HasNormals() || !t.HasNormals()) return 0.0f;
    return RmseWith(MI_ICP_EST_SYMMETRIC, s, t, c);
}
Eigen::Matrix4f TransformationEstimationSymmetricMethod::ComputeTransformation(const geometry::PointCloud& s,
                                                                               const geometry::PointCloud& t,
                                                                               const CorrespondenceSet& c) const {
    if (!s.HasNormals() || !t.HasNormals()) return Eigen::Matrix4f::Identity();
    return ComputeWith(MI_ICP_EST_SYMMETRIC, det_thresh_, s, t, c);
}
float TransformationEstimationForGeneralizedICP::ComputeRMSE(const geometry::PointCloud& s,
                                                             const geometry::PointCloud& t,
                                                             const CorrespondenceSet& c) const {
    if (!s.HasCovariances() || !t.HasCovariances()) return 0.0f;
    return RmseWith(MI_ICP_EST_GENERALIZED, s, t, c);
}
Eigen::Matrix4f TransformationEstimationForGeneralizedICP::ComputeTransformation(const geometry::PointCloud& s,
                                                                                 const geometry::PointCloud& t,
                                                                                 const CorrespondenceSet& c) const {
    if (!s.HasCovariances() || !t.HasCovariances()) return Eigen::Matrix4f::Identity();
    return ComputeWith(MI_ICP_EST_GENERALIZED, -1.0f, s, t, c);
}

RegistrationResult EvaluateRegistration(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                        float max_correspondence_distance, const Eigen::Matrix4f& transformation) {
    LoadClouds(source, target);
    mi_icp_result r;
    Check(mi_icp_evaluate_registration(Engine(), max_correspondence_distance, transformation.data(), &r));
    return MakeResult(r);
}

// exact type, not dynamic_cast: a user subclass of a built-in estimator may override
// ComputeTransformation, and the reference always makes the virtual call (registration.cu:157)
static bool IsBuiltin(const TransformationEstimation& e) {
    const std::type_info& t = typeid(e);
    return t == typeid(TransformationEstimationPointToPoint) || t == typeid(TransformationEstimationPointToPlane) ||
           t == typeid(TransformationEstimationSymmetricMethod) ||
           t == typeid(TransformationEstimationForGeneralizedICP);
}

RegistrationResult RegistrationICP(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                   float max_correspondence_distance, const Eigen::Matrix4f& init,
                                   const TransformationEstimation& estimation,
                                   const ICPConvergenceCriteria& criteria) {
    if (max_correspondence_distance <= 0.0f) LogError("Invalid max_correspondence_distance.");  // registration.cu:130-132
    const auto type = estimation.GetTransformationEstimationType();
    if ((type == TransformationEstimationType::PointToPlane || type == TransformationEstimationType::ColoredICP) &&
        !target.HasNormals())
        LogError("TransformationEstimationPointToPlane and TransformationEstimationColoredICP require "
                 "pre-computed target normal vectors.");  // registration.cu:134-143

    if (IsBuiltin(estimation)) {  // fused device loop
        float det = -1.0f;
        if (auto* p = dynamic_cast<const TransformationEstimationPointToPlane*>(&estimation)) det = p->det_thresh_;
        if (auto* p = dynamic_cast<const TransformationEstimationSymmetricMethod*>(&estimation)) det = p->det_thresh_;
        LoadClouds(source, target);
        mi_icp_params prm = {criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_, det};
        mi_icp_result r;
        // utility::LogDebug("ICP Iteration #{:d}: Fitness {:.4f}, RMSE {:.4f}", ...) (registration.cu:155-156)
        const bool debug = utility::GetVerbosityLevel() <= utility::VerbosityLevel::Debug;
        if (debug)
            mi_icp_set_iteration_callback(Engine(), [](void*, int i, float fitness, float rmse) {
                std::fprintf(stderr, "[cupoch_amd] Debug: ICP Iteration #%d: Fitness %.4f, RMSE %.4f\n", i, fitness, rmse);
            }, nullptr);
        const int rc = mi_icp_registration_icp(Engine(), EstType(type), max_correspondence_distance, init.data(), &prm, &r);
        if (debug) mi_icp_set_iteration_callback(Engine(), nullptr, nullptr);
        Check(rc);
        return MakeResult(r);
    }

    // user-defined estimator: the reference loop (registration.cu:144-171).  The engine keeps the
    // ORIGINAL source and the target tree and evaluates under the accumulated transformation
    // (seeded by the previous iteration's matches); the estimator still sees the transformed copy.
    // If the estimator itself goes through the engine, the clouds are simply loaded again.
    Eigen::Matrix4f transformation = init;
    geometry::PointCloud pcd = source;
    if (!init.isIdentity()) pcd.Transform(init);
    unsigned long long loaded = 0;
    auto evaluate = [&](const Eigen::Matrix4f& T) {
        if (loaded == 0 || loaded != g_load_generation) {
            LoadClouds(source, target);
            loaded = g_load_generation;
        }
        mi_icp_result r;
        Check(mi_icp_evaluate_registration(Engine(), max_correspondence_distance, T.data(), &r));
        RegistrationResult res = MakeResult(r);
        res.transformation_ = T;
        return res;
    };
    RegistrationResult result = evaluate(transformation);
    for (int i = 0; i < criteria.max_iteration_; ++i) {
        const Eigen::Matrix4f update = estimation.ComputeTransformation(pcd, target, result.correspondence_set_);
        transformation = update * transformation;
        pcd.Transform(update);
        RegistrationResult backup = result;
        result = evaluate(transformation);
        if (std::fabs(backup.fitness_ - result.fitness_) < criteria.relative_fitness_ &&
            std::fabs(backup.inlier_rmse_ - result.inlier_rmse_) < criteria.relative_rmse_)
            break;
    }
    return result;
}

// InitializePointCloudForGeneralizedICP (generalized_icp.cu:37-61)
static std::shared_ptr<geometry::PointCloud> InitializeForGICP(const geometry::PointCloud& pcd, float epsilon) {
    auto out = std::make_shared<geometry::PointCloud>(pcd);
    if (out->HasCovariances()) return out;
    if (!out->HasNormals()) out->EstimateNormals(knn::KDTreeSearchParamKNN(20));
    out->covariances_.resize(out->points_.size());
    if (!out->points_.empty())
        Check(mi_icp_covariances_from_normals(Engine(), Ptr(out->normals_), (int64_t)out->points_.size(), epsilon,
                                              out->covariances_.data()->data(), MI_ICP_DEVICE));
    return out;
}

RegistrationResult RegistrationGeneralizedICP(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                              float max_correspondence_distance, const Eigen::Matrix4f& init,
                                              const TransformationEstimationForGeneralizedICP& estimation,
                                              const ICPConvergenceCriteria& criteria) {
    return RegistrationICP(*InitializeForGICP(source, estimation.epsilon_), *InitializeForGICP(target, estimation.epsilon_),
                           max_correspondence_distance, init, estimation, criteria);
}

// registration::RegistrationColoredICP (colored_icp.cu:329-341)
RegistrationResult RegistrationColoredICP(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                          float max_distance, const Eigen::Matrix4f& init,
                                          const ICPConvergenceCriteria& criteria, float lambda_geometric,
                                          float det_thresh) {
    if (max_distance <= 0.0f) LogError("Invalid max_correspondence_distance.");
    if (!target.HasNormals())
        LogError("TransformationEstimationPointToPlane and TransformationEstimationColoredICP require "
                 "pre-computed target normal vectors.");
    LoadClouds(source, target);
    if (target.HasNormals() && target.HasColors())
        Check(mi_icp_set_target_colors(Engine(), Ptr(target.colors_), MI_ICP_DEVICE));
    if (source.HasColors()) Check(mi_icp_set_source_colors(Engine(), Ptr(source.colors_), MI_ICP_DEVICE));
    mi_icp_params prm = {criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_, det_thresh};
    mi_icp_result r;
    Check(mi_icp_registration_colored_icp(Engine(), max_distance, init.data(), &prm, lambda_geometric, &r));
    return MakeResult(r);
}

Eigen::Matrix4f_u Kabsch(const utility::device_vector<Eigen::Vector3f>& model,
                         const utility::device_vector<Eigen::Vector3f>& target) {
    // all points paired by index (kabsch.cu:122-): an identity correspondence set
    const size_t n = model.size();
    std::vector<Eigen::Vector2i> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = Eigen::Vector2i((int)i, (int)i);
    CorrespondenceSet corres(h);
    geometry::PointCloud s, t;
    s.points_ = model;
    t.points_ = target;
    return ComputeWith(MI_ICP_EST_POINT_TO_POINT, -1.0f, s, t, corres);
}

}  // namespace registration
// ============================================================================
// knn::KDTreeFlann (knn/kdtree_flann.h:43-124)
// ============================================================================
namespace knn {

static void CheckCtx(mi_icp_ctx* c, int rc) {
    if (rc < 0) throw std::runtime_error(std::string("mi_icp: ") + mi_icp_last_error(c));
}

KDTreeFlann::KDTreeFlann() {}
KDTreeFlann::KDTreeFlann(const utility::device_vector<Eigen::Vector3f>& data) { SetRawData(data); }
KDTreeFlann::~KDTreeFlann() {
    if (ctx_) mi_icp_destroy(ctx_);
}

bool KDTreeFlann::SetRawData(const utility::device_vector<Eigen::Vector3f>& data) {
    dataset_size_ = 0;
    if (data.empty()) {
        LogWarning("[KDTreeFlann::SetRawData] Failed due to no data.");   // kdtree_flann.inl:129-132
        return false;
    }
    if (!ctx_) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (mi_icp_create(dev, &ctx_) != MI_ICP_OK)
            throw std::runtime_error("mi_icp_create failed: no MI355X device available (there is no CPU fallback)");
    }
    CheckCtx(ctx_, mi_icp_set_target(ctx_, data.data()->data(), nullptr, nullptr, (int64_t)data.size(), MI_ICP_DEVICE));
    dataset_size_ = data.size();
    return true;
}

int KDTreeFlann::SearchMany(const utility::device_vector<Eigen::Vector3f>& query, int knn, float radius,
                            utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const {
    if (dataset_size_ == 0 || query.empty() || knn <= 0) return -1;   // kdtree_flann.cu:52-54,72-73
    indices.resize(query.size() * (size_t)knn);
    distance2.resize(query.size() * (size_t)knn);
    int64_t found = 0;
    CheckCtx(ctx_, mi_icp_search_knn(ctx_, query.data()->data(), (int64_t)query.size(), knn, radius, indices.data(),
                                     distance2.data(), &found, MI_ICP_DEVICE));
    return (int)found;
}

int KDTreeFlann::SearchKNN(const utility::device_vector<Eigen::Vector3f>& query, int knn,
                           utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const {
    return SearchMany(query, knn, 0.0f, indices, distance2);
}
int KDTreeFlann::SearchRadius(const utility::device_vector<Eigen::Vector3f>& query, float radius, int max_nn,
                              utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const {
    if (radius <= 0.0f) return -1;
    return SearchMany(query, max_nn, radius, indices, distance2);
}
int KDTreeFlann::Search(const utility::device_vector<Eigen::Vector3f>& query, const KDTreeSearchParam& param,
                        utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const {
    switch (param.GetSearchType()) {
        case KDTreeSearchParam::SearchType::Knn:
            return SearchKNN(query, ((const KDTreeSearchParamKNN&)param).knn_, indices, distance2);
        case KDTreeSearchParam::SearchType::Radius:
            return SearchRadius(query, ((const KDTreeSearchParamRadius&)param).radius_,
                                ((const KDTreeSearchParamRadius&)param).max_nn_, indices, distance2);
        default: return -1;
    }
}

// single query: results trimmed to the neighbours found, like FLANN's host overloads
static int One(const KDTreeFlann& tree, const Eigen::Vector3f& query, int knn, float radius, bool is_radius,
               thrust::host_vector<int>& indices, thrust::host_vector<float>& distance2) {
    utility::device_vector<Eigen::Vector3f> q(std::vector<Eigen::Vector3f>{query});
    utility::device_vector<int> di;
    utility::device_vector<float> dd;
    const int k = is_radius ? tree.SearchRadius(q, radius, knn, di, dd) : tree.SearchKNN(q, knn, di, dd);
    indices.clear();
    distance2.clear();
    if (k < 0) return k;
    const auto hi = di.to_host();
    const auto hd = dd.to_host();
    indices.assign(hi.begin(), hi.begin() + k);
    distance2.assign(hd.begin(), hd.begin() + k);
    return k;
}
int KDTreeFlann::SearchKNN(const Eigen::Vector3f& query, int knn, thrust::host_vector<int>& indices,
                           thrust::host_vector<float>& distance2) const {
    return One(*this, query, knn, 0.0f, false, indices, distance2);
}
int KDTreeFlann::SearchRadiusOne(const Eigen::Vector3f& query, float radius, int max_nn,
                                 thrust::host_vector<int>& indices, thrust::host_vector<float>& distance2) const {
    return One(*this, query, max_nn, radius, true, indices, distance2);
}
int KDTreeFlann::Search(const Eigen::Vector3f& query, const KDTreeSearchParam& param, thrust::host_vector<int>& indices,
                        thrust::host_vector<float>& distance2) const {
    switch (param.GetSearchType()) {
        case KDTreeSearchParam::SearchType::Knn:
            return SearchKNN(query, ((const KDTreeSearchParamKNN&)param).knn_, indices, distance2);
        case KDTreeSearchParam::SearchType::Radius:
            return SearchRadiusOne(query, ((const KDTreeSearchParamRadius&)param).radius_,
                                   ((const KDTreeSearchParamRadius&)param).max_nn_, indices, distance2);
        default: return -1;
    }
}

}  // namespace knn

// ---------------------------------------------------------------- odometry
namespace odometry {

static bool OdometryInputsOk(const geometry::RGBDImage& source, const geometry::RGBDImage& target) {
    auto is_float_image = [](const geometry::Image& im) { return im.num_of_channels_ == 1 && im.bytes_per_channel_ == 4; };
    const geometry::Image &sc = source.color_, &sd = source.depth_, &tc = target.color_, &td = target.depth_;
    const bool same = sc.width_ == tc.width_ && sc.height_ == tc.height_ && sd.width_ == td.width_ &&
                      sd.height_ == td.height_ && sc.width_ == sd.width_ && sc.height_ == sd.height_;
    if (same && is_float_image(sc) && is_float_image(sd) && is_float_image(tc) && is_float_image(td)) return true;
    LogWarning("[RGBDOdometry] Two RGBD pairs should be same in size.");  // odometry.cu:845-851
    return false;
}

static mi_icp_odometry_option OdometryOptionC(const OdometryOption& option) {
    mi_icp_odometry_option opt = {};
    opt.num_levels = (int32_t)option.iteration_number_per_pyramid_level_.size();
    for (int i = 0; i < opt.num_levels && i < MI_ICP_ODOMETRY_MAX_LEVELS; ++i)
        opt.iterations[i] = option.iteration_number_per_pyramid_level_[(size_t)i];
    opt.max_depth_diff = option.max_depth_diff_;
    opt.min_depth = option.min_depth_;
    opt.max_depth = option.max_depth_;
    opt.nu = option.nu_;
    opt.sigma2_init = option.sigma2_init_;
    for (int i = 0; i < 6; ++i) opt.inv_sigma_mat_diag[i] = option.inv_sigma_mat_diag_[i];
    return opt;
}

static Eigen::Matrix6f Info6(const double* info) {
    Eigen::Matrix6f I;
    for (int r = 0; r < 6; ++r)
        for (int c2 = 0; c2 < 6; ++c2) I(r, c2) = (float)info[r * 6 + c2];
    return I;
}

std::tuple<bool, Eigen::Matrix4f, Eigen::Matrix6f> ComputeRGBDOdometry(
        const geometry::RGBDImage& source, const geometry::RGBDImage& target,
        const camera::PinholeCameraIntrinsic& intrinsic, const Eigen::Matrix4f& odo_init,
        const RGBDOdometryJacobian& jacobian_method, const OdometryOption& option) {
    if (!OdometryInputsOk(source, target))
        return std::make_tuple(false, Eigen::Matrix4f::Identity(), Eigen::Matrix6f::Zero());
    const mi_icp_odometry_option opt = OdometryOptionC(option);
    const float k4[4] = {intrinsic.fx_, intrinsic.fy_, intrinsic.cx_, intrinsic.cy_};
    int ok = 0;
    Eigen::Matrix4f T;
    double info[36];
    Check(mi_icp_compute_rgbd_odometry(Engine(), (const float*)source.color_.data_.data(),
                                       (const float*)source.depth_.data_.data(),
                                       (const float*)target.color_.data_.data(),
                                       (const float*)target.depth_.data_.data(), source.color_.width_,
                                       source.color_.height_, k4, odo_init.data(), (int)jacobian_method.jacobian_type_,
                                       &opt, &ok, T.data(), info, MI_ICP_DEVICE));
    return std::make_tuple(ok != 0, T, Info6(info));
}

std::tuple<bool, Eigen::Matrix4f, Eigen::Vector6f, Eigen::Matrix6f> ComputeWeightedRGBDOdometry(
        const geometry::RGBDImage& source, const geometry::RGBDImage& target,
        const camera::PinholeCameraIntrinsic& intrinsic, const Eigen::Matrix4f& odo_init,
        const Eigen::Vector6f& prev_twist, const RGBDOdometryJacobian& /*always the hybrid term, odometry.cu:937-941*/,
        const OdometryOption& option) {
    if (!OdometryInputsOk(source, target))
        return std::make_tuple(false, Eigen::Matrix4f::Identity(), Eigen::Vector6f::Zero(), Eigen::Matrix6f::Zero());
    const mi_icp_odometry_option opt = OdometryOptionC(option);
    const float k4[4] = {intrinsic.fx_, intrinsic.fy_, intrinsic.cx_, intrinsic.cy_};
    int ok = 0;
    Eigen::Matrix4f T;
    Eigen::Vector6f twist;
    double info[36];
    Check(mi_icp_compute_weighted_rgbd_odometry(Engine(), (const float*)source.color_.data_.data(),
                                                (const float*)source.depth_.data_.data(),
                                                (const float*)target.color_.data_.data(),
                                                (const float*)target.depth_.data_.data(), source.color_.width_,
                                                source.color_.height_, k4, odo_init.data(), prev_twist.data(), &opt, &ok,
                                                T.data(), twist.data(), info, MI_ICP_DEVICE));
    return std::make_tuple(ok != 0, T, twist, Info6(info));
}

}  // namespace odometry

// ---------------------------------------------------------------- kinfu
namespace kinfu {

PointCloudPyramid CreatePointCloudPyramid(const std::vector<geometry::RGBDImage>& image_pyramid,
                                          const camera::PinholeCameraIntrinsic& intrinsic,
                                          const KinfuOption& option) {
    PointCloudPyramid out((size_t)option.num_pyramid_levels_);
    for (int i = 0; i < option.num_pyramid_levels_; ++i)
        out[(size_t)i] = geometry::PointCloud::CreateFromRGBDImage(image_pyramid[(size_t)i],
                                                                   intrinsic.CreatePyramidLevel((size_t)i),
                                                                   Eigen::Matrix4f::Identity(), true,
                                                                   option.depth_cutoff_, true);
    return out;
}

std::tuple<Eigen::Matrix4f, bool> PoseEstimation(const KinfuOption& option, const Eigen::Matrix4f& extrinsic,
                                                 const PointCloudPyramid& frame_data,
                                                 const PointCloudPyramid& target_data) {
    Eigen::Matrix4f cur = extrinsic;
    for (int level = option.num_pyramid_levels_ - 1; level >= 0; --level) {
        registration::ICPConvergenceCriteria criteria;
        criteria.max_iteration_ = option.icp_iterations_[(size_t)level];
        switch (option.tf_type_) {
            case registration::TransformationEstimationType::PointToPlane: {
                auto res = registration::RegistrationICP(*frame_data[(size_t)level], *target_data[(size_t)level],
                                                         option.distance_threshold_, cur,
                                                         registration::TransformationEstimationPointToPlane(100000),
                                                         criteria);
                cur = res.transformation_;
                break;
            }
            case registration::TransformationEstimationType::ColoredICP: {
                auto res = registration::RegistrationColoredICP(*frame_data[(size_t)level], *target_data[(size_t)level],
                                                                option.distance_threshold_, cur, criteria, 0.968f, 100000);
                cur = res.transformation_;
                break;
            }
            default:
                LogError("[KinfuPipeline::PoseEstimation] Unsupported transformation type.");
                break;
        }
    }
    return std::make_tuple(cur, true);
}

}  // namespace kinfu

}  // namespace cupoch
