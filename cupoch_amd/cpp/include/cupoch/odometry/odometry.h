// cupoch/odometry/odometry.h -- odometry::ComputeRGBDOdometry (reference: odometry/odometry.h:43-53,
// odometry_option.h:30-62, rgbdodometry_jacobian.h:33-134).  The images are geometry::Image
// containers (float intensity + float depth, as RGBDImage::CreateFromColorAndDepth leaves them);
// everything runs in libmi_icp.so (mi_icp_compute_rgbd_odometry, mi_icp_compute_weighted_rgbd_odometry).
#pragma once
#include <tuple>
#include <vector>

#include "cupoch/camera/pinhole_camera_intrinsic.h"
#include "cupoch/geometry/image.h"
#include "cupoch/utility/eigen.h"

namespace cupoch {
namespace odometry {

class OdometryOption {
public:
    OdometryOption(const std::vector<int>& iteration_number_per_pyramid_level = {20, 10, 5},
                   float max_depth_diff = 0.03, float min_depth = 0.0, float max_depth = 4.0, float nu = 5.0,
                   float sigma2_init = 1.0, const Eigen::Vector6f& inv_sigma_mat_diag = Eigen::Vector6f::Zero())
        : iteration_number_per_pyramid_level_(iteration_number_per_pyramid_level),
          max_depth_diff_(max_depth_diff),
          min_depth_(min_depth),
          max_depth_(max_depth),
          nu_(nu),
          sigma2_init_(sigma2_init),
          inv_sigma_mat_diag_(inv_sigma_mat_diag) {}
    std::vector<int> iteration_number_per_pyramid_level_;
    float max_depth_diff_;
    float min_depth_;
    float max_depth_;
    float nu_;
    float sigma2_init_;
    Eigen::Vector6f inv_sigma_mat_diag_;
};

class RGBDOdometryJacobian {
public:
    enum OdometryJacobianType { COLOR_TERM = 0, HYBRID_TERM = 1 };
    explicit RGBDOdometryJacobian(OdometryJacobianType jacobian_type) : jacobian_type_(jacobian_type) {}
    virtual ~RGBDOdometryJacobian() {}
    OdometryJacobianType jacobian_type_;
};
class RGBDOdometryJacobianFromColorTerm : public RGBDOdometryJacobian {
public:
    RGBDOdometryJacobianFromColorTerm() : RGBDOdometryJacobian(COLOR_TERM) {}
};
class RGBDOdometryJacobianFromHybridTerm : public RGBDOdometryJacobian {
public:
    RGBDOdometryJacobianFromHybridTerm() : RGBDOdometryJacobian(HYBRID_TERM) {}
};

/// (is_success, transformation mapping the source frame onto the target frame, information matrix)
std::tuple<bool, Eigen::Matrix4f, Eigen::Matrix6f> ComputeRGBDOdometry(
        const geometry::RGBDImage& source, const geometry::RGBDImage& target,
        const camera::PinholeCameraIntrinsic& pinhole_camera_intrinsic = camera::PinholeCameraIntrinsic(),
        const Eigen::Matrix4f& odo_init = Eigen::Matrix4f::Identity(),
        const RGBDOdometryJacobian& jacobian_method = RGBDOdometryJacobianFromHybridTerm(),
        const OdometryOption& option = OdometryOption());

/// (is_success, transformation, velocity of this call as a twist, information matrix); t-distribution
/// weights + the motion prior inv_sigma_mat_diag . (prev_twist - velocity); always the hybrid term
std::tuple<bool, Eigen::Matrix4f, Eigen::Vector6f, Eigen::Matrix6f> ComputeWeightedRGBDOdometry(
        const geometry::RGBDImage& source, const geometry::RGBDImage& target,
        const camera::PinholeCameraIntrinsic& pinhole_camera_intrinsic = camera::PinholeCameraIntrinsic(),
        const Eigen::Matrix4f& odo_init = Eigen::Matrix4f::Identity(),
        const Eigen::Vector6f& prev_twist = Eigen::Vector6f::Zero(),
        const RGBDOdometryJacobian& jacobian_method = RGBDOdometryJacobianFromHybridTerm(),
        const OdometryOption& option = OdometryOption());

}  // namespace odometry
}  // namespace cupoch
