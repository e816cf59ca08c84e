// cupoch/registration/registration.h (reference: registration/registration.h:28-94)
#pragma once
#include "cupoch/registration/transformation_estimation.h"

namespace cupoch {
namespace registration {

class ICPConvergenceCriteria {
public:
    ICPConvergenceCriteria(float relative_fitness = 1e-6, float relative_rmse = 1e-6,
                           int max_iteration = 30)
        : relative_fitness_(relative_fitness), relative_rmse_(relative_rmse), max_iteration_(max_iteration) {}
    float relative_fitness_;
    float relative_rmse_;
    int max_iteration_;
};

class RegistrationResult {
public:
    RegistrationResult(const Eigen::Matrix4f& transformation = Eigen::Matrix4f::Identity())
        : transformation_(transformation) {}
    void SetCorrespondenceSet(const thrust::host_vector<Eigen::Vector2i>& corres) { correspondence_set_ = corres; }
    thrust::host_vector<Eigen::Vector2i> GetCorrespondenceSet() const { return correspondence_set_.to_host(); }

    Eigen::Matrix4f_u transformation_;
    CorrespondenceSet correspondence_set_;
    float inlier_rmse_ = 0.0f;
    float fitness_ = 0.0f;
};

RegistrationResult EvaluateRegistration(const geometry::PointCloud& source,
                                        const geometry::PointCloud& target,
                                        float max_correspondence_distance,
                                        const Eigen::Matrix4f& transformation = Eigen::Matrix4f::Identity());

RegistrationResult RegistrationICP(
        const geometry::PointCloud& source, const geometry::PointCloud& target,
        float max_correspondence_distance, const Eigen::Matrix4f& init = Eigen::Matrix4f::Identity(),
        const TransformationEstimation& estimation = TransformationEstimationPointToPoint(),
        const ICPConvergenceCriteria& criteria = ICPConvergenceCriteria());

/// registration/generalized_icp.h:59-66
RegistrationResult RegistrationGeneralizedICP(
        const geometry::PointCloud& source, const geometry::PointCloud& target,
        float max_correspondence_distance, const Eigen::Matrix4f& init = Eigen::Matrix4f::Identity(),
        const TransformationEstimationForGeneralizedICP& estimation = TransformationEstimationForGeneralizedICP(),
        const ICPConvergenceCriteria& criteria = ICPConvergenceCriteria());

/// registration/colored_icp.h:40-48
RegistrationResult RegistrationColoredICP(
        const geometry::PointCloud& source, const geometry::PointCloud& target, float max_distance,
        const Eigen::Matrix4f& init = Eigen::Matrix4f::Identity(),
        const ICPConvergenceCriteria& criteria = ICPConvergenceCriteria(), float lambda_geometric = 0.968,
        float det_thresh = 1.0e-6);

/// registration/kabsch.h: all points paired by index
Eigen::Matrix4f_u Kabsch(const utility::device_vector<Eigen::Vector3f>& model,
                         const utility::device_vector<Eigen::Vector3f>& target);

}  // namespace registration
}  // namespace cupoch
