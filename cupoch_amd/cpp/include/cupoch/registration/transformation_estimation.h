// cupoch/registration/transformation_estimation.h
// (reference: registration/transformation_estimation.h:30-146)
#pragma once
#include "cupoch/utility/device_vector.h"
#include "cupoch/utility/eigen.h"

namespace cupoch {
namespace geometry {
class PointCloud;
}
namespace registration {

typedef utility::device_vector<Eigen::Vector2i> CorrespondenceSet;

enum class TransformationEstimationType {
    Unspecified = 0,
    PointToPoint = 1,
    PointToPlane = 2,
    SymmetricMethod = 3,
    ColoredICP = 4,
    GeneralizedICP = 5,
};

class TransformationEstimation {
public:
    TransformationEstimation() {}
    virtual ~TransformationEstimation() {}
    virtual TransformationEstimationType GetTransformationEstimationType() const = 0;
    virtual float ComputeRMSE(const geometry::PointCloud& source, const geometry::PointCloud& target,
                              const CorrespondenceSet& corres) const = 0;
    virtual Eigen::Matrix4f ComputeTransformation(const geometry::PointCloud& source,
                                                  const geometry::PointCloud& target,
                                                  const CorrespondenceSet& corres) const = 0;
};

#define CUPOCH_AMD_ESTIMATION_BODY(TYPE)                                                        \
    TransformationEstimationType GetTransformationEstimationType() const override {            \
        return TransformationEstimationType::TYPE;                                              \
    }                                                                                           \
    float ComputeRMSE(const geometry::PointCloud& source, const geometry::PointCloud& target,   \
                      const CorrespondenceSet& corres) const override;                          \
    Eigen::Matrix4f ComputeTransformation(const geometry::PointCloud& source,                   \
                                          const geometry::PointCloud& target,                   \
                                          const CorrespondenceSet& corres) const override;

class TransformationEstimationPointToPoint : public TransformationEstimation {
public:
    TransformationEstimationPointToPoint() {}
    CUPOCH_AMD_ESTIMATION_BODY(PointToPoint)
};

class TransformationEstimationPointToPlane : public TransformationEstimation {
public:
    TransformationEstimationPointToPlane(float det_thresh = 1.0e-6) : det_thresh_(det_thresh) {}
    CUPOCH_AMD_ESTIMATION_BODY(PointToPlane)
    float det_thresh_;
};

class TransformationEstimationSymmetricMethod : public TransformationEstimation {
public:
    TransformationEstimationSymmetricMethod(float det_thresh = 1.0e-6) : det_thresh_(det_thresh) {}
    CUPOCH_AMD_ESTIMATION_BODY(SymmetricMethod)
    float det_thresh_;
};

/// registration/generalized_icp.h:14-40
class TransformationEstimationForGeneralizedICP : public TransformationEstimation {
public:
    TransformationEstimationForGeneralizedICP(float epsilon = 1e-3) : epsilon_(epsilon) {}
    CUPOCH_AMD_ESTIMATION_BODY(GeneralizedICP)
    float epsilon_;
};

#undef CUPOCH_AMD_ESTIMATION_BODY

}  // namespace registration
}  // namespace cupoch
