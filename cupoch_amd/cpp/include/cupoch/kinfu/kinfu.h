// cupoch/kinfu/kinfu.h -- the ICP side of kinfu::KinfuPipeline (reference:
// kinfu/kinfu.h:36-121, kinfu.cpp:87-143): the point-cloud pyramid of SurfaceMeasurement
// and the coarse-to-fine PoseEstimation.  The TSDF volume, its raycaster and the image
// filters are producers / consumers of the path and are not built; PoseEstimation is
// therefore a free function taking the option block instead of a pipeline member.
#pragma once
#include <memory>
#include <tuple>
#include <vector>

#include "cupoch/camera/pinhole_camera_intrinsic.h"
#include "cupoch/geometry/image.h"
#include "cupoch/geometry/pointcloud.h"
#include "cupoch/registration/transformation_estimation.h"

namespace cupoch {
namespace kinfu {

typedef std::vector<std::shared_ptr<geometry::PointCloud>> PointCloudPyramid;

class KinfuOption {
public:
    KinfuOption(int num_pyramid_levels = 4,
                float depth_cutoff = 3.0f,
                float distance_threshold = 0.5f,
                const std::vector<int>& icp_iterations = {20, 20, 20, 20},
                registration::TransformationEstimationType tf_type =
                        registration::TransformationEstimationType::PointToPlane)
        : num_pyramid_levels_(num_pyramid_levels),
          depth_cutoff_(depth_cutoff),
          distance_threshold_(distance_threshold),
          icp_iterations_(icp_iterations),
          tf_type_(tf_type) {}
    int num_pyramid_levels_;
    float depth_cutoff_;
    float distance_threshold_;
    std::vector<int> icp_iterations_;
    registration::TransformationEstimationType tf_type_;
};

/// kinfu.cpp:95-100: level i of an (already filtered) RGB-D pyramid ->
/// CreateFromRGBDImage(level, intrinsic.CreatePyramidLevel(i), I, true, depth_cutoff, true)
PointCloudPyramid CreatePointCloudPyramid(const std::vector<geometry::RGBDImage>& image_pyramid,
                                          const camera::PinholeCameraIntrinsic& intrinsic,
                                          const KinfuOption& option);

/// KinfuPipeline::PoseEstimation (kinfu.cpp:105-143)
std::tuple<Eigen::Matrix4f, bool> PoseEstimation(const KinfuOption& option,
                                                 const Eigen::Matrix4f& extrinsic,
                                                 const PointCloudPyramid& frame_data,
                                                 const PointCloudPyramid& target_data);

}  // namespace kinfu
}  // namespace cupoch
