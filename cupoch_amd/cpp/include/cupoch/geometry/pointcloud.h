// cupoch/geometry/pointcloud.h -- geometry::PointCloud as the ICP path sees it
// (reference: geometry/pointcloud.h:43-263, geometry/geometry_base.h,
// geometry/geometry.h).  Same public members and layouts; the operations run
// HIP kernels through libmi_icp.so's C ABI (include/mi_icp.h).
#pragma once
#include <memory>

#include "cupoch/camera/pinhole_camera_intrinsic.h"
#include "cupoch/geometry/image.h"
#include "cupoch/knn/kdtree_search_param.h"
#include "cupoch/utility/device_vector.h"
#include "cupoch/utility/eigen.h"

namespace cupoch {
namespace geometry {

class Geometry {
public:
    enum class GeometryType { Unspecified = 0, PointCloud = 1 };
    virtual ~Geometry() {}
    GeometryType GetGeometryType() const { return type_; }
    int Dimension() const { return dimension_; }
    virtual Geometry& Clear() = 0;
    virtual bool IsEmpty() const = 0;

protected:
    Geometry(GeometryType type, int dimension) : type_(type), dimension_(dimension) {}

private:
    GeometryType type_;
    int dimension_;
};

class PointCloud : public Geometry {
public:
    PointCloud() : Geometry(GeometryType::PointCloud, 3) {}
    PointCloud(const thrust::host_vector<Eigen::Vector3f>& points)
        : Geometry(GeometryType::PointCloud, 3), points_(points) {}
    PointCloud(const PointCloud& other) = default;
    PointCloud& operator=(const PointCloud& other) = default;
    ~PointCloud() override {}

    void SetPoints(const thrust::host_vector<Eigen::Vector3f>& points) { points_ = points; }
    thrust::host_vector<Eigen::Vector3f> GetPoints() const { return points_.to_host(); }
    void SetNormals(const thrust::host_vector<Eigen::Vector3f>& normals) { normals_ = normals; }
    thrust::host_vector<Eigen::Vector3f> GetNormals() const { return normals_.to_host(); }
    void SetColors(const thrust::host_vector<Eigen::Vector3f>& colors) { colors_ = colors; }
    thrust::host_vector<Eigen::Vector3f> GetColors() const { return colors_.to_host(); }

    PointCloud& Clear() override {
        points_.clear();
        normals_.clear();
        colors_.clear();
        covariances_.clear();
        return *this;
    }
    bool IsEmpty() const override { return !HasPoints(); }
    bool HasPoints() const { return !points_.empty(); }
    bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }
    bool HasColors() const { return !points_.empty() && colors_.size() == points_.size(); }
    bool HasCovariances() const { return !points_.empty() && covariances_.size() == points_.size(); }

    Eigen::Vector3f GetMinBound() const;
    Eigen::Vector3f GetMaxBound() const;

    /// pointcloud.cu:293-299
    PointCloud& Transform(const Eigen::Matrix4f& transformation);
    /// down_sample.cu:170-273
    std::shared_ptr<PointCloud> VoxelDownSample(float voxel_size) const;
    /// estimate_normals.cu:82-127 (KNN search parameter; knn <= 32)
    bool EstimateNormals(const knn::KDTreeSearchParam& search_param = knn::KDTreeSearchParamKNN());

    /// pointcloud_factory.cu:329-351 (float or uint16 depth)
    static std::shared_ptr<PointCloud> CreateFromDepthImage(const Image& depth,
                                                            const camera::PinholeCameraIntrinsic& intrinsic,
                                                            const Eigen::Matrix4f& extrinsic = Eigen::Matrix4f::Identity(),
                                                            float depth_scale = 1000.0, float depth_trunc = 1000.0,
                                                            int stride = 1);
    /// pointcloud_factory.cu:353-376 (colour uint8 x 3 or float x 1; an empty colour image
    /// gives a cloud without colours)
    static std::shared_ptr<PointCloud> CreateFromRGBDImage(const RGBDImage& image,
                                                           const camera::PinholeCameraIntrinsic& intrinsic,
                                                           const Eigen::Matrix4f& extrinsic = Eigen::Matrix4f::Identity(),
                                                           bool project_valid_depth_only = true,
                                                           float depth_cutoff = -1.0f, bool compute_normals = false);

public:
    utility::device_vector<Eigen::Vector3f> points_;
    utility::device_vector<Eigen::Vector3f> normals_;
    utility::device_vector<Eigen::Vector3f> colors_;
    utility::device_vector<Eigen::Matrix3f> covariances_;
};

}  // namespace geometry
}  // namespace cupoch
