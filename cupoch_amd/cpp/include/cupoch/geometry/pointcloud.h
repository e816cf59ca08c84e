// cupoch/geometry/pointcloud.h -- geometry::PointCloud as the ICP path sees it
// (reference: geometry/pointcloud.h:43-263, geometry/geometry_base.h,
// geometry/geometry.h).  Same public members and layouts; the operations run
// HIP kernels through libmi_icp.so's C ABI (include/mi_icp.h).
#pragma once
#include <cmath>
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
    enum class GeometryType { Unspecified = 0, PointCloud = 1, AxisAlignedBoundingBox = 13 };  // geometry.h:37-68
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

class AxisAlignedBoundingBox3;

/// geometry/geometry_base.h:33-90 with VectorT = Vector3f, MatrixT = Matrix3f, TransformT = Matrix4f
class GeometryBase3D : public Geometry {
protected:
    GeometryBase3D(GeometryType type) : Geometry(type, 3) {}

public:
    GeometryBase3D& Clear() override = 0;
    bool IsEmpty() const override = 0;
    virtual Eigen::Vector3f GetMinBound() const = 0;
    virtual Eigen::Vector3f GetMaxBound() const = 0;
    virtual Eigen::Vector3f GetCenter() const = 0;
    virtual AxisAlignedBoundingBox3 GetAxisAlignedBoundingBox() const = 0;
    virtual GeometryBase3D& Transform(const Eigen::Matrix4f& transformation) = 0;
    virtual GeometryBase3D& Translate(const Eigen::Vector3f& translation, bool relative = true) = 0;
    virtual GeometryBase3D& Scale(const float scale, bool center = true) = 0;
    virtual GeometryBase3D& Rotate(const Eigen::Matrix3f& R, bool center = true) = 0;
};

/// geometry::AxisAlignedBoundingBox<3> (geometry/boundingvolume.h:121-230): the value
/// PointCloud::GetAxisAlignedBoundingBox hands out -- bounds, centre, extent, volume and the
/// GeometryBase3D moves of a box (geometry/boundingvolume.cu; Rotate is not defined for an
/// axis-aligned box and logs an error there as here).
class AxisAlignedBoundingBox3 : public GeometryBase3D {
public:
    AxisAlignedBoundingBox3()
        : GeometryBase3D(GeometryType::AxisAlignedBoundingBox),
          min_bound_(Eigen::Vector3f::Zero()), max_bound_(Eigen::Vector3f::Zero()), color_(Eigen::Vector3f::Zero()) {}
    AxisAlignedBoundingBox3(const Eigen::Vector3f& min_bound, const Eigen::Vector3f& max_bound)
        : GeometryBase3D(GeometryType::AxisAlignedBoundingBox),
          min_bound_(min_bound), max_bound_(max_bound), color_(Eigen::Vector3f::Zero()) {}
    AxisAlignedBoundingBox3& Clear() override {
        min_bound_ = max_bound_ = Eigen::Vector3f::Zero();
        return *this;
    }
    bool IsEmpty() const override { return Volume() <= 0; }
    Eigen::Vector3f GetMinBound() const override { return min_bound_; }
    Eigen::Vector3f GetMaxBound() const override { return max_bound_; }
    Eigen::Vector3f GetCenter() const override { return (min_bound_ + max_bound_) * 0.5f; }
    AxisAlignedBoundingBox3 GetAxisAlignedBoundingBox() const override;
    AxisAlignedBoundingBox3& Transform(const Eigen::Matrix4f& transformation) override;
    AxisAlignedBoundingBox3& Translate(const Eigen::Vector3f& translation, bool relative = true) override;
    AxisAlignedBoundingBox3& Scale(const float scale, bool center = true) override;
    AxisAlignedBoundingBox3& Rotate(const Eigen::Matrix3f& R, bool center = true) override;
    Eigen::Vector3f GetExtent() const { return max_bound_ - min_bound_; }
    Eigen::Vector3f GetHalfExtent() const { return GetExtent() * 0.5f; }
    float GetMaxExtent() const {
        const Eigen::Vector3f e = GetExtent();
        return std::fmax(e[0], std::fmax(e[1], e[2]));
    }
    float Volume() const {
        const Eigen::Vector3f e = GetExtent();
        return e[0] * e[1] * e[2];
    }

public:
    Eigen::Vector3f min_bound_, max_bound_, color_;
};
template <int Dim>
struct AxisAlignedBoundingBoxOf;
template <>
struct AxisAlignedBoundingBoxOf<3> {
    typedef AxisAlignedBoundingBox3 type;
};
/// the reference spells the 3-D box AxisAlignedBoundingBox<3>
template <int Dim>
using AxisAlignedBoundingBox = typename AxisAlignedBoundingBoxOf<Dim>::type;

class PointCloud : public GeometryBase3D {
public:
    PointCloud() : GeometryBase3D(GeometryType::PointCloud) {}
    PointCloud(const thrust::host_vector<Eigen::Vector3f>& points)
        : GeometryBase3D(GeometryType::PointCloud), points_(points) {}
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

    /// pointcloud.cu:205-219 (device reductions; zero vectors for an empty cloud)
    Eigen::Vector3f GetMinBound() const override;
    Eigen::Vector3f GetMaxBound() const override;
    Eigen::Vector3f GetCenter() const override;
    AxisAlignedBoundingBox3 GetAxisAlignedBoundingBox() const override;
    /// pointcloud.cu:225-242
    PointCloud& Translate(const Eigen::Vector3f& translation, bool relative = true) override;
    PointCloud& Scale(const float scale, bool center = true) override;
    PointCloud& Rotate(const Eigen::Matrix3f& R, bool center = true) override;

    /// pointcloud.cu:293-299
    PointCloud& Transform(const Eigen::Matrix4f& transformation) override;
    /// down_sample.cu:170-273
    std::shared_ptr<PointCloud> VoxelDownSample(float voxel_size) const;
    /// estimate_normals.cu:82-127 (KNN or radius search parameter; up to knn::NUM_MAX_NN neighbours)
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
