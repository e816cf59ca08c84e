// cupoch/io/class_io/pointcloud_io.h -- the point-cloud files on either side of the ICP path
// (reference: io/class_io/pointcloud_io.h:38-96, io/class_io/pointcloud_io.cpp,
// io/file_format/file_pcd.cu, io/file_format/file_ply.cu).  Same entry points and defaults;
// host-side parsing, the arrays are uploaded once into the cloud's device_vectors.
//   PCD  fields x y z [normal_x normal_y normal_z] [rgb | rgba], DATA ascii / binary /
//        binary_compressed (LZF, mi_icp_lzf_*), any field order / extra fields / SIZE-TYPE-COUNT;
//   PLY  element vertex with x y z [nx ny nz] [red green blue], ascii / binary_little_endian /
//        binary_big_endian, any scalar property types, other elements skipped.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "cupoch/geometry/pointcloud.h"

namespace cupoch {
namespace io {

/// Factory function to create a pointcloud from a file; an empty pointcloud if the read fails.
std::shared_ptr<geometry::PointCloud> CreatePointCloudFromFile(const std::string& filename,
                                                               const std::string& format = "auto",
                                                               bool print_progress = false);

/// The general entrance for reading a PointCloud from a file: dispatches on the extension
/// ("auto") or on `format` ("pcd", "ply"); false (and a warning) on failure.
bool ReadPointCloud(const std::string& filename, geometry::PointCloud& pointcloud, const std::string& format = "auto",
                    bool remove_nan_points = true, bool remove_infinite_points = true, bool print_progress = false);

/// The general entrance for writing a PointCloud to a file (extension decides).
bool WritePointCloud(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii = false,
                     bool compressed = false, bool print_progress = false);

/// The parse alone, into host arrays (what ReadPointCloud uploads): for callers that stage the cloud themselves, and
/// for checking the readers where there is no device (tests/cpp/test_io_malformed.cpp).  normals / colors come back
/// empty when the file has none.
bool ReadPointCloudToHost(const std::string& filename, std::vector<Eigen::Vector3f>& points,
                          std::vector<Eigen::Vector3f>& normals, std::vector<Eigen::Vector3f>& colors,
                          const std::string& format = "auto", bool remove_nan_points = true,
                          bool remove_infinite_points = true);

bool ReadPointCloudFromPLY(const std::string& filename, geometry::PointCloud& pointcloud, bool print_progress = false);
bool WritePointCloudToPLY(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii = false,
                          bool compressed = false, bool print_progress = false);
bool ReadPointCloudFromPCD(const std::string& filename, geometry::PointCloud& pointcloud, bool print_progress = false);
bool WritePointCloudToPCD(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii = false,
                          bool compressed = false, bool print_progress = false);

}  // namespace io
}  // namespace cupoch
