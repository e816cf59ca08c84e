// cupoch/camera/pinhole_camera_intrinsic.h -- camera::PinholeCameraIntrinsic as the
// depth-image factories and the KinFu pose estimation use it
// (reference: camera/pinhole_camera_intrinsic.h:40-120, .cpp:40-92).
#pragma once
#include <cmath>
#include <utility>

#include "cupoch/utility/eigen.h"

namespace cupoch {
namespace camera {

class PinholeCameraIntrinsic {
public:
    PinholeCameraIntrinsic() {}
    PinholeCameraIntrinsic(int width, int height, float fx, float fy, float cx, float cy) {
        SetIntrinsics(width, height, fx, fy, cx, cy);
    }
    void SetIntrinsics(int width, int height, float fx, float fy, float cx, float cy) {
        width_ = width;
        height_ = height;
        fx_ = fx;
        fy_ = fy;
        cx_ = cx;
        cy_ = cy;
    }
    std::pair<float, float> GetFocalLength() const { return {fx_, fy_}; }
    std::pair<float, float> GetPrincipalPoint() const { return {cx_, cy_}; }
    bool IsValid() const { return width_ > 0 && height_ > 0; }
    /// pinhole_camera_intrinsic.cpp:82-92
    PinholeCameraIntrinsic CreatePyramidLevel(size_t level) const {
        if (level == 0 || width_ <= 0 || height_ <= 0) return *this;
        const float s = std::pow(0.5f, static_cast<float>(level));
        return PinholeCameraIntrinsic(width_ >> level, height_ >> level, fx_ * s, fy_ * s,
                                      (cx_ + 0.5f) * s - 0.5f, (cy_ + 0.5f) * s - 0.5f);
    }

public:
    int width_ = -1;
    int height_ = -1;
    float fx_ = 0.0f, fy_ = 0.0f, cx_ = 0.0f, cy_ = 0.0f;  // the reference keeps these in a 3x3 intrinsic_matrix_
};

}  // namespace camera
}  // namespace cupoch
