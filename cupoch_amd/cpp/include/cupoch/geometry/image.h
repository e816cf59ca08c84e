// cupoch/geometry/image.h -- geometry::Image / geometry::RGBDImage as containers
// (reference: geometry/image.h:52-110, geometry/rgbdimage.h:38-120): width, height,
// channels, bytes per channel and the pixel bytes on the device.  Image processing
// (pyramids, filters, conversions) is outside the ICP path and not provided.
#pragma once
#include <cstdint>
#include <vector>

#include "cupoch/utility/device_vector.h"

namespace cupoch {
namespace geometry {

class Image {
public:
    Image() {}
    Image& Prepare(int width, int height, int num_of_channels, int bytes_per_channel) {
        width_ = width;
        height_ = height;
        num_of_channels_ = num_of_channels;
        bytes_per_channel_ = bytes_per_channel;
        data_.resize((size_t)width * height * num_of_channels * bytes_per_channel);
        return *this;
    }
    void SetData(const std::vector<uint8_t>& bytes) { data_ = bytes; }
    bool IsEmpty() const { return width_ <= 0 || height_ <= 0 || data_.empty(); }

public:
    int width_ = 0;
    int height_ = 0;
    int num_of_channels_ = 0;
    int bytes_per_channel_ = 0;
    utility::device_vector<uint8_t> data_;
};

class RGBDImage {
public:
    RGBDImage() {}
    RGBDImage(const Image& color, const Image& depth) : color_(color), depth_(depth) {}

public:
    Image color_;
    Image depth_;
};

}  // namespace geometry
}  // namespace cupoch
