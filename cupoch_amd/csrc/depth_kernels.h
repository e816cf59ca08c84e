// depth_kernels.h -- depth / RGB-D image -> point cloud, the data format on the input
// side of the ICP path when its caller is a tracker (KinfuPipeline::SurfaceMeasurement,
// kinfu.cpp:87-104): PointCloud::CreateFromDepthImage and CreateFromRGBDImage
// (geometry/pointcloud_factory.cu:43-110,117-220,286-376) followed by
// RemoveNoneFinitePoints (pointcloud.cu:40-54,360-385).
//
// The reference materialises a structured W x H cloud (points, colours, normals from the
// 4-neighbourhood of that cloud) and then stream-compacts three arrays.  Here a pixel's
// point is cheap to recompute (one depth load, 6 FMAs), so nothing structured is stored:
//   depth_valid_flags : 1 if the pixel's back-projected point is finite
//   (exclusive scan, primitives.h)
//   depth_emit        : back-project the pixel again, back-project its 4 neighbours for
//                       the normal, convert the colour, write at the compacted position.
// Traffic: 2 depth reads (+4 cached neighbour reads) + colour read per pixel, 12 B per
// attribute per valid pixel written -- against >= 72 B/pixel written and re-read by the
// structured form.
#pragma once
#include "device_utils.h"

namespace mi {

struct DepthArgs {
    const void* depth;   // [height][width] float32, or uint16 when depth_u16
    const void* color;   // null, [h][w][3] uint8, or [h][w] float32
    int width, height, stride;
    int depth_u16;       // uint16 input: value / depth_scale, >= depth_trunc -> 0 (image.cu:339-348)
    int color_kind;      // 0 none, 1 uint8 x 3, 2 float32 x 1
    int rgbd;            // CreateFromRGBDImage's validity rule (depth_cutoff) instead of CreateFromDepthImage's
    int depth_scale, depth_trunc;  // the reference holds both as int (image.cu:340-343)
    float depth_cutoff;
    float fx, fy, cx, cy;
    float pose[16];      // camera pose = extrinsic^-1, column-major
};

__device__ __forceinline__ float depth_value(const DepthArgs& a, int64_t pix) {
    if (a.depth_u16) {
        float f = (float)((const uint16_t*)a.depth)[pix];
        f /= (float)a.depth_scale;
        if (f >= (float)a.depth_trunc) f = 0.0f;
        return f;
    }
    return ((const float*)a.depth)[pix];
}

// point of pixel (row, col); +inf in every coordinate (and false) when the depth is rejected
__device__ __forceinline__ bool back_project(const DepthArgs& a, int row, int col, float p[3]) {
    const float d = depth_value(a, (int64_t)row * a.width + col);
    const bool ok = a.rgbd ? (d > 0.0f && (a.depth_cutoff <= 0.0f || a.depth_cutoff > d)) : !(d <= 0.0f);
    if (!ok) {
        p[0] = p[1] = p[2] = INFINITY;
        return false;
    }
    const float z = d;
    const float x = ((float)col - a.cx) * z / a.fx;
    const float y = ((float)row - a.cy) * z / a.fy;
#pragma unroll
    for (int r = 0; r < 3; ++r)
        p[r] = a.pose[r] * x + a.pose[4 + r] * y + a.pose[8 + r] * z + a.pose[12 + r];
    return true;
}

__device__ __forceinline__ bool finite3(const float p[3]) {
    return fabsf(p[0]) < INFINITY && fabsf(p[1]) < INFINITY && fabsf(p[2]) < INFINITY;
}

// output element idx -> pixel (pointcloud_factory.cu:61-64)
__device__ __forceinline__ void strided_pixel(const DepthArgs& a, int64_t idx, int& row, int& col) {
    const int sw = a.width / a.stride;
    row = (int)(idx / sw) * a.stride;
    col = (int)(idx % sw) * a.stride;
}

static __global__ __launch_bounds__(256) void depth_valid_flags(DepthArgs a, int64_t count, uint32_t* __restrict__ flags) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    int row, col;
    strided_pixel(a, idx, row, col);
    float p[3];
    back_project(a, row, col, p);
    flags[idx] = finite3(p) ? 1u : 0u;
}

// pos == null: every pixel is written at its own index (project_valid_depth_only = false)
static __global__ __launch_bounds__(256) void depth_emit(DepthArgs a, int64_t count, const uint32_t* __restrict__ pos,
                                                  float* __restrict__ out_xyz, float* __restrict__ out_nrm,
                                                  float* __restrict__ out_rgb) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    int row, col;
    strided_pixel(a, idx, row, col);
    float p[3];
    const bool depth_ok = back_project(a, row, col, p);
    const bool fin = finite3(p);
    if (pos && !fin) return;
    const int64_t o = pos ? (int64_t)pos[idx] : idx;
    out_xyz[o * 3] = p[0];
    out_xyz[o * 3 + 1] = p[1];
    out_xyz[o * 3 + 2] = p[2];
    if (out_rgb) {
        float c[3] = {INFINITY, INFINITY, INFINITY};  // rejected pixels carry +inf colours too (:153-158)
        if (depth_ok) {
            const int64_t pix = (int64_t)row * a.width + col;
            if (a.color_kind == 1) {
                const uint8_t* pc = (const uint8_t*)a.color + pix * 3;
                c[0] = (float)pc[0] / 255.0f;
                c[1] = (float)pc[1] / 255.0f;
                c[2] = (float)pc[2] / 255.0f;
            } else {
                c[0] = c[1] = c[2] = ((const float*)a.color)[pix] / 1.0f;
            }
        }
        out_rgb[o * 3] = c[0];
        out_rgb[o * 3 + 1] = c[1];
        out_rgb[o * 3 + 2] = c[2];
    }
    if (out_nrm) {
        // compute_normals_from_structured_pointcloud_functor (:161-199), stride 1.  The
        // reference's bounds test is `i < 1 || i >= height || j < 1 || j >= width`, so the
        // last column's right neighbour is the next row's first pixel (reproduced) and the
        // last row's lower neighbour lies past the buffer (undefined there; zero here, the
        // value the functor substitutes for any non-finite neighbour).
        float n[3] = {0.0f, 0.0f, 0.0f};
        if (row >= 1 && col >= 1) {
            const int64_t total = (int64_t)a.width * a.height;
            const int64_t pix = (int64_t)row * a.width + col;
            float l[3], r[3] = {0.0f, 0.0f, 0.0f}, u[3], d[3] = {0.0f, 0.0f, 0.0f};
            back_project(a, row, col - 1, l);
            if (!finite3(l)) l[0] = l[1] = l[2] = 0.0f;
            if (pix + 1 < total) {
                back_project(a, (int)((pix + 1) / a.width), (int)((pix + 1) % a.width), r);
                if (!finite3(r)) r[0] = r[1] = r[2] = 0.0f;
            }
            back_project(a, row - 1, col, u);
            if (!finite3(u)) u[0] = u[1] = u[2] = 0.0f;
            if (pix + a.width < total) {
                back_project(a, row + 1, col, d);
                if (!finite3(d)) d[0] = d[1] = d[2] = 0.0f;
            }
            const float hx = l[0] - r[0], hy = l[1] - r[1], hz = l[2] - r[2];
            const float vx = u[0] - d[0], vy = u[1] - d[1], vz = u[2] - d[2];
            n[0] = hy * vz - hz * vy;
            n[1] = hz * vx - hx * vz;
            n[2] = hx * vy - hy * vx;
            const float norm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (norm == 0.0f) {
                n[0] = n[1] = n[2] = 0.0f;
            } else {
                n[0] /= norm;
                n[1] /= norm;
                n[2] /= norm;
                if (n[2] > 0.0f) {
                    n[0] *= -1.0f;
                    n[1] *= -1.0f;
                    n[2] *= -1.0f;
                }
            }
        }
        out_nrm[o * 3] = n[0];
        out_nrm[o * 3 + 1] = n[1];
        out_nrm[o * 3 + 2] = n[2];
    }
}

}  // namespace mi
