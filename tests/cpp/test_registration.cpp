// C++ drop-in check: the same calls a cupoch user writes (examples/cpp/registration.cpp,
// src/tests/registration/kabsch.cpp), against libcupoch_amd.so.  Prints one JSON
// object; tests/test_gpu_cpp.py asserts on it.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "cupoch/cupoch.h"

using namespace cupoch;
using Eigen::Matrix4f;
using Eigen::Vector3f;

static Matrix4f Rigid(float angle, float ax, float ay, float az, float tx, float ty, float tz) {
    const float n = std::sqrt(ax * ax + ay * ay + az * az);
    Eigen::Vector6f x;
    x[0] = angle * ax / n; x[1] = angle * ay / n; x[2] = angle * az / n;
    x[3] = tx; x[4] = ty; x[5] = tz;
    return utility::TransformVector6fToMatrix4f(x);
}

static float Fro(const Matrix4f& a, const Matrix4f& b) { return (a - b).norm(); }


// depth frame (z in the camera frame) of the height field z = f(x, y) seen from `pose` (camera -> world)
static float Surface(float x, float y) { return 3.0f + 0.3f * std::sin(1.5f * x) + 0.3f * std::cos(1.2f * y) + 0.1f * x * y; }
static geometry::Image RenderDepth(const camera::PinholeCameraIntrinsic& k, const Matrix4f& pose) {
    std::vector<float> d((size_t)k.width_ * k.height_);
    for (int v = 0; v < k.height_; ++v)
        for (int u = 0; u < k.width_; ++u) {
            const float dc[3] = {(u - k.cx_) / k.fx_, (v - k.cy_) / k.fy_, 1.0f};
            float dir[3];
            for (int r = 0; r < 3; ++r) dir[r] = pose(r, 0) * dc[0] + pose(r, 1) * dc[1] + pose(r, 2) * dc[2];
            float s = 3.0f;
            for (int it = 0; it < 40; ++it) {
                const float px = pose(0, 3) + s * dir[0], py = pose(1, 3) + s * dir[1], pz = pose(2, 3) + s * dir[2];
                s += (Surface(px, py) - pz) / dir[2];
            }
            d[(size_t)v * k.width_ + u] = s;
        }
    geometry::Image img;
    img.Prepare(k.width_, k.height_, 1, 4);
    std::vector<uint8_t> bytes(d.size() * 4);
    std::memcpy(bytes.data(), d.data(), bytes.size());
    img.SetData(bytes);
    return img;
}

// intensity frame that goes with RenderDepth: a smooth world-space texture
static geometry::Image RenderIntensity(const camera::PinholeCameraIntrinsic& k, const Matrix4f& pose) {
    std::vector<float> c((size_t)k.width_ * k.height_);
    for (int v = 0; v < k.height_; ++v)
        for (int u = 0; u < k.width_; ++u) {
            const float dc[3] = {(u - k.cx_) / k.fx_, (v - k.cy_) / k.fy_, 1.0f};
            float dir[3];
            for (int r = 0; r < 3; ++r) dir[r] = pose(r, 0) * dc[0] + pose(r, 1) * dc[1] + pose(r, 2) * dc[2];
            float s = 3.0f;
            for (int it = 0; it < 40; ++it)
                s += (Surface(pose(0, 3) + s * dir[0], pose(1, 3) + s * dir[1]) - (pose(2, 3) + s * dir[2])) / dir[2];
            const float px = pose(0, 3) + s * dir[0], py = pose(1, 3) + s * dir[1];
            c[(size_t)v * k.width_ + u] = 0.5f + 0.2f * std::sin(3.0f * px) * std::cos(2.5f * py) + 0.2f * std::cos(2.0f * py + px);
        }
    geometry::Image img;
    img.Prepare(k.width_, k.height_, 1, 4);
    std::vector<uint8_t> bytes(c.size() * 4);
    std::memcpy(bytes.data(), c.data(), bytes.size());
    img.SetData(bytes);
    return img;
}

// a user-defined estimator: exercises the virtual interface / generic loop
class MyPointToPlane : public registration::TransformationEstimation {
public:
    registration::TransformationEstimationType GetTransformationEstimationType() const override {
        return registration::TransformationEstimationType::Unspecified;
    }
    float ComputeRMSE(const geometry::PointCloud& s, const geometry::PointCloud& t,
                      const registration::CorrespondenceSet& c) const override {
        return inner.ComputeRMSE(s, t, c);
    }
    Matrix4f ComputeTransformation(const geometry::PointCloud& s, const geometry::PointCloud& t,
                                   const registration::CorrespondenceSet& c) const override {
        ++calls;
        return inner.ComputeTransformation(s, t, c);
    }
    registration::TransformationEstimationPointToPlane inner{-1.0f};
    mutable int calls = 0;
};

int main(int argc, char** argv) {
    const std::string io_dir = argc > 1 ? argv[1] : "/tmp";
    const int n = 50000;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::normal_distribution<float> N(0.0f, 1.0f);
    std::vector<Vector3f> tgt(n), nrm(n);
    for (int i = 0; i < n; ++i) {
        tgt[i] = Vector3f(U(rng), U(rng), U(rng));
        Vector3f v(N(rng), N(rng), N(rng));
        const float l = v.norm();
        nrm[i] = Vector3f(v[0] / l, v[1] / l, v[2] / l);
    }
    const float s = std::pow((float)n, -1.0f / 3.0f);
    const Matrix4f T_gt = Rigid(0.2f * s, 1, 2, 3, 0.1f * s, -0.1f * s, 0.1f * s);
    const Matrix4f T_inv = utility::InverseTransform(T_gt);

    geometry::PointCloud target(tgt);
    target.SetNormals(nrm);
    geometry::PointCloud source = target;       // deep copy
    source.Transform(T_inv);                    // source = T_gt^-1 * target

    std::printf("{");
    // Transform round trip (src/tests/geometry/pointcloud.cpp:143-174)
    {
        geometry::PointCloud p = target;
        p.Transform(T_gt);
        p.Transform(T_inv);
        auto a = p.GetPoints();
        double worst = 0;
        for (int i = 0; i < n; ++i) worst = std::fmax(worst, (a[i] - tgt[i]).norm());
        std::printf("\"roundtrip_max\": %.3g, ", worst);
    }
    const float r = 2.0f * s;
    registration::ICPConvergenceCriteria crit;  // defaults 1e-6, 1e-6, 30
    {
        auto res = registration::RegistrationICP(source, target, r);  // default: point-to-point, identity init
        std::printf("\"p2p_err\": %.3g, \"p2p_fitness\": %.6f, \"p2p_ncorr\": %zu, ", Fro(res.transformation_, T_gt),
                    res.fitness_, res.correspondence_set_.size());
        auto cs = res.GetCorrespondenceSet();
        bool asc = true;
        for (size_t i = 1; i < cs.size(); ++i) asc = asc && cs[i][0] > cs[i - 1][0];
        std::printf("\"p2p_corr_ascending\": %s, ", asc ? "true" : "false");
    }
    {
        auto res = registration::RegistrationICP(source, target, r, Matrix4f::Identity(),
                                                 registration::TransformationEstimationPointToPlane(-1.0f), crit);
        std::printf("\"pt2pl_err\": %.3g, \"pt2pl_rmse\": %.3g, ", Fro(res.transformation_, T_gt), res.inlier_rmse_);
        auto ev = registration::EvaluateRegistration(source, target, r, res.transformation_);
        std::printf("\"eval_fitness\": %.6f, ", ev.fitness_);
    }
    {
        auto res = registration::RegistrationICP(source, target, r, Matrix4f::Identity(),
                                                 registration::TransformationEstimationSymmetricMethod(-1.0f), crit);
        std::printf("\"sym_err\": %.3g, ", Fro(res.transformation_, T_gt));
    }
    {
        auto res = registration::RegistrationGeneralizedICP(source, target, r);   // covariances from normals, eps 1e-3
        std::printf("\"gicp_err\": %.3g, ", Fro(res.transformation_, T_gt));
    }
    {
        MyPointToPlane mine;
        auto res = registration::RegistrationICP(source, target, r, Matrix4f::Identity(), mine,
                                                 registration::ICPConvergenceCriteria(1e-6, 1e-6, 8));
        std::printf("\"custom_err\": %.3g, \"custom_calls\": %d, ", Fro(res.transformation_, T_gt), mine.calls);
    }
    {   // no target normals: logged error, identity updates (registration.cu:134-143, transformation_estimation.cu:199-200)
        geometry::PointCloud bare(tgt);
        auto res = registration::RegistrationICP(source, bare, r, Matrix4f::Identity(),
                                                 registration::TransformationEstimationPointToPlane(), crit);
        std::printf("\"no_normals_is_identity\": %s, ", res.transformation_.isIdentity() ? "true" : "false");
    }
    {   // RegistrationColoredICP on a textured plane: the in-plane motion is only visible in the colours
        const int m = 20000;
        std::vector<Vector3f> cp(m), cn(m, Vector3f(0, 0, 1)), cc(m);
        for (int i = 0; i < m; ++i) {
            const float x = U(rng), y = U(rng);
            cp[i] = Vector3f(100 * x, 100 * y, 0);
            const float in = 0.5f + 0.4f * std::sin(9 * x) * std::cos(7 * y);
            cc[i] = Vector3f(in, 0.9f * in, 0.8f * in);
        }
        geometry::PointCloud ct(cp);
        ct.SetNormals(cn);
        ct.SetColors(cc);
        const Matrix4f Tc = Rigid(0.01f, 0, 0, 1, 0.4f, -0.3f, 0.0f);
        geometry::PointCloud cs = ct;
        cs.Transform(utility::InverseTransform(Tc));
        auto res = registration::RegistrationColoredICP(cs, ct, 3.0f);   // defaults: identity, 30 its, 0.968, 1e-6
        auto pl = registration::RegistrationICP(cs, ct, 3.0f, Matrix4f::Identity(),
                                                registration::TransformationEstimationPointToPlane(-1.0f), crit);
        std::printf("\"colored_err\": %.3g, \"colored_vs_plane_err\": %.3g, \"colored_motion\": %.3g, ",
                    Fro(res.transformation_, Tc), Fro(pl.transformation_, Tc), Fro(Matrix4f::Identity(), Tc));
        geometry::PointCloud grey(cp);   // no colours on the source: identity updates (colored_icp.cu:222-224)
        auto none = registration::RegistrationColoredICP(grey, ct, 3.0f);
        std::printf("\"colored_no_colors_is_identity\": %s, ", none.transformation_.isIdentity() ? "true" : "false");
    }
    {   // knn::KDTreeFlann as a search object, against brute force
        knn::KDTreeFlann tree(target.points_);
        const Vector3f q(0.41f, 0.52f, 0.63f);
        thrust::host_vector<int> idx;
        thrust::host_vector<float> d2;
        const int k = tree.SearchKNN(q, 10, idx, d2);
        std::vector<std::pair<float, int>> all(n);
        for (int i = 0; i < n; ++i) {
            const Vector3f d = tgt[i] - q;
            all[i] = {d[2] * d[2] + (d[1] * d[1] + d[0] * d[0]), i};
        }
        std::sort(all.begin(), all.end());
        bool same = k == 10 && idx.size() == 10;
        for (int i = 0; same && i < 10; ++i) same = idx[i] == all[i].second && std::fabs(d2[i] - all[i].first) < 1e-9f;
        const int kr = tree.SearchRadius<Vector3f>(q, 0.03f, 32, idx, d2);
        int expect = 0;
        for (auto& a : all) expect += a.first < 0.03f * 0.03f ? 1 : 0;
        utility::device_vector<int> di;
        utility::device_vector<float> dd;
        const int kb = tree.Search(source.points_, knn::KDTreeSearchParamKNN(4), di, dd);   // batch form
        knn::KDTreeFlann empty;
        const int kr_size = (int)idx.size();
        std::printf("\"kdtree_knn_ok\": %s, \"kdtree_radius\": [%d, %d, %d], \"kdtree_batch\": %d, \"kdtree_empty\": %d, ",
                    same ? "true" : "false", kr, kr_size, std::min(expect, 32), kb, empty.SearchKNN(q, 3, idx, d2));
    }
    {   // tracker-side callers: depth frames -> cloud pyramids -> kinfu::PoseEstimation (kinfu.cpp:87-143)
        const camera::PinholeCameraIntrinsic k0(320, 240, 262.5f, 262.5f, 159.5f, 119.5f);
        const Matrix4f pose_b = Rigid(0.02f, 1, 2, 3, 0.02f, -0.02f, 0.01f);
        kinfu::KinfuOption opt(2, 6.0f, 0.03f, {10, 10});
        std::vector<geometry::RGBDImage> fa, fb;
        for (int l = 0; l < 2; ++l) {
            const auto kl = k0.CreatePyramidLevel((size_t)l);
            fa.emplace_back(geometry::Image(), RenderDepth(kl, Matrix4f::Identity()));
            fb.emplace_back(geometry::Image(), RenderDepth(kl, pose_b));
        }
        auto model = kinfu::CreatePointCloudPyramid(fa, k0, opt);
        auto frame = kinfu::CreatePointCloudPyramid(fb, k0, opt);
        Matrix4f T;
        bool ok;
        std::tie(T, ok) = kinfu::PoseEstimation(opt, Matrix4f::Identity(), frame, model);
        const size_t n0 = model[0]->points_.size(), n1 = model[1]->points_.size();
        const bool has_n = model[0]->HasNormals() && !model[0]->HasColors();
        std::printf("\"kinfu_err\": %.3g, \"kinfu_ok\": %s, \"kinfu_points\": [%zu, %zu], \"kinfu_normals\": %s, ",
                    Fro(T, pose_b), ok ? "true" : "false", n0, n1, has_n ? "true" : "false");
        // CreateFromDepthImage: stride, and the reference's error path for a 3-channel "depth"
        auto strided = geometry::PointCloud::CreateFromDepthImage(fa[0].depth_, k0, Matrix4f::Identity(), 1000.0f, 1000.0f, 4);
        geometry::Image bad;
        bad.Prepare(8, 8, 3, 1);
        auto none = geometry::PointCloud::CreateFromDepthImage(bad, k0);
        const size_t ns = strided->points_.size();
        std::printf("\"depth_strided\": %zu, \"depth_bad_empty\": %s, ", ns, none->IsEmpty() ? "true" : "false");
    }
    {   // RGB-D odometry between two rendered frames (odometry/odometry.cu)
        const camera::PinholeCameraIntrinsic k0(320, 240, 262.5f, 262.5f, 159.5f, 119.5f);
        const Matrix4f pose_b = Rigid(0.02f, 1, 2, 3, 0.02f, -0.02f, 0.01f);
        geometry::RGBDImage target(RenderIntensity(k0, Matrix4f::Identity()), RenderDepth(k0, Matrix4f::Identity()));
        geometry::RGBDImage source(RenderIntensity(k0, pose_b), RenderDepth(k0, pose_b));
        bool ok;
        Matrix4f T;
        Eigen::Matrix6f info;
        std::tie(ok, T, info) = odometry::ComputeRGBDOdometry(source, target, k0, Matrix4f::Identity(),
                                                              odometry::RGBDOdometryJacobianFromHybridTerm(),
                                                              odometry::OdometryOption({20, 10, 5}, 0.03f, 0.0f, 6.0f));
        const float e_h = Fro(T, pose_b);
        std::tie(ok, T, info) = odometry::ComputeRGBDOdometry(source, target, k0, Matrix4f::Identity(),
                                                              odometry::RGBDOdometryJacobianFromColorTerm(),
                                                              odometry::OdometryOption({20, 10, 5}, 0.03f, 0.0f, 6.0f));
        geometry::Image small;
        small.Prepare(8, 8, 1, 4);
        bool bad_ok;
        Matrix4f bad_T;
        std::tie(bad_ok, bad_T, info) = odometry::ComputeRGBDOdometry(geometry::RGBDImage(small, small), target, k0);
        const float motion = Fro(Matrix4f::Identity(), pose_b), e_c = Fro(T, pose_b);
        std::printf("\"odometry_ok\": %s, \"odometry_hybrid_err\": %.3g, \"odometry_color_err\": %.3g, \"odometry_motion\": %.3g, "
                    "\"odometry_mismatch_fails\": %s, ",
                    ok ? "true" : "false", e_h, e_c, motion, (!bad_ok && bad_T.isIdentity()) ? "true" : "false");
    }
    {   // Kabsch golden shape (src/tests/registration/kabsch.cpp:35-55)
        std::vector<Vector3f> pts(20);
        for (auto& p : pts) p = Vector3f(1000 * U(rng), 1000 * U(rng), 1000 * U(rng));
        const Matrix4f ref = Rigid(30.0f / 180.0f * (float)M_PI, 0, 0, 1, 0, 0, 0);
        geometry::PointCloud a(pts), b(pts);
        b.Transform(ref);
        const Matrix4f res = registration::Kabsch(a.points_, b.points_);
        std::printf("\"kabsch_ok\": %s, ", res.isApprox(ref, 1e-3f) ? "true" : "false");
    }
    {   // VoxelDownSample + EstimateNormals on the path's callers side
        auto down = target.VoxelDownSample(0.05f);
        bool unit = true;
        auto dn = down->GetNormals();
        for (auto& v : dn) unit = unit && std::fabs(v.norm() - 1.0f) < 1e-4f;
        std::printf("\"voxels\": %zu, \"voxel_normals_unit\": %s, ", down->points_.size(), unit ? "true" : "false");
        auto empty = target.VoxelDownSample(0.0f);
        std::printf("\"voxel_zero_empty\": %s, ", empty->IsEmpty() ? "true" : "false");
        geometry::PointCloud q(tgt);
        q.EstimateNormals(knn::KDTreeSearchParamKNN(20));
        std::printf("\"has_normals\": %s, ", q.HasNormals() ? "true" : "false");
    }
    {   // GeometryBase3D on a cloud, through the base-class interface (geometry_base.h:44-90)
        geometry::PointCloud p = target;
        geometry::GeometryBase3D& g = p;
        Vector3f mn(1e9f, 1e9f, 1e9f), mx(-1e9f, -1e9f, -1e9f);
        double cs[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::fmin(mn[d], tgt[i][d]);
                mx[d] = std::fmax(mx[d], tgt[i][d]);
                cs[d] += tgt[i][d];
            }
        const Vector3f c0((float)(cs[0] / n), (float)(cs[1] / n), (float)(cs[2] / n));
        const bool bounds_ok = g.GetMinBound() == mn && g.GetMaxBound() == mx && (g.GetCenter() - c0).norm() < 1e-6f;
        const geometry::AxisAlignedBoundingBox<3> box = g.GetAxisAlignedBoundingBox();
        const bool box_ok = box.GetMinBound() == mn && box.GetMaxBound() == mx &&
                            std::fabs(box.Volume() - (mx[0] - mn[0]) * (mx[1] - mn[1]) * (mx[2] - mn[2])) < 1e-6f;
        g.Translate(Vector3f(1.0f, -2.0f, 0.5f));                       // relative
        const bool tr_ok = (g.GetCenter() - (c0 + Vector3f(1.0f, -2.0f, 0.5f))).norm() < 2e-6f;
        g.Translate(Vector3f(0.0f, 0.0f, 0.0f), false);                 // centre moved TO the origin
        const bool tr_abs_ok = g.GetCenter().norm() < 2e-6f;
        g.Scale(2.0f);                                                  // about the centre
        const Vector3f ext = g.GetMaxBound() - g.GetMinBound();
        const bool sc_ok = (ext - 2.0f * (mx - mn)).norm() < 1e-5f && g.GetCenter().norm() < 4e-6f;
        const Matrix4f R4 = Rigid(0.7f, 1, 1, 0, 0, 0, 0);
        Eigen::Matrix3f R;
        for (int r2 = 0; r2 < 3; ++r2) for (int q2 = 0; q2 < 3; ++q2) R(r2, q2) = R4(r2, q2);
        g.Rotate(R, false);
        auto pr = p.GetPoints();
        auto nr = p.GetNormals();
        double worst = 0, worst_n = 0;
        for (int i = 0; i < n; i += 97) {
            const Vector3f want = R * ((tgt[i] - c0) * 2.0f);
            worst = std::fmax(worst, (pr[i] - want).norm());
            worst_n = std::fmax(worst_n, (nr[i] - R * nrm[i]).norm());
        }
        std::printf("\"base3d_bounds\": %s, \"base3d_box\": %s, \"base3d_translate\": %s, \"base3d_translate_abs\": %s, "
                    "\"base3d_scale\": %s, \"base3d_rotate_err\": %.3g, \"base3d_rotate_normals_err\": %.3g, "
                    "\"base3d_empty_center_zero\": %s",
                    bounds_ok ? "true" : "false", box_ok ? "true" : "false", tr_ok ? "true" : "false",
                    tr_abs_ok ? "true" : "false", sc_ok ? "true" : "false", worst, worst_n,
                    geometry::PointCloud().GetCenter().norm() == 0.0f ? "true" : "false");
    }
    {   // io::WritePointCloud / ReadPointCloud (pointcloud_io.h): PCD ascii / binary / binary_compressed, PLY ascii / binary
        geometry::PointCloud pc(std::vector<Vector3f>(tgt.begin(), tgt.begin() + 5000));
        pc.SetNormals(std::vector<Vector3f>(nrm.begin(), nrm.begin() + 5000));
        std::vector<Vector3f> col(5000);
        for (int i = 0; i < 5000; ++i) col[i] = Vector3f((float)(i % 256) / 255.0f, (float)((i / 7) % 256) / 255.0f, (float)((i / 3) % 256) / 255.0f);
        pc.SetColors(col);
        struct Case { const char* file; bool ascii, comp; float tol; } cases[] = {
                {"cpp_bin.pcd", false, false, 0.0f}, {"cpp_ascii.pcd", true, false, 1e-6f}, {"cpp_comp.pcd", false, true, 0.0f},
                {"cpp_bin.ply", false, false, 0.0f}, {"cpp_ascii.ply", true, false, 0.0f}};
        bool all_ok = true;
        std::printf(", \"io\": {");
        for (const auto& cs : cases) {
            const std::string path = io_dir + "/" + cs.file;
            const bool w = io::WritePointCloud(path, pc, cs.ascii, cs.comp);
            geometry::PointCloud back;
            const bool r = io::ReadPointCloud(path, back);
            double worst = 1e9;
            if (w && r && back.points_.size() == 5000 && back.HasNormals() && back.HasColors()) {
                worst = 0;
                auto bp = back.GetPoints(), bn = back.GetNormals(), bc = back.GetColors();
                for (int i = 0; i < 5000; ++i) {
                    worst = std::fmax(worst, (bp[i] - tgt[i]).norm() / std::fmax(1e-9f, tgt[i].norm()));
                    worst = std::fmax(worst, (bn[i] - nrm[i]).norm());
                    worst = std::fmax(worst, (bc[i] - col[i]).norm());
                }
            }
            all_ok = all_ok && worst <= cs.tol + 1e-12;
            std::printf("\"%s\": %.3g, ", cs.file, worst);
        }
        // files written by the Python side (tests/test_gpu_cpp.py) before this program started
        double py_sum = 0;
        int py_n = 0;
        for (const char* f : {"py_comp.pcd", "py_bin.ply"}) {
            auto q = io::CreatePointCloudFromFile(io_dir + "/" + f);
            py_n += (int)q->points_.size() + (q->HasNormals() ? 1 : 0);
            for (const auto& v : q->GetPoints()) py_sum += (double)v[0] + 2.0 * (double)v[1] + 3.0 * (double)v[2];
        }
        // NaN / inf points are dropped on reading (pointcloud_io.cpp:92-95); unknown extension fails
        {
            FILE* f = std::fopen((io_dir + "/nan.pcd").c_str(), "w");
            std::fprintf(f, "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA ascii\n"
                            "1 2 3\nnan 0 0\n4 5 6\ninf 1 1\n");
            std::fclose(f);
        }
        geometry::PointCloud nanpc, raw;
        io::ReadPointCloud(io_dir + "/nan.pcd", nanpc);
        io::ReadPointCloud(io_dir + "/nan.pcd", raw, "auto", false, false);
        geometry::PointCloud none;
        const bool bad = io::ReadPointCloud(io_dir + "/cloud.xyz123", none) || io::WritePointCloud(io_dir + "/cloud.xyz123", pc);
        std::printf("\"all_ok\": %s, \"py_points\": %d, \"py_checksum\": %.17g, \"nan_removed\": %zu, \"nan_kept\": %zu, "
                    "\"unknown_ext_fails\": %s}",
                    all_ok ? "true" : "false", py_n, py_sum, nanpc.points_.size(), raw.points_.size(), bad ? "false" : "true");
    }
    std::printf("}\n");
    return 0;
}
