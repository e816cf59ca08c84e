// cupoch_pybind -- the reference's pybind11 module for the ICP path, built on this repository's C++
// surface (namespace cupoch, libcupoch_amd.so) instead of the CUDA library.
//
// Same module layout, names, defaults and property names as src/python/cupoch_pybind of the reference:
//   cupoch_pybind.utility       Vector3fVector (device vector wrapper with .cpu()), initialize_allocator
//                               (src/python/cupoch_pybind/utility/eigen.cpp:123-200, cupoch_pybind.cpp:46-49)
//   cupoch_pybind.geometry      PointCloud, KDTreeSearchParamKNN / Radius, KDTreeFlann
//                               (geometry/pointcloud.cpp:33-160, geometry/kdtreeflann.cpp)
//   cupoch_pybind.registration  ICPConvergenceCriteria, RegistrationResult, TransformationEstimation*,
//                               registration_icp, evaluate_registration, registration_generalized_icp,
//                               registration_colored_icp  (registration/registration.cpp:64-448)
// The reference converts Eigen types through pybind11/eigen.h; Eigen is absent here, so 4x4 / 3-vectors
// cross the boundary as float32 numpy arrays converted by hand.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "cupoch/geometry/pointcloud.h"
#include "cupoch/knn/kdtree_flann.h"
#include "cupoch/knn/kdtree_search_param.h"
#include "cupoch/registration/generalized_icp.h"
#include "cupoch/registration/registration.h"
#include "cupoch/registration/transformation_estimation.h"

namespace py = pybind11;
using namespace py::literals;
using namespace cupoch;

namespace {

typedef py::array_t<float, py::array::c_style | py::array::forcecast> farray;

Eigen::Matrix4f to_matrix4(const farray& a) {
    if (a.ndim() != 2 || a.shape(0) != 4 || a.shape(1) != 4) throw std::invalid_argument("expected a 4x4 float32 array");
    Eigen::Matrix4f m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = a.at(r, c);
    return m;
}

farray from_matrix4(const Eigen::Matrix4f& m) {
    farray a({4, 4});
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) a.mutable_at(r, c) = m(r, c);
    return a;
}

farray identity4() { return from_matrix4(Eigen::Matrix4f::Identity()); }

Eigen::Vector3f to_vector3(const farray& a) {
    if (a.size() != 3) throw std::invalid_argument("expected 3 floats");
    Eigen::Vector3f v;
    const float* p = a.data();
    v(0) = p[0];
    v(1) = p[1];
    v(2) = p[2];
    return v;
}

farray from_vector3(const Eigen::Vector3f& v) {
    farray a(3);
    for (int i = 0; i < 3; ++i) a.mutable_at(i) = v(i);
    return a;
}

// utility.Vector3fVector: a device vector of Vector3f, constructed from an (n, 3) array (one H2D
// copy), .cpu() copies back (device_vector_wrapper.cu:39-44,119-125)
struct Vector3fVector {
    utility::device_vector<Eigen::Vector3f> data;
    Vector3fVector() = default;
    explicit Vector3fVector(const farray& a) { assign(a); }
    void assign(const farray& a) {
        if (a.ndim() != 2 || a.shape(1) != 3) throw std::invalid_argument("expected an (n, 3) float32 array");
        std::vector<Eigen::Vector3f> h((size_t)a.shape(0));
        if (!h.empty()) std::memcpy((void*)h.data(), a.data(), h.size() * sizeof(Eigen::Vector3f));
        data = h;
    }
    farray cpu() const {
        const std::vector<Eigen::Vector3f> h = data.to_host();
        farray a({(py::ssize_t)h.size(), (py::ssize_t)3});
        if (!h.empty()) std::memcpy(a.mutable_data(), (const void*)h.data(), h.size() * sizeof(Eigen::Vector3f));
        return a;
    }
};

Vector3fVector wrap(const utility::device_vector<Eigen::Vector3f>& v) {
    Vector3fVector w;
    w.data = v;
    return w;
}

// a property that accepts a Vector3fVector or anything array-like
void set_vec(utility::device_vector<Eigen::Vector3f>& dst, const py::object& o) {
    if (py::isinstance<Vector3fVector>(o)) dst = o.cast<const Vector3fVector&>().data;
    else dst = Vector3fVector(o.cast<farray>()).data;
}

py::array_t<int> corres_to_array(const registration::CorrespondenceSet& c) {
    const std::vector<Eigen::Vector2i> h = c.to_host();
    py::array_t<int> a({(py::ssize_t)h.size(), (py::ssize_t)2});
    if (!h.empty()) std::memcpy(a.mutable_data(), (const void*)h.data(), h.size() * sizeof(Eigen::Vector2i));
    return a;
}

}  // namespace

PYBIND11_MODULE(cupoch_pybind, m) {
    m.doc() = "cupoch's ICP registration path on MI355X: the reference's pybind11 surface over libcupoch_amd / libmi_icp";
    // cupoch/__init__.py calls this on import (rmm pool set-up in the reference): nothing to do here
    m.def("initialize_allocator", [](py::args, py::kwargs) {}, "kept for compatibility; the engine sizes its buffers per cloud");

    // ---------------------------------------------------------------- utility
    py::module mu = m.def_submodule("utility");
    py::class_<Vector3fVector>(mu, "Vector3fVector")
            .def(py::init<>())
            .def(py::init<const farray&>(), "array"_a)
            .def("cpu", &Vector3fVector::cpu)
            .def("__len__", [](const Vector3fVector& v) { return v.data.size(); });

    // ---------------------------------------------------------------- geometry
    py::module mg = m.def_submodule("geometry");
    py::class_<knn::KDTreeSearchParam>(mg, "KDTreeSearchParam");
    py::class_<knn::KDTreeSearchParamKNN, knn::KDTreeSearchParam>(mg, "KDTreeSearchParamKNN")
            .def(py::init<int>(), "knn"_a = 30)
            .def_readwrite("knn", &knn::KDTreeSearchParamKNN::knn_);
    py::class_<knn::KDTreeSearchParamRadius, knn::KDTreeSearchParam>(mg, "KDTreeSearchParamRadius")
            .def(py::init<float, int>(), "radius"_a, "max_nn"_a)
            .def_readwrite("radius", &knn::KDTreeSearchParamRadius::radius_)
            .def_readwrite("max_nn", &knn::KDTreeSearchParamRadius::max_nn_);

    py::class_<geometry::PointCloud, std::shared_ptr<geometry::PointCloud>>(mg, "PointCloud")
            .def(py::init<>())
            .def(py::init([](const py::object& pts) {
                     auto pc = std::make_shared<geometry::PointCloud>();
                     set_vec(pc->points_, pts);
                     return pc;
                 }),
                 "points"_a)
            .def_property(
                    "points", [](const geometry::PointCloud& pc) { return wrap(pc.points_); },
                    [](geometry::PointCloud& pc, const py::object& o) { set_vec(pc.points_, o); })
            .def_property(
                    "normals", [](const geometry::PointCloud& pc) { return wrap(pc.normals_); },
                    [](geometry::PointCloud& pc, const py::object& o) { set_vec(pc.normals_, o); })
            .def_property(
                    "colors", [](const geometry::PointCloud& pc) { return wrap(pc.colors_); },
                    [](geometry::PointCloud& pc, const py::object& o) { set_vec(pc.colors_, o); })
            .def("has_points", &geometry::PointCloud::HasPoints)
            .def("has_normals", &geometry::PointCloud::HasNormals)
            .def("has_colors", &geometry::PointCloud::HasColors)
            .def("has_covariances", &geometry::PointCloud::HasCovariances)
            .def("is_empty", &geometry::PointCloud::IsEmpty)
            .def("clear", [](geometry::PointCloud& pc) { pc.Clear(); })
            .def("get_min_bound", [](const geometry::PointCloud& pc) { return from_vector3(pc.GetMinBound()); })
            .def("get_max_bound", [](const geometry::PointCloud& pc) { return from_vector3(pc.GetMaxBound()); })
            .def("get_center", [](const geometry::PointCloud& pc) { return from_vector3(pc.GetCenter()); })
            .def("transform",
                 [](std::shared_ptr<geometry::PointCloud> pc, const farray& T) {
                     pc->Transform(to_matrix4(T));
                     return pc;
                 },
                 "transformation"_a)
            .def("translate",
                 [](std::shared_ptr<geometry::PointCloud> pc, const farray& t, bool relative) {
                     pc->Translate(to_vector3(t), relative);
                     return pc;
                 },
                 "translation"_a, "relative"_a = true)
            .def("scale",
                 [](std::shared_ptr<geometry::PointCloud> pc, float s, bool center) {
                     pc->Scale(s, center);
                     return pc;
                 },
                 "scale"_a, "center"_a = true)
            .def("voxel_down_sample", &geometry::PointCloud::VoxelDownSample, "voxel_size"_a)
            .def("estimate_normals", &geometry::PointCloud::EstimateNormals,
                 "search_param"_a = knn::KDTreeSearchParamKNN())
            .def("__len__", [](const geometry::PointCloud& pc) { return pc.points_.size(); });

    // ---------------------------------------------------------------- registration
    py::module mr = m.def_submodule("registration");
    py::class_<registration::ICPConvergenceCriteria>(mr, "ICPConvergenceCriteria")
            .def(py::init<float, float, int>(), "relative_fitness"_a = 1e-6f, "relative_rmse"_a = 1e-6f,
                 "max_iteration"_a = 30)
            .def_readwrite("relative_fitness", &registration::ICPConvergenceCriteria::relative_fitness_)
            .def_readwrite("relative_rmse", &registration::ICPConvergenceCriteria::relative_rmse_)
            .def_readwrite("max_iteration", &registration::ICPConvergenceCriteria::max_iteration_);

    py::enum_<registration::TransformationEstimationType>(mr, "TransformationEstimationType")
            .value("Unspecified", registration::TransformationEstimationType::Unspecified)
            .value("PointToPoint", registration::TransformationEstimationType::PointToPoint)
            .value("PointToPlane", registration::TransformationEstimationType::PointToPlane)
            .value("SymmetricMethod", registration::TransformationEstimationType::SymmetricMethod)
            .value("ColoredICP", registration::TransformationEstimationType::ColoredICP)
            .value("GeneralizedICP", registration::TransformationEstimationType::GeneralizedICP);

    py::class_<registration::TransformationEstimation>(mr, "TransformationEstimation")
            .def("get_transformation_estimation_type",
                 &registration::TransformationEstimation::GetTransformationEstimationType);
    py::class_<registration::TransformationEstimationPointToPoint, registration::TransformationEstimation>(
            mr, "TransformationEstimationPointToPoint")
            .def(py::init<>());
    py::class_<registration::TransformationEstimationPointToPlane, registration::TransformationEstimation>(
            mr, "TransformationEstimationPointToPlane")
            .def(py::init<float>(), "det_thresh"_a = 1e-6f)
            .def_readwrite("det_thresh", &registration::TransformationEstimationPointToPlane::det_thresh_);
    py::class_<registration::TransformationEstimationSymmetricMethod, registration::TransformationEstimation>(
            mr, "TransformationEstimationSymmetricMethod")
            .def(py::init<float>(), "det_thresh"_a = 1e-6f)
            .def_readwrite("det_thresh", &registration::TransformationEstimationSymmetricMethod::det_thresh_);
    py::class_<registration::TransformationEstimationForGeneralizedICP, registration::TransformationEstimation>(
            mr, "TransformationEstimationForGeneralizedICP")
            .def(py::init<float>(), "epsilon"_a = 1e-3f)
            .def_readwrite("epsilon", &registration::TransformationEstimationForGeneralizedICP::epsilon_);

    py::class_<registration::RegistrationResult>(mr, "RegistrationResult")
            .def(py::init<>())
            .def_property(
                    "transformation",
                    [](const registration::RegistrationResult& r) { return from_matrix4(r.transformation_); },
                    [](registration::RegistrationResult& r, const farray& T) { r.transformation_ = to_matrix4(T); })
            .def_property_readonly("correspondence_set",
                                   [](const registration::RegistrationResult& r) { return corres_to_array(r.correspondence_set_); })
            .def_readwrite("inlier_rmse", &registration::RegistrationResult::inlier_rmse_)
            .def_readwrite("fitness", &registration::RegistrationResult::fitness_)
            .def("__repr__", [](const registration::RegistrationResult& r) {
                return "registration::RegistrationResult with fitness = " + std::to_string(r.fitness_) +
                       ", inlier_rmse = " + std::to_string(r.inlier_rmse_) + ", and correspondence_set size of " +
                       std::to_string(r.correspondence_set_.size());
            });

    mr.def(
            "evaluate_registration",
            [](const geometry::PointCloud& s, const geometry::PointCloud& t, float d, const farray& T) {
                return registration::EvaluateRegistration(s, t, d, to_matrix4(T));
            },
            "source"_a, "target"_a, "max_correspondence_distance"_a, "transformation"_a = identity4());
    mr.def(
            "registration_icp",
            [](const geometry::PointCloud& s, const geometry::PointCloud& t, float d, const farray& init,
               const registration::TransformationEstimation& est, const registration::ICPConvergenceCriteria& crit) {
                return registration::RegistrationICP(s, t, d, to_matrix4(init), est, crit);
            },
            "source"_a, "target"_a, "max_correspondence_distance"_a, "init"_a = identity4(),
            "estimation_method"_a = registration::TransformationEstimationPointToPoint(),
            "criteria"_a = registration::ICPConvergenceCriteria());
    mr.def(
            "registration_generalized_icp",
            [](const geometry::PointCloud& s, const geometry::PointCloud& t, float d, const farray& init,
               const registration::TransformationEstimationForGeneralizedICP& est,
               const registration::ICPConvergenceCriteria& crit) {
                return registration::RegistrationGeneralizedICP(s, t, d, to_matrix4(init), est, crit);
            },
            "source"_a, "target"_a, "max_correspondence_distance"_a, "init"_a = identity4(),
            "estimation"_a = registration::TransformationEstimationForGeneralizedICP(),
            "criteria"_a = registration::ICPConvergenceCriteria());
    mr.def(
            "registration_colored_icp",
            [](const geometry::PointCloud& s, const geometry::PointCloud& t, float d, const farray& init,
               const registration::ICPConvergenceCriteria& crit, float lambda_geometric, float det_thresh) {
                return registration::RegistrationColoredICP(s, t, d, to_matrix4(init), crit, lambda_geometric, det_thresh);
            },
            "source"_a, "target"_a, "max_distance"_a, "init"_a = identity4(),
            "criteria"_a = registration::ICPConvergenceCriteria(), "lambda_geometric"_a = 0.968f, "det_thresh"_a = 1e-6f);
}
