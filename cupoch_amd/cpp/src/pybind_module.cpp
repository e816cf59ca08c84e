// cupoch_pybind -- the reference's pybind11 module for the ICP path, built on this repository's C++
// surface (namespace cupoch, libcupoch_amd.so) instead of the CUDA library.
//
// Same module layout, names, defaults and property names as src/python/cupoch_pybind of the reference:
//   cupoch_pybind.utility       Vector3fVector (device vector wrapper with .cpu()), initialize_allocator
//                               (src/python/cupoch_pybind/utility/eigen.cpp:123-200, cupoch_pybind.cpp:46-49)
//   cupoch_pybind.geometry      PointCloud, KDTreeSearchParamKNN / Radius, KDTreeFlann
//                               (geometry/pointcloud.cpp:33-160, geometry/kdtreeflann.cpp)
//   cupoch_pybind.registration  ICPConvergenceCriteria, RegistrationResult, TransformationEstimation* (each behind the
//                               PyTransformationEstimation trampoline: Python-defined estimators, registration.cpp:36-60),
//                               registration_icp, evaluate_registration, registration_generalized_icp,
//                               registration_colored_icp  (registration/registration.cpp:64-448)
//   PointCloud.to_/from_{points,normals,colors}_dlpack  (geometry/pointcloud.cpp:82-100, utility/dl_converter.cu)
// The reference converts Eigen types through pybind11/eigen.h; Eigen is absent here, so 4x4 / 3-vectors
// cross the boundary as float32 numpy arrays converted by hand.
#include <hip/hip_runtime_api.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <type_traits>
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

// utility.Vector2iVector: the CorrespondenceSet as the reference hands it to Python estimators
// (utility/eigen.cpp:340-342): a device vector of (source, target) index pairs with .cpu()
struct Vector2iVector {
    registration::CorrespondenceSet data;
    Vector2iVector() = default;
    explicit Vector2iVector(const py::array_t<int, py::array::c_style | py::array::forcecast>& a) {
        if (a.ndim() != 2 || a.shape(1) != 2) throw std::invalid_argument("expected an (n, 2) int32 array");
        std::vector<Eigen::Vector2i> h((size_t)a.shape(0));
        if (!h.empty()) std::memcpy((void*)h.data(), a.data(), h.size() * sizeof(Eigen::Vector2i));
        data = h;
    }
};

py::array_t<int> corres_to_array(const registration::CorrespondenceSet& c) {
    const std::vector<Eigen::Vector2i> h = c.to_host();
    py::array_t<int> a({(py::ssize_t)h.size(), (py::ssize_t)2});
    if (!h.empty()) std::memcpy(a.mutable_data(), (const void*)h.data(), h.size() * sizeof(Eigen::Vector2i));
    return a;
}

// ---------------------------------------------------------------------------------------------------------
// Python-defined estimators (registration/registration.cpp:36-60 of the reference: PyTransformationEstimation with
// PYBIND11_OVERLOAD_PURE).  A Python subclass overrides the C++ virtuals under their C++ NAMES, as that macro has
// it -- GetTransformationEstimationType / ComputeRMSE / ComputeTransformation -- or under the snake_case names the
// methods are bound with; RegistrationICP then runs the reference's host loop (registration.cu:144-171) and makes
// the virtual calls.  pybind11/eigen.h is not available (no Eigen), so the 4x4 comes back as a numpy array and is
// converted by hand.  A built-in estimator that Python did not subclass is constructed as its plain C++ type (pybind11
// makes the alias only for Python-derived instances), so it still takes the device-resident loop.
template <class Base = registration::TransformationEstimation>
class PyTransformationEstimation : public Base {
public:
    using Base::Base;

    registration::TransformationEstimationType GetTransformationEstimationType() const override {
        py::gil_scoped_acquire gil;
        const py::function f = find("GetTransformationEstimationType", "get_transformation_estimation_type");
        if (f) return f().template cast<registration::TransformationEstimationType>();
        if constexpr (std::is_abstract<Base>::value) pure("GetTransformationEstimationType");
        else return Base::GetTransformationEstimationType();
    }
    float ComputeRMSE(const geometry::PointCloud& source, const geometry::PointCloud& target,
                      const registration::CorrespondenceSet& corres) const override {
        py::gil_scoped_acquire gil;
        const py::function f = find("ComputeRMSE", "compute_rmse");
        if (f) return call(f, source, target, corres).template cast<float>();
        if constexpr (std::is_abstract<Base>::value) pure("ComputeRMSE");
        else return Base::ComputeRMSE(source, target, corres);
    }
    Eigen::Matrix4f ComputeTransformation(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                          const registration::CorrespondenceSet& corres) const override {
        py::gil_scoped_acquire gil;
        const py::function f = find("ComputeTransformation", "compute_transformation");
        if (f) return to_matrix4(call(f, source, target, corres).template cast<farray>());
        if constexpr (std::is_abstract<Base>::value) pure("ComputeTransformation");
        else return Base::ComputeTransformation(source, target, corres);
    }

private:
    py::function find(const char* cpp_name, const char* py_name) const {
        py::function f = py::get_override(static_cast<const Base*>(this), cpp_name);
        if (!f) f = py::get_override(static_cast<const Base*>(this), py_name);
        return f;
    }
    // the clouds by reference (the estimator sees the loop's own objects, as in the reference), the set as a Vector2iVector
    static py::object call(const py::function& f, const geometry::PointCloud& source, const geometry::PointCloud& target,
                           const registration::CorrespondenceSet& corres) {
        Vector2iVector v;
        v.data = corres;
        return f(py::cast(source, py::return_value_policy::reference), py::cast(target, py::return_value_policy::reference),
                 py::cast(std::move(v)));
    }
    [[noreturn]] static void pure(const char* name) {
        py::pybind11_fail(std::string("Tried to call pure virtual function \"TransformationEstimation::") + name + "\"");
    }
};

// ---------------------------------------------------------------------------------------------------------
// DLPack (geometry/pointcloud.cpp:82-100, utility/dl_converter.cu:42-118, cupoch_pybind/dl_converter.inl of the
// reference; its third_party/dlpack is an empty submodule here, so the ABI structs of DLPack's stable v0.x
// `DLManagedTensor` -- what a capsule named "dltensor" carries -- are restated).  Export: an (n, 3) float32 tensor on
// kDLROCM over the cloud's OWN buffer -- zero-copy, as the reference publishes it (dl_converter.cu:60-86) -- kept alive
// by a shared handle on the block (utility::device_vector::share): the tensor stays valid when the cloud is destroyed
// or re-allocated, sees what the cloud's in-place operations (Transform, ...) write until then, and costs nothing
// (the reference's handle is a thrust copy of the vector, 120 MB for a 10M-point cloud, that nothing reads).  Import:
// host (kDLCPU / pinned) or device (kDLROCM, or kDLCUDA as PyTorch-ROCm builds before 2.x labelled it) memory is COPIED
// into the cloud's vector; the capsule is left to its owner, as in the reference.
extern "C" {
typedef struct { int32_t device_type; int32_t device_id; } MiDLDevice;
typedef struct { uint8_t code; uint8_t bits; uint16_t lanes; } MiDLDataType;
typedef struct {
    void* data;
    MiDLDevice device;
    int32_t ndim;
    MiDLDataType dtype;
    int64_t* shape;
    int64_t* strides;
    uint64_t byte_offset;
} MiDLTensor;
typedef struct MiDLManagedTensor {
    MiDLTensor dl_tensor;
    void* manager_ctx;
    void (*deleter)(struct MiDLManagedTensor*);
} MiDLManagedTensor;
}
enum { kMiDLCPU = 1, kMiDLCUDA = 2, kMiDLCUDAHost = 3, kMiDLROCM = 10, kMiDLROCMHost = 11 };
enum { kMiDLFloat = 2 };

struct Vec3Export {
    std::shared_ptr<void> handle;  // the cloud's block, shared
    int64_t shape[2];
    MiDLManagedTensor tensor;
};

py::capsule to_dlpack_capsule(const utility::device_vector<Eigen::Vector3f>& src) {
    Vec3Export* e = new Vec3Export();
    e->handle = src.share();  // no copy: the tensor and the cloud hold the same block
    int dev = 0;
    (void)hipGetDevice(&dev);
    // (the legacy capsule has no stream handshake: everything enqueued that writes the block has finished before a
    // consumer on any stream sees the pointer -- ADVICE r5; INTEGRATION.md section 2 lists what aliasing means otherwise)
    (void)hipDeviceSynchronize();
    e->shape[0] = (int64_t)src.size();
    e->shape[1] = 3;
    MiDLTensor& t = e->tensor.dl_tensor;
    t.data = (void*)src.data();
    t.device.device_type = kMiDLROCM;
    t.device.device_id = dev;
    t.ndim = 2;
    t.dtype.code = kMiDLFloat;
    t.dtype.bits = 32;
    t.dtype.lanes = 1;
    t.shape = e->shape;
    t.strides = nullptr;  // compact row-major
    t.byte_offset = 0;
    e->tensor.manager_ctx = e;
    e->tensor.deleter = [](MiDLManagedTensor* m) { delete static_cast<Vec3Export*>(m->manager_ctx); };
    return py::capsule(&e->tensor, "dltensor", [](PyObject* obj) {
        // a consumer renames the capsule ("used_dltensor") and owns the tensor from then on
        void* ptr = PyCapsule_IsValid(obj, "dltensor") ? PyCapsule_GetPointer(obj, "dltensor") : nullptr;
        if (ptr) {
            MiDLManagedTensor* m = static_cast<MiDLManagedTensor*>(ptr);
            if (m->deleter) m->deleter(m);
        } else {
            PyErr_Clear();
        }
    });
}

void from_dlpack_capsule(const py::capsule& cap, utility::device_vector<Eigen::Vector3f>& dst) {
    if (!PyCapsule_IsValid(cap.ptr(), "dltensor")) {
        PyErr_Clear();
        throw std::invalid_argument("expected a DLPack capsule named \"dltensor\" (not yet consumed)");
    }
    const MiDLManagedTensor* m = static_cast<const MiDLManagedTensor*>(PyCapsule_GetPointer(cap.ptr(), "dltensor"));
    const MiDLTensor& t = m->dl_tensor;
    if (t.ndim != 2 || t.shape[1] != 3 || t.dtype.code != kMiDLFloat || t.dtype.bits != 32 || t.dtype.lanes != 1)
        throw std::invalid_argument("from_*_dlpack: expected an (n, 3) float32 tensor");
    if (t.strides && t.shape[0] > 0 && (t.strides[1] != 1 || (t.shape[0] > 1 && t.strides[0] != 3)))
        throw std::invalid_argument("from_*_dlpack: the tensor must be contiguous");
    const size_t n = (size_t)t.shape[0];
    const char* src = static_cast<const char*>(t.data) + t.byte_offset;
    const int kind = t.device.device_type;
    dst.resize(0);
    dst.resize(n);
    if (n == 0) return;
    const size_t bytes = n * sizeof(Eigen::Vector3f);
    hipError_t e;
    if (kind == kMiDLCPU || kind == kMiDLCUDAHost || kind == kMiDLROCMHost) e = hipMemcpy(dst.data(), src, bytes, hipMemcpyHostToDevice);
    else if (kind == kMiDLROCM || kind == kMiDLCUDA) e = hipMemcpy(dst.data(), src, bytes, hipMemcpyDeviceToDevice);
    else throw std::invalid_argument("from_*_dlpack: unsupported device type");  // utility::LogError in the reference (dl_converter.cu:116)
    if (e != hipSuccess) throw std::runtime_error(std::string("from_*_dlpack: ") + hipGetErrorString(e));
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
    py::class_<Vector2iVector>(mu, "Vector2iVector")
            .def(py::init<>())
            .def(py::init<const py::array_t<int, py::array::c_style | py::array::forcecast>&>(), "array"_a)
            .def("cpu", [](const Vector2iVector& v) { return corres_to_array(v.data); })
            .def("__len__", [](const Vector2iVector& v) { return v.data.size(); });

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
            .def("to_points_dlpack", [](geometry::PointCloud& pc) { return to_dlpack_capsule(pc.points_); })
            .def("to_normals_dlpack", [](geometry::PointCloud& pc) { return to_dlpack_capsule(pc.normals_); })
            .def("to_colors_dlpack", [](geometry::PointCloud& pc) { return to_dlpack_capsule(pc.colors_); })
            .def("from_points_dlpack", [](geometry::PointCloud& pc, py::capsule c) { from_dlpack_capsule(c, pc.points_); })
            .def("from_normals_dlpack", [](geometry::PointCloud& pc, py::capsule c) { from_dlpack_capsule(c, pc.normals_); })
            .def("from_colors_dlpack", [](geometry::PointCloud& pc, py::capsule c) { from_dlpack_capsule(c, pc.colors_); })
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

    // the estimators, each behind the trampoline (a Python class may derive from any of them)
    auto set_arg = [](const py::object& o) {  // a Vector2iVector or anything (n, 2)-array-like
        if (py::isinstance<Vector2iVector>(o)) return o.cast<const Vector2iVector&>().data;
        return Vector2iVector(o.cast<py::array_t<int, py::array::c_style | py::array::forcecast>>()).data;
    };
    py::class_<registration::TransformationEstimation, PyTransformationEstimation<>>(mr, "TransformationEstimation")
            .def(py::init<>())
            .def("get_transformation_estimation_type",
                 &registration::TransformationEstimation::GetTransformationEstimationType)
            .def("compute_rmse",
                 [set_arg](const registration::TransformationEstimation& e, const geometry::PointCloud& s,
                           const geometry::PointCloud& t, const py::object& corres) { return e.ComputeRMSE(s, t, set_arg(corres)); },
                 "source"_a, "target"_a, "corres"_a)
            .def("compute_transformation",
                 [set_arg](const registration::TransformationEstimation& e, const geometry::PointCloud& s,
                           const geometry::PointCloud& t, const py::object& corres) {
                     return from_matrix4(e.ComputeTransformation(s, t, set_arg(corres)));
                 },
                 "source"_a, "target"_a, "corres"_a);
    py::class_<registration::TransformationEstimationPointToPoint,
               PyTransformationEstimation<registration::TransformationEstimationPointToPoint>, registration::TransformationEstimation>(
            mr, "TransformationEstimationPointToPoint")
            .def(py::init<>());
    py::class_<registration::TransformationEstimationPointToPlane,
               PyTransformationEstimation<registration::TransformationEstimationPointToPlane>, registration::TransformationEstimation>(
            mr, "TransformationEstimationPointToPlane")
            .def(py::init<float>(), "det_thresh"_a = 1e-6f)
            .def_readwrite("det_thresh", &registration::TransformationEstimationPointToPlane::det_thresh_);
    py::class_<registration::TransformationEstimationSymmetricMethod,
               PyTransformationEstimation<registration::TransformationEstimationSymmetricMethod>, registration::TransformationEstimation>(
            mr, "TransformationEstimationSymmetricMethod")
            .def(py::init<float>(), "det_thresh"_a = 1e-6f)
            .def_readwrite("det_thresh", &registration::TransformationEstimationSymmetricMethod::det_thresh_);
    py::class_<registration::TransformationEstimationForGeneralizedICP,
               PyTransformationEstimation<registration::TransformationEstimationForGeneralizedICP>, registration::TransformationEstimation>(
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
