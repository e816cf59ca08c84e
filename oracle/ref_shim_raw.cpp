// C wrapper around the reference's own deterministic test-data generator
// unit_test::Raw (src/tests/test_utility/raw.{h,cpp}), compiled from the
// reference tree where it lies (see oracle/Makefile, target _ref).
// TEST INFRASTRUCTURE ONLY -- used by oracle/make_golden.py to regenerate the
// inputs of the reference's golden-vector tests.  No reference source is
// copied into this repository; this file only calls its public class.
#include "tests/test_utility/raw.h"

extern "C" {

// unit_test::Rand(host_vector<Vector3f>&, vmin, vmax, seed)
// (src/tests/test_utility/rand.cpp:115-131; that file needs Eigen/thrust and
// cannot be compiled here, so its three-line loop body is restated on top of
// the reference's Raw::Next<float>()).
void ref_rand_vec3f(float* out, int n, const float* vmin, const float* vmax, int seed) {
    unit_test::Raw raw(seed);
    float factor[3];
    for (int d = 0; d < 3; ++d) factor[d] = vmax[d] - vmin[d];
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d)
            out[3 * i + d] = vmin[d] + raw.Next<float>() * factor[d];
}

// unit_test::Rand(float* const, size, vmin, vmax, seed) (rand.cpp:241-252)
void ref_rand_floats(float* out, int n, float vmin, float vmax, int seed) {
    unit_test::Raw raw(seed);
    const float factor = vmax - vmin;
    for (int i = 0; i < n; ++i) out[i] = vmin + raw.Next<float>() * factor;
}

// unit_test::Rand(host_vector<Vector4i>&, int vmin, int vmax, seed) (rand.cpp:137-151)
void ref_rand_vec4i(int* out, int n, int vmin, int vmax, int seed) {
    unit_test::Raw raw(seed);
    const double factor = (double)(vmax - vmin) / unit_test::Raw::VMAX;
    for (int i = 0; i < 4 * n; ++i) out[i] = vmin + (int)(raw.Next<int>() * factor);
}

}  // extern "C"
