// C wrapper around the reference's vendored FLANN *CPU* kd-tree
// (third_party/flann/algorithms/kdtree_index.h), compiled from the reference
// tree where it lies (see oracle/Makefile, target _ref).  The ICP path itself
// uses FLANN's CUDA index (kdtree_cuda_3d_index.*), which cannot be built
// here (needs nvcc); the CPU index of the same library is the closest piece of
// the reference that runs, and is used to cross-check the oracle's exact
// kd-tree (tests/test_oracle_ref.py) and, optionally, as the "reference"-kind
// CPU search baseline.  TEST INFRASTRUCTURE ONLY.
#include <flann/algorithms/dist.h>
#include <flann/algorithms/kdtree_index.h>

#include <cmath>
#include <vector>

extern "C" {

// exact k-NN (checks = FLANN_CHECKS_UNLIMITED), squared L2, sorted ascending
int ref_flann_knn(const float* data, int n, const float* query, int nq, int k,
                  int* idx, float* d2) {
    flann::Matrix<float> ds(const_cast<float*>(data), n, 3);
    flann::KDTreeIndex<flann::L2_Simple<float>> index(ds, flann::KDTreeIndexParams(1));
    index.buildIndex();
    flann::Matrix<float> q(const_cast<float*>(query), nq, 3);
    std::vector<size_t> ids((size_t)nq * k);
    flann::Matrix<size_t> mi(ids.data(), nq, k);
    flann::Matrix<float> md(d2, nq, k);
    flann::SearchParams sp(flann::FLANN_CHECKS_UNLIMITED, 0.0f, true);
    int r = index.knnSearch(q, mi, md, k, sp);
    for (size_t i = 0; i < ids.size(); ++i) idx[i] = (int)ids[i];
    return r;
}

// radius search keeping the max_nn nearest within radius^2 (as
// cupoch::knn::KDTreeFlann::SearchRadius sets max_neighbors = max_nn);
// unfilled slots idx = -1, d2 = +inf
int ref_flann_radius(const float* data, int n, const float* query, int nq,
                     float radius, int max_nn, int* idx, float* d2) {
    flann::Matrix<float> ds(const_cast<float*>(data), n, 3);
    flann::KDTreeIndex<flann::L2_Simple<float>> index(ds, flann::KDTreeIndexParams(1));
    index.buildIndex();
    flann::SearchParams sp(flann::FLANN_CHECKS_UNLIMITED, 0.0f, true);
    sp.max_neighbors = max_nn;
    int total = 0;
    for (int i = 0; i < nq; ++i) {
        flann::Matrix<float> q(const_cast<float*>(query) + 3 * (size_t)i, 1, 3);
        std::vector<std::vector<size_t>> vi;
        std::vector<std::vector<float>> vd;
        index.radiusSearch(q, vi, vd, radius * radius, sp);
        int c = (int)vi[0].size();
        if (c > max_nn) c = max_nn;
        for (int s = 0; s < max_nn; ++s) {
            idx[(size_t)i * max_nn + s] = (s < c) ? (int)vi[0][s] : -1;
            d2[(size_t)i * max_nn + s] = (s < c) ? vd[0][s] : INFINITY;
        }
        total += c;
    }
    return total;
}

}  // extern "C"
