// cupoch/knn/kdtree_flann.h (reference: knn/kdtree_flann.h:43-124, .inl:46-144, .cu:34-79)
// Same class and method names; the index behind it is the engine's kd tree
// (mi_icp_set_target / mi_icp_search_knn), one engine context per KDTreeFlann.
// Up to knn::NUM_MAX_NN = 100 neighbours per query, as the reference.
#pragma once
#include <memory>

#include "cupoch/knn/kdtree_search_param.h"
#include "cupoch/utility/device_vector.h"
#include "cupoch/utility/eigen.h"

struct mi_icp_ctx;

namespace cupoch {
namespace knn {

class KDTreeFlann {
public:
    KDTreeFlann();
    explicit KDTreeFlann(const utility::device_vector<Eigen::Vector3f>& data);
    ~KDTreeFlann();
    KDTreeFlann(const KDTreeFlann&) = delete;
    KDTreeFlann& operator=(const KDTreeFlann&) = delete;

    bool SetRawData(const utility::device_vector<Eigen::Vector3f>& data);

    // many queries: indices / distance2 are [query.size()][knn] row-major, -1 / +inf padded;
    // returns the number of neighbours found, -1 on empty data or query (kdtree_flann.cu:52-54)
    int Search(const utility::device_vector<Eigen::Vector3f>& query, const KDTreeSearchParam& param,
               utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const;
    int SearchKNN(const utility::device_vector<Eigen::Vector3f>& query, int knn,
                  utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const;
    int SearchRadius(const utility::device_vector<Eigen::Vector3f>& query, float radius, int max_nn,
                     utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const;

    // one query, host results (kdtree_flann.cu:34-79)
    int Search(const Eigen::Vector3f& query, const KDTreeSearchParam& param, thrust::host_vector<int>& indices,
               thrust::host_vector<float>& distance2) const;
    int SearchKNN(const Eigen::Vector3f& query, int knn, thrust::host_vector<int>& indices,
                  thrust::host_vector<float>& distance2) const;
    template <typename T = Eigen::Vector3f>
    int SearchRadius(const T& query, float radius, int max_nn, thrust::host_vector<int>& indices,
                     thrust::host_vector<float>& distance2) const {
        return SearchRadiusOne(query, radius, max_nn, indices, distance2);
    }

private:
    int SearchMany(const utility::device_vector<Eigen::Vector3f>& query, int knn, float radius,
                   utility::device_vector<int>& indices, utility::device_vector<float>& distance2) const;
    int SearchRadiusOne(const Eigen::Vector3f& query, float radius, int max_nn, thrust::host_vector<int>& indices,
                        thrust::host_vector<float>& distance2) const;
    mi_icp_ctx* ctx_ = nullptr;
    size_t dataset_size_ = 0;
};

}  // namespace knn
}  // namespace cupoch
