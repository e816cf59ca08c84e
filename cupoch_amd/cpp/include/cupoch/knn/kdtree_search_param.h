// cupoch/knn/kdtree_search_param.h (reference: knn/kdtree_search_param.h:20-80)
#pragma once
namespace cupoch {
namespace knn {

static const int NUM_MAX_NN = 100;

class KDTreeSearchParam {
public:
    enum class SearchType { Knn = 0, Radius = 1 };
    virtual ~KDTreeSearchParam() {}
    SearchType GetSearchType() const { return search_type_; }

protected:
    KDTreeSearchParam(SearchType type) : search_type_(type) {}

private:
    SearchType search_type_;
};

class KDTreeSearchParamKNN : public KDTreeSearchParam {
public:
    KDTreeSearchParamKNN(int knn = 30) : KDTreeSearchParam(SearchType::Knn), knn_(knn) {}
    int knn_;
};

class KDTreeSearchParamRadius : public KDTreeSearchParam {
public:
    KDTreeSearchParamRadius(float radius, int max_nn)
        : KDTreeSearchParam(SearchType::Radius), radius_(radius), max_nn_(max_nn) {}
    float radius_;
    int max_nn_;
};

}  // namespace knn
}  // namespace cupoch
