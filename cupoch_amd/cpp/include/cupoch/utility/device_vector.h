// cupoch/utility/device_vector.h -- utility::device_vector<T> of the reference
// (rmm::device_vector, utility/device_vector.h:74-106) reduced to what the ICP
// surface needs: a typed, owning HIP device buffer that converts from / to a
// host vector.  thrust::host_vector is std::vector here.
#pragma once
#include <cstddef>
#include <memory>
#include <utility>
#include <vector>

namespace thrust {
template <typename T>
using host_vector = std::vector<T>;
}

namespace cupoch {
namespace utility {

void* device_alloc(size_t bytes);
void device_free(void* p);
void copy_h2d(void* dst, const void* src, size_t bytes);
void copy_d2h(void* dst, const void* src, size_t bytes);
void copy_d2d(void* dst, const void* src, size_t bytes);

/// utility::InitializeAllocator (utility/device_vector.cu:28-69): the engine
/// keeps one arena per context; kept for source compatibility.
enum rmmAllocationMode_t { CudaDefaultAllocation = 0, PoolAllocation = 1, CudaManagedMemory = 2 };
inline void InitializeAllocator(rmmAllocationMode_t = CudaDefaultAllocation, size_t = 0,
                                const std::vector<int>& = {}) {}

template <typename T>
class device_vector {
public:
    device_vector() = default;
    explicit device_vector(size_t n) { resize(n); }
    device_vector(const device_vector& o) { assign_device(o.data_, o.size_); }
    device_vector(device_vector&& o) noexcept { swap(o); }
    device_vector(const std::vector<T>& h) { assign_host(h.data(), h.size()); }
    device_vector& operator=(const device_vector& o) {
        if (this != &o) assign_device(o.data_, o.size_);
        return *this;
    }
    device_vector& operator=(device_vector&& o) noexcept {
        swap(o);
        return *this;
    }
    device_vector& operator=(const std::vector<T>& h) {
        assign_host(h.data(), h.size());
        return *this;
    }
    operator std::vector<T>() const { return to_host(); }
    std::vector<T> to_host() const {
        std::vector<T> h(size_);
        if (size_) copy_d2h(h.data(), data_, size_ * sizeof(T));
        return h;
    }
    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    T* data() { return data_; }
    const T* data() const { return data_; }
    void clear() { size_ = 0; }
    void resize(size_t n) {
        if (n > cap_) {
            T* p = (T*)device_alloc(n * sizeof(T));
            if (size_) copy_d2d(p, data_, size_ * sizeof(T));
            store_.reset(p, [](void* q) { device_free(q); });  // (the former block goes when its last holder does)
            data_ = p;
            cap_ = n;
        }
        size_ = n;
    }
    /// The block behind data(), shared: whoever holds it keeps the memory alive -- the zero-copy DLPack export
    /// (the reference's handle, utility/dl_converter.cu:60-69).  The vector itself never shares on copy (copies are
    /// deep, as thrust's); a later resize beyond the capacity moves the VECTOR to a new block and leaves this one
    /// to its holders.
    std::shared_ptr<void> share() const { return store_; }
    void swap(device_vector& o) noexcept {
        std::swap(store_, o.store_);
        std::swap(data_, o.data_);
        std::swap(size_, o.size_);
        std::swap(cap_, o.cap_);
    }

private:
    void assign_host(const T* h, size_t n) {
        resize(0);
        resize(n);
        if (n) copy_h2d(data_, h, n * sizeof(T));
    }
    void assign_device(const T* d, size_t n) {
        resize(0);
        resize(n);
        if (n) copy_d2d(data_, d, n * sizeof(T));
    }
    std::shared_ptr<void> store_;  // owns the block data_ points to
    T* data_ = nullptr;
    size_t size_ = 0, cap_ = 0;
};

}  // namespace utility
}  // namespace cupoch
