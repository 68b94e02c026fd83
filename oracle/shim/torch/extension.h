// TEST INFRASTRUCTURE ONLY. Fake torch::Tensor: the reference .cu files use only
// data_ptr(), sizes()[i] and scalar_type() (raymarch_shared.h:59-62).
#pragma once
#include <vector>
#include <cstdint>
namespace at {
enum class ScalarType { Byte, Int, Float, Half };
struct Half { unsigned short v; operator float() const { return 0.f; } Half() {} Half(float) {} };
}
namespace torch {
struct Tensor {
    void* p = nullptr; std::vector<int64_t> s; at::ScalarType t = at::ScalarType::Float;
    Tensor() {}
    Tensor(void* p_, std::vector<int64_t> s_, at::ScalarType t_ = at::ScalarType::Float) : p(p_), s(s_), t(t_) {}
    void* data_ptr() const { return p; }
    template <typename T> T* data_ptr() const { return (T*)p; }
    const std::vector<int64_t>& sizes() const { return s; }
    at::ScalarType scalar_type() const { return t; }
};
}
