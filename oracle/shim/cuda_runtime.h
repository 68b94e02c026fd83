// TEST INFRASTRUCTURE ONLY (oracle/). "CUDA-on-CPU" shim: just enough of the CUDA
// runtime surface for the reference's extensions/ngp_raymarch/src/*.cu to compile
// with g++ and run serially on the host, so the reference's OWN code is the oracle
// for K1..K11. Nothing here is shipped in the product path.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <cassert>
#include <cstdlib>
#include <type_traits>
#include <algorithm>

#define __host__
#define __device__
#define __global__
#define __shared__
#ifndef __forceinline__
#define __forceinline__ inline
#endif

struct xr_shim_dim3 { unsigned x = 0, y = 0, z = 0; };
extern thread_local xr_shim_dim3 threadIdx, blockIdx, blockDim, gridDim;
static const int warpSize = 32;

typedef void* cudaStream_t;
typedef int cudaError_t;
inline cudaError_t cudaDeviceSynchronize() { return 0; }
inline cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t) { memset(p, v, n); return 0; }

struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int4 { int x, y, z, w; };

// serial atomics: the shim launcher runs one "thread" at a time
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline uint32_t atomicMax(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
template <typename T> inline T __shfl_xor_sync(unsigned, T v, int) { return v; }
inline void __syncthreads() {}

// CUDA's mixed-type global min/max overloads (ray_sampler_header.h:42,53 use min(uint32_t,int))
template <typename A, typename B>
inline typename std::common_type<A, B>::type min(A a, B b) {
    typedef typename std::common_type<A, B>::type C; return (C)a < (C)b ? (C)a : (C)b; }
template <typename A, typename B>
inline typename std::common_type<A, B>::type max(A a, B b) {
    typedef typename std::common_type<A, B>::type C; return (C)a > (C)b ? (C)a : (C)b; }
