// TEST INFRASTRUCTURE ONLY -- a HIP-on-CPU execution shim: the kernel sources of xrnerf_amd/csrc/*.hip are compiled
// UNCHANGED for the host (clang++ -x c++; this directory first on the include path) and run by tests/hip_emu/emu.cpp,
// one cooperative fiber per GPU thread, wave64 semantics: __shfl* / __ballot / v_mfma_f32_32x32x2_f32 are collectives
// over the 64 lanes of a wave, __syncthreads over the workgroup, LDS = ordinary memory shared by the workgroup's fibers.
// It checks FUNCTION (indexing, collectives, layouts, edge cases) without a GPU; timing means nothing here and libm's
// sin/exp stand in for ocml's.  Nothing under xrnerf_amd/ includes or links this.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#define __HIP_EMU__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* kernel-local: one instance per host thread (a thread runs one workgroup at a time) */
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3e { unsigned x, y, z; };
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 2; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
template <typename F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }

struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 { float x, y; } __attribute__((aligned(8)));
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct float3 { float x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
struct uint2 { unsigned x, y; } __attribute__((aligned(8)));
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct uint4 { unsigned x, y, z, w; } __attribute__((aligned(16)));
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct int2 { int x, y; } __attribute__((aligned(8)));
struct double2 { double x, y; } __attribute__((aligned(16)));
inline double2 make_double2(double x, double y) { return double2{x, y}; }
// HIP's global min / max overloads
template <typename T> inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> inline T max(T a, T b) { return a < b ? b : a; }
inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
inline unsigned long min(unsigned long a, unsigned b) { return min(a, (unsigned long)b); }
inline unsigned long min(unsigned a, unsigned long b) { return min((unsigned long)a, b); }
inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }
// round-to-nearest single operations that must not be contracted (the build uses -ffp-contract=off anyway)
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __frcp_rn(float a) { volatile float r = 1.0f / a; return r; }

namespace emu {
struct Lane;                               // one GPU thread
Lane* cur();
const uint3e& tid(); const uint3e& bid(); const dim3& bdim(); const dim3& gdim();
void launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()>& body);
void block_barrier();
int block_or(int v);
// wave collectives on raw 64-bit payloads
uint64_t wave_exchange(uint64_t mine, int src_lane);          // value deposited by lane src_lane (own value if that lane is gone)
uint64_t wave_ballot(bool pred);
void wave_mfma_32x32x2(float a, float b, float* c16);
void wave_mfma_32x32x16(const float* a8, const float* b8, float* c16);
void* dyn_lds();
}
#define threadIdx (emu::tid())
#define blockIdx (emu::bid())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { emu::block_barrier(); }
inline int __syncthreads_or(int v) { return emu::block_or(v); }
inline void __threadfence_block() {}
inline void __threadfence() {}
// on the GPU a wave runs in lockstep, so lanes may hand data to each other through LDS with nothing but a scheduling
// barrier in between; here lanes are separate fibers, so the same spot has to be a real rendezvous
namespace emu { void wave_barrier(); }
inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier(); }
inline void __builtin_amdgcn_sched_barrier(int) {}

namespace emu {
template <typename T> inline uint64_t pack(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&u, &v, sizeof(T)); return u; }
template <typename T> inline T unpack(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
int lane_id();
}
// HIP semantics with sub-wave groups: `width` consecutive lanes form a group, source lanes are relative to it
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    const int l = emu::lane_id(), g0 = l & ~(width - 1);
    return emu::unpack<T>(emu::wave_exchange(emu::pack(v), g0 + (src & (width - 1))));
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = emu::lane_id(), g0 = l & ~(width - 1);
    return emu::unpack<T>(emu::wave_exchange(emu::pack(v), l - (int)d >= g0 ? l - (int)d : l));
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = emu::lane_id(), g0 = l & ~(width - 1);
    return emu::unpack<T>(emu::wave_exchange(emu::pack(v), l + (int)d < g0 + width ? l + (int)d : l));
}
template <typename T> inline T __shfl_xor(T v, int m, int width = 64) {
    const int l = emu::lane_id(), g0 = l & ~(width - 1), s = l ^ m;
    return emu::unpack<T>(emu::wave_exchange(emu::pack(v), s < g0 + width && s >= g0 ? s : l));
}
inline unsigned long long __ballot(int p) { return emu::wave_ballot(p != 0); }
inline int __any(int p) { return emu::wave_ballot(p != 0) != 0; }
// v_med3_f32 (no NaN operands in the kernels' uses)
inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b)); }
// value of the wave's lane 0 (the kernels only use it on wave-uniform values, to tell the compiler they ARE uniform)
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32_emu(float a, float b, emu_f32x16 c, int, int, int) {
    float t[16];
    for (int r = 0; r < 16; ++r) t[r] = c[r];
    emu::wave_mfma_32x32x2(a, b, t);
    for (int r = 0; r < 16; ++r) c[r] = t[r];
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 __builtin_amdgcn_mfma_f32_32x32x2f32_emu
// v_mfma_f32_32x32x16_f16: A / B = 8 halves per lane (k-slots (lane >> 5, e)), products exact in fp32, fp32 accumulation
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16_emu(emu_h8 a, emu_h8 b, emu_f32x16 c, int, int, int) {
    float fa[8], fb[8], t[16];
    for (int e = 0; e < 8; ++e) { fa[e] = (float)a[e]; fb[e] = (float)b[e]; }
    for (int r = 0; r < 16; ++r) t[r] = c[r];
    emu::wave_mfma_32x32x16(fa, fb, t);
    for (int r = 0; r < 16; ++r) c[r] = t[r];
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 __builtin_amdgcn_mfma_f32_32x32x16_f16_emu
// v_mfma_f32_32x32x16_bf16: the same operand / accumulator layout with bf16 operands (host __bf16: round-to-nearest-even
// conversions like v_cvt_pk_bf16_f32)
typedef __bf16 emu_b8 __attribute__((ext_vector_type(8)));
inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16_emu(emu_b8 a, emu_b8 b, emu_f32x16 c, int, int, int) {
    float fa[8], fb[8], t[16];
    for (int e = 0; e < 8; ++e) { fa[e] = (float)a[e]; fb[e] = (float)b[e]; }
    for (int r = 0; r < 16; ++r) t[r] = c[r];
    emu::wave_mfma_32x32x16(fa, fb, t);
    for (int r = 0; r < 16; ++r) c[r] = t[r];
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 __builtin_amdgcn_mfma_f32_32x32x16_bf16_emu

// ds_read_b64_tr_b16 (gfx950's transposing LDS load): inside every group of 16 lanes, lane j receives element (j & 3) of the four 8-byte
// rows addressed by lanes (j >> 2) + 4 e, e = 0..3 -- the mapping measured on the MI355X (profiles/r02_ds_read_tr_b16_lane_mapping.txt)
typedef short emu_s4 __attribute__((ext_vector_type(4)));
inline emu_s4 emu_ds_read_tr16_b64(const void* my_row) {
    const int l = emu::lane_id(), g0 = l & ~15, j = l & 15;
    emu_s4 out;
    for (int e = 0; e < 4; ++e) {
        const uint64_t addr = emu::wave_exchange((uint64_t)(uintptr_t)my_row, g0 + (j >> 2) + 4 * e);
        out[e] = reinterpret_cast<const short*>((uintptr_t)addr)[j & 3];
    }
    return out;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(p))
#define XR_LDS_PTR(T, p) ((T*)(p))

// cooperative fibers never pre-empt each other: plain read-modify-write is atomic here
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <typename T> inline T unsafeAtomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
inline void sincosf_emu(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
