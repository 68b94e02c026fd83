// Shared device/host helpers for libxrnerf_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/xrnerf_mi355.h"
// a generic pointer into LDS as the LDS-address-space pointer the ds_* builtins take (the host build of the kernels, tests/hip_emu, defines it
// as a plain cast before this header is read)
#ifndef XR_LDS_PTR
#define XR_LDS_PTR(T, p) ((T __attribute__((address_space(3)))*)(p))
#endif
extern "C" __attribute__((visibility("hidden"))) int xr_device_cus(void);        // compute units of the current device (xr_mlp.hip); not exported

#define XR_WAVE 64

void xr_set_error(const char* fmt, ...);

#define XR_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) { xr_set_error("%s: %s", __func__, msg); return XR_EINVAL; } \
    } while (0)

#define XR_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) { xr_set_error("%s: %s", __func__, hipGetErrorString(e_)); return XR_EHIP; } \
    } while (0)

#define XR_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) { xr_set_error("%s: %s", __func__, hipGetErrorString(e_)); return XR_EHIP; } \
    } while (0)

static inline uint32_t xr_div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------- constants
// /root/reference/extensions/ngp_raymarch/include/raymarch_shared.h:41-56
#define XR_SQRT3 1.73205080757f
__host__ __device__ inline float xr_min_step() { return XR_SQRT3 / 1024.0f; }
__host__ __device__ inline float xr_max_step() { return XR_SQRT3 / 1024.0f * 128.0f * 1024.0f / 128.0f; }
__host__ __device__ inline float xr_max_warp_step() { return XR_SQRT3 / 1024.0f * 128.0f; }

// ---------------------------------------------------------------- PCG32 (pcg32.h:39-166)
struct xr_pcg32 {
    uint64_t state, inc;
    __host__ __device__ uint32_t next_uint() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    __host__ __device__ float next_float() {
        uint32_t u = (next_uint() >> 9) | 0x3f800000u;
        float f;
#if defined(__HIP_DEVICE_COMPILE__)
        f = __uint_as_float(u);
#else
        memcpy(&f, &u, 4);
#endif
        return f - 1.0f;
    }
    __host__ __device__ void advance(uint64_t delta) {
        uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        while (delta > 0) {
            if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
    __host__ __device__ void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u; inc = (initseq << 1u) | 1u;
        next_uint(); state += initstate; next_uint();
    }
};

// ---------------------------------------------------------------- Morton (raymarch_shared.h:128-136,753-768)
__host__ __device__ inline uint32_t xr_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ inline uint32_t xr_morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return xr_expand_bits(x) | (xr_expand_bits(y) << 1) | (xr_expand_bits(z) << 2);
}
__host__ __device__ inline uint32_t xr_morton3d_invert(uint32_t x) {
    x = x & 0x49249249; x = (x | (x >> 2)) & 0xc30c30c3; x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff; x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

// ---------------------------------------------------------------- activations
__device__ inline float xr_clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
__device__ inline float xr_logistic(float x) { return 1.0f / (1.0f + __expf(-x)); }
// ray_sampler_header.h:440-456
__device__ inline float xr_act_rgb(float v, int a) {
    switch (a) { case 0: return v; case 1: return v > 0.f ? v : 0.f; case 2: return xr_logistic(v);
                 default: return __expf(xr_clampf(v, -10.f, 10.f)); }
}
// raymarch_shared.h:626-642
__device__ inline float xr_act_density(float v, int a) {
    switch (a) { case 0: return v; case 1: return v > 0.f ? v : 0.f; case 2: return xr_logistic(v);
                 default: return __expf(v); }
}
// ray_sampler_header.h:534-574
__device__ inline float xr_dact_rgb(float v, int a) {
    switch (a) { case 0: return 1.f; case 1: return v > 0.f ? 1.f : 0.f;
                 case 2: { float s = xr_logistic(v); return s * (1.f - s); }
                 default: return __expf(xr_clampf(v, -10.f, 10.f)); }
}
__device__ inline float xr_dact_density(float v, int a) {
    switch (a) { case 0: return 1.f; case 1: return v > 0.f ? 1.f : 0.f;
                 case 2: { float s = xr_logistic(v); return s * (1.f - s); }
                 default: return __expf(xr_clampf(v, -15.f, 15.f)); }
}
__host__ __device__ inline float xr_unwarp_dt(float dt) {   // ray_sampler_header.h:388-392
    return dt * (xr_max_warp_step() - xr_min_step()) + xr_min_step();
}

// rows per segment of the live-row compaction (xr_live_rows counts / ranks per segment; xr_composite_train2 can produce the counts)
#define XR_LIVE_SEG 1024u

// the caller-provided helper stream of this thread (xr_set_helper_stream), or nullptr: see xr_scatter.hip
struct XrHelper { hipStream_t stream; hipEvent_t fork, join; };
const XrHelper* xr_internal_helper();
// every level of this geometry has a non-atomic scatter path at a capacity of n rows (what the update inside the scatter needs): xr_encode.hip
int xr_internal_hashgrid_bwd_adam_supported(uint32_t n, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                            const uint32_t* offset_host);
// library-internal (not part of the C ABI): see xr_mlp.hip
void xr_internal_defer_mlp_reduce(bool on);
int xr_internal_mlp_bwd_reduce(const void* workspace, uint32_t n, int n_hidden_density, int n_hidden_color, float* grad_w_density, float* grad_w_color, int overwrite, void* stream);
