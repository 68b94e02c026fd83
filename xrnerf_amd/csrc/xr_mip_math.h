// Per-element arithmetic of the Mip-NeRF kernels (xr_mip.hip), fp32 in the reference's operation order.
// __host__ __device__ so that tests/host_harness/mip_math_host.cpp (tests/test_mip_host_math.py) can run the very same expressions on the CPU
// against the numpy oracle before a kernel ever reaches a GPU (compile BOTH with -ffp-contract=off).
//
// Reference (relative to /root/reference/):
//   xrnerf/datasets/pipelines/create.py:502-531            GetZvals
//   xrnerf/models/networks/utils/mip.py:65-131             lift_gaussian, conical_frustum_to_gaussian, cylinder_to_gaussian
//   xrnerf/models/embedders/mipnerf_embedder.py:34-83      expected_sin, integrated_pos_enc, pos_enc
//   xrnerf/models/renders/nerf_render.py:64-88             sigmoid / padding / density activation
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef __HIPCC__
#define __host__
#define __device__
#endif

#define XR_MIP_HALF_PI 1.5707964f          /* 0.5 * float(pi) */

// torch.linspace(start, end, n)[j] in fp32 (aten RangeFactories: symmetric around the middle; its vectorised CPU
// kernel and its GPU kernel both evaluate start + step*j with ONE rounding, hence the explicit fmaf -- checked
// bit for bit against torch.linspace in tests/test_mip_host_math.py)
__host__ __device__ inline float xr_torch_linspace(float start, float end, uint32_t n, uint32_t j) {
    if (n <= 1) return start;
    float step = (end - start) / (float)(n - 1);
    return j < n / 2 ? fmaf(step, (float)j, start) : fmaf(-step, (float)(n - 1 - j), end);
}

// un-jittered z of GetZvals (create.py:509-516)
__host__ __device__ inline float xr_mip_zval(float near, float far, uint32_t n, uint32_t j, int lindisp) {
    float t = xr_torch_linspace(0.f, 1.f, n, j);
    if (!lindisp) return near * (1.f - t) + far * t;
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
}

// conical_frustum_to_gaussian (stable=True) / cylinder_to_gaussian + lift_gaussian(diag=True), one interval of one ray.
// o, d: ray origin / direction; out: mean[3], cov[3]
__host__ __device__ inline void xr_mip_gaussian(const float o[3], const float d[3], float radius, float t0, float t1,
                                                int cylinder, float mean[3], float cov[3]) {
    float t_mean, t_var, r_var;
    if (!cylinder) {
        float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
        float mu2 = mu * mu, hw2 = hw * hw;
        float den = 3.f * mu2 + hw2;
        float hw4 = hw2 * hw2;                                // torch: hw**4 == (hw**2)**2 in fp32
        t_mean = mu + (2.f * mu * hw2) / den;
        t_var = hw2 / 3.f - (float)(4.0 / 15.0) * ((hw4 * (12.f * mu2 - hw2)) / (den * den));
        r_var = (radius * radius) * (mu2 / 4.f + (float)(5.0 / 12.0) * hw2 - (float)(4.0 / 15.0) * hw4 / den);
    } else {
        t_mean = (t0 + t1) / 2.f;
        r_var = radius * radius / 4.f;
        float dt = t1 - t0;
        t_var = dt * dt / 12.f;
    }
    float d2[3] = {d[0] * d[0], d[1] * d[1], d[2] * d[2]};
    float mag = fmaxf(1e-10f, d2[0] + d2[1] + d2[2]);
    for (int a = 0; a < 3; ++a) {
        mean[a] = d[a] * t_mean + o[a];
        float null_outer = 1.f - d2[a] / mag;
        cov[a] = t_var * d2[a] + r_var * null_outer;
    }
}

// one column of MipNerfEmbedder.forward's row for a sample with gaussian (mean, cov) and view direction vd.
// columns: [sin block 3L | sin(.+pi/2) block 3L | (vd identity 3) | view sin 3Lv | view sin(.+pi/2) 3Lv]
__host__ __device__ inline float xr_mip_feature(uint32_t c, const float mean[3], const float cov[3], const float vd[3],
                                                int min_deg, int max_deg, int min_deg_view, int max_deg_view,
                                                int append_identity) {
    const uint32_t L3 = 3u * (uint32_t)(max_deg - min_deg);
    if (c < 2 * L3) {
        uint32_t cc = c < L3 ? c : c - L3;
        int k = min_deg + (int)(cc / 3u);
        uint32_t a = cc % 3u;
        float y = ldexpf(mean[a], k);                          // x * 2^k: exact
        float yv = ldexpf(cov[a], 2 * k);                      // cov * (2^k)^2: exact
        if (c >= L3) y = y + XR_MIP_HALF_PI;
        return expf(-0.5f * yv) * sinf(y);
    }
    c -= 2 * L3;
    if (append_identity) {
        if (c < 3) return vd[c];
        c -= 3;
    }
    const uint32_t V3 = 3u * (uint32_t)(max_deg_view - min_deg_view);
    uint32_t cc = c < V3 ? c : c - V3;
    int k = min_deg_view + (int)(cc / 3u);
    float y = ldexpf(vd[cc % 3u], k);
    if (c >= V3) y = y + XR_MIP_HALF_PI;
    return sinf(y);
}

// ---- render (nerf_render.py:64-88)
__host__ __device__ inline float xr_mip_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
// F.softplus(beta=1, threshold=20) / F.relu
__host__ __device__ inline float xr_mip_density_act(float x, int relu) {
    if (relu) return x > 0.f ? x : 0.f;
    return x > 20.f ? x : log1pf(expf(x));
}
__host__ __device__ inline float xr_mip_density_dact(float x, int relu) {
    if (relu) return x > 0.f ? 1.f : 0.f;
    return x > 20.f ? 1.f : xr_mip_sigmoid(x);
}
