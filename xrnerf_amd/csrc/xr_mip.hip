// Mip-NeRF sampling / integrated positional encoding / renderer kernels (BASELINE config #3)
// for gfx950.  The reference runs this as ~60 PyTorch elementwise / reduction launches per level
// (xrnerf/models/networks/utils/mip.py, embedders/mipnerf_embedder.py, renders/{nerf,mipnerf}_render.py),
// each streaming [R,S,*] tensors through HBM; here each stage is ONE launch whose only HBM traffic is its
// algorithmic input and output:
//   k_mip_zvals     thread = (ray, edge)                      8 B/ray in, 4 B/edge out
//   k_mip_encode    workgroup = 64 consecutive samples; the gaussians are built once per sample in LDS, then the
//                   lanes sweep the tile's flattened [64 x ch] output so that every store is a fully coalesced
//                   dword row segment (ch = 123 floats = 492 B/sample, not 16-B aligned: no vector stores)
//   k_mip_render*   wave = ray; 64 intervals per sweep, transmittance through a wave-wide fp64 prefix sum
//                   (torch's CPU cumsum accumulates fp32 in double), reductions by butterfly
//   k_mip_resample  wave = ray; blurred pdf -> fp64 scan -> cdf in LDS -> per-output binary search
// All HBM-streaming: bytes per unit are in DESIGN.md section 8.
#include "xr_common.h"
#include "xr_mip_math.h"

#define MIP_TILE 64
#define MIP_BLOCK 256
#define MIP_MAX_NZ 2048u

// ------------------------------------------------------------------------------------------ wave helpers
__device__ inline double wave_incl_scan(double v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------ GetZvals
__global__ void k_mip_zvals(const float* __restrict__ near, const float* __restrict__ far, uint32_t n_rays,
                            uint32_t n_z, int lindisp, const float* __restrict__ z_rand, float* __restrict__ z_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_rays * n_z) return;
    const uint32_t r = (uint32_t)(i / n_z), j = (uint32_t)(i % n_z);
    const float nr = near[r], fr = far[r];
    const float z = xr_mip_zval(nr, fr, n_z, j, lindisp);
    if (z_rand == nullptr) { z_out[i] = z; return; }
    // create.py:518-525: one uniform draw inside [lower, upper] = neighbouring midpoints (ends clamp to z itself)
    const float zl = j > 0 ? xr_mip_zval(nr, fr, n_z, j - 1, lindisp) : z;
    const float zu = j + 1 < n_z ? xr_mip_zval(nr, fr, n_z, j + 1, lindisp) : z;
    const float lower = j > 0 ? 0.5f * (z + zl) : z;
    const float upper = j + 1 < n_z ? 0.5f * (zu + z) : z;
    z_out[i] = lower + (upper - lower) * z_rand[i];
}

// ------------------------------------------------------------------------------------------ cast_rays + IPE + view PE
struct MipEncArgs {
    const float* rays_o; const float* rays_d; const float* viewdirs; const float* radii; const float* z_vals;
    const float* means; const float* covs;      // alternative source: gaussians already built (mip.py's data['samples'])
    uint32_t n_rays, n_s;                      // n_s = intervals per ray
    int min_deg, max_deg, min_deg_view, max_deg_view, append_identity, cylinder;
    uint32_t ch, ld;
    float* out;
};

__global__ void __launch_bounds__(MIP_BLOCK) k_mip_encode(MipEncArgs a) {
    __shared__ float s_g[9][MIP_TILE];          // mean3, cov3, viewdir3 per sample of the tile
    const uint64_t n_total = (uint64_t)a.n_rays * a.n_s;
    const uint64_t g0 = (uint64_t)blockIdx.x * MIP_TILE;
    const uint32_t tile = (uint32_t)(n_total - g0 < MIP_TILE ? n_total - g0 : MIP_TILE);
    if (threadIdx.x < tile) {
        const uint64_t g = g0 + threadIdx.x;
        const uint32_t r = (uint32_t)(g / a.n_s), s = (uint32_t)(g % a.n_s);
        float mean[3], cov[3];
        if (a.means != nullptr) {
            for (int k = 0; k < 3; ++k) { mean[k] = a.means[g * 3 + k]; cov[k] = a.covs[g * 3 + k]; }
        } else {
            const float o[3] = {a.rays_o[r * 3ull], a.rays_o[r * 3ull + 1], a.rays_o[r * 3ull + 2]};
            const float d[3] = {a.rays_d[r * 3ull], a.rays_d[r * 3ull + 1], a.rays_d[r * 3ull + 2]};
            const float* z = a.z_vals + (uint64_t)r * (a.n_s + 1) + s;
            xr_mip_gaussian(o, d, a.radii[r], z[0], z[1], a.cylinder, mean, cov);
        }
        for (int k = 0; k < 3; ++k) {
            s_g[k][threadIdx.x] = mean[k];
            s_g[3 + k][threadIdx.x] = cov[k];
            s_g[6 + k][threadIdx.x] = a.viewdirs[r * 3ull + k];
        }
    }
    __syncthreads();
    const uint32_t n_el = tile * a.ch;
    for (uint32_t e = threadIdx.x; e < n_el; e += MIP_BLOCK) {
        const uint32_t sl = e / a.ch, c = e - sl * a.ch;
        const float mean[3] = {s_g[0][sl], s_g[1][sl], s_g[2][sl]};
        const float cov[3] = {s_g[3][sl], s_g[4][sl], s_g[5][sl]};
        const float vd[3] = {s_g[6][sl], s_g[7][sl], s_g[8][sl]};
        a.out[(g0 + sl) * a.ld + c] =
            xr_mip_feature(c, mean, cov, vd, a.min_deg, a.max_deg, a.min_deg_view, a.max_deg_view, a.append_identity);
    }
}

// ------------------------------------------------------------------------------------------ renderer
struct MipRenderArgs {
    const float* raw; const float* z_vals; const float* rays_d;
    uint32_t n_rays, n_s;
    float density_bias, rgb_padding;
    int white_bkgd, relu;
};

// per-interval quantities shared by forward and backward
struct MipSample { float rgb[3], sg[3], x, dd, dist, zmid; };
__device__ inline MipSample mip_load_sample(const MipRenderArgs& a, uint32_t r, uint32_t i, float dnorm) {
    MipSample s;
    const float4 v = reinterpret_cast<const float4*>(a.raw)[(uint64_t)r * a.n_s + i];
    const float* z = a.z_vals + (uint64_t)r * (a.n_s + 1) + i;
    const float z0 = z[0], z1 = z[1];
    s.dist = (z1 - z0) * dnorm;
    s.zmid = 0.5f * (z0 + z1);
    const float in[3] = {v.x, v.y, v.z};
    for (int c = 0; c < 3; ++c) {
        s.sg[c] = xr_mip_sigmoid(in[c]);
        s.rgb[c] = s.sg[c] * (1.f + 2.f * a.rgb_padding) - a.rgb_padding;
    }
    s.x = v.w + a.density_bias;
    s.dd = xr_mip_density_act(s.x, a.relu) * s.dist;
    return s;
}

__global__ void __launch_bounds__(MIP_BLOCK) k_mip_render_fwd(MipRenderArgs a, float* __restrict__ rgb_out,
                                                              float* __restrict__ dist_out, float* __restrict__ acc_out,
                                                              float* __restrict__ weights_out) {
    const uint32_t r = blockIdx.x * (MIP_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rays) return;                 // whole waves leave together
    const float dx = a.rays_d[r * 3ull], dy = a.rays_d[r * 3ull + 1], dz = a.rays_d[r * 3ull + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    double carry = 0.0;                        // sum of density_delta over the intervals before this sweep
    float acc_c[3] = {0.f, 0.f, 0.f}, acc_w = 0.f, acc_z = 0.f;
    for (uint32_t base = 0; base < a.n_s; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i < a.n_s;
        MipSample s;
        double dd = 0.0;
        if (live) { s = mip_load_sample(a, r, i, dnorm); dd = (double)s.dd; }
        const double incl = wave_incl_scan(dd);
        if (live) {
            const float before = (float)(carry + (incl - dd));          // exclusive cumsum, rounded like torch's output
            const float w = (1.f - expf(-s.dd)) * expf(-before);
            weights_out[(uint64_t)r * a.n_s + i] = w;
            acc_w += w;
            acc_z += w * s.zmid;
            for (int c = 0; c < 3; ++c) acc_c[c] += w * s.rgb[c];
        }
        carry += __shfl(incl, 63, 64);
    }
    const float acc = wave_sum(acc_w), depth = wave_sum(acc_z);
    float col[3];
    for (int c = 0; c < 3; ++c) col[c] = wave_sum(acc_c[c]);
    if (lane == 0) {
        const float* z = a.z_vals + (uint64_t)r * (a.n_s + 1);
        float q = depth / acc;                                          // mipnerf_render.py:17-23
        if (q != q) q = INFINITY;
        q = fmaxf(fminf(q, z[a.n_s]), z[0]);
        dist_out[r] = q;
        acc_out[r] = acc;
        for (int c = 0; c < 3; ++c) rgb_out[r * 3ull + c] = a.white_bkgd ? col[c] + (1.f - acc) : col[c];
    }
}

// dL/draw given dL/drgb (the reference trains on the rendered colours only, networks/mipnerf.py:52-60).
// With c' = rgb - [white], gc_k = sum_ch g_ch c'_k,ch:
//   dL/d(dd_k) = T_{k+1} gc_k - sum_{i>k} w_i gc_i,   sum_{i>k} = total - inclusive prefix (both in fp64)
__global__ void __launch_bounds__(MIP_BLOCK) k_mip_render_bwd(MipRenderArgs a, const float* __restrict__ grad_rgb,
                                                              float* __restrict__ grad_raw) {
    const uint32_t r = blockIdx.x * (MIP_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rays) return;
    const float dx = a.rays_d[r * 3ull], dy = a.rays_d[r * 3ull + 1], dz = a.rays_d[r * 3ull + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float g[3] = {grad_rgb[r * 3ull], grad_rgb[r * 3ull + 1], grad_rgb[r * 3ull + 2]};
    const float white = a.white_bkgd ? 1.f : 0.f;
    // pass 1: total = sum_i w_i gc_i
    double carry = 0.0, total = 0.0;
    for (uint32_t base = 0; base < a.n_s; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i < a.n_s;
        MipSample s;
        double dd = 0.0, wg = 0.0;
        if (live) { s = mip_load_sample(a, r, i, dnorm); dd = (double)s.dd; }
        const double incl = wave_incl_scan(dd);
        if (live) {
            const float before = (float)(carry + (incl - dd));
            const float w = (1.f - expf(-s.dd)) * expf(-before);
            float gc = 0.f;
            for (int c = 0; c < 3; ++c) gc += g[c] * (s.rgb[c] - white);
            wg = (double)w * (double)gc;
        }
        total += wave_sum(wg);
        carry += __shfl(incl, 63, 64);
    }
    // pass 2: gradients
    carry = 0.0;
    double carry_wg = 0.0;
    for (uint32_t base = 0; base < a.n_s; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i < a.n_s;
        MipSample s;
        double dd = 0.0, wg = 0.0;
        float w = 0.f, gc = 0.f, before = 0.f;
        if (live) { s = mip_load_sample(a, r, i, dnorm); dd = (double)s.dd; }
        const double incl = wave_incl_scan(dd);
        if (live) {
            before = (float)(carry + (incl - dd));
            w = (1.f - expf(-s.dd)) * expf(-before);
            for (int c = 0; c < 3; ++c) gc += g[c] * (s.rgb[c] - white);
            wg = (double)w * (double)gc;
        }
        const double incl_wg = wave_incl_scan(wg);
        if (live) {
            const double suffix = total - (carry_wg + incl_wg);
            const float t_next = expf(-before) * expf(-s.dd);           // T_{k+1}
            const float d_dd = (float)((double)(t_next * gc) - suffix);
            float4 o;
            const float k = 1.f + 2.f * a.rgb_padding;
            o.x = g[0] * w * (s.sg[0] * (1.f - s.sg[0])) * k;
            o.y = g[1] * w * (s.sg[1] * (1.f - s.sg[1])) * k;
            o.z = g[2] * w * (s.sg[2] * (1.f - s.sg[2])) * k;
            o.w = d_dd * s.dist * xr_mip_density_dact(s.x, a.relu);
            reinterpret_cast<float4*>(grad_raw)[(uint64_t)r * a.n_s + i] = o;
        }
        carry += __shfl(incl, 63, 64);
        carry_wg += __shfl(incl_wg, 63, 64);
    }
}

// ------------------------------------------------------------------------------------------ resample_along_rays
// mip.py:151-176 + sorted_piecewise_constant_pdf (:7-62): z_new [R, n_z] from (z [R, n_z], weights [R, n_z-1]).
// LDS per wave: cdf [n_z] and the blurred weights / pdf [n_z-1] (bins are read from global: two loads per output).
__global__ void __launch_bounds__(MIP_BLOCK) k_mip_resample(const float* __restrict__ z_vals,
                                                            const float* __restrict__ weights,
                                                            const float* __restrict__ rand, float resample_padding,
                                                            uint32_t n_rays, uint32_t n_z, float s32, float span32,
                                                            float* __restrict__ z_out) {
    extern __shared__ float s_mem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (MIP_BLOCK / 64) + wave;
    if (r >= n_rays) return;                   // no block-wide barrier below: waves are independent
    const uint32_t n = n_z - 1;                // intervals
    float* cdf = s_mem + (size_t)wave * (2 * n_z);
    float* wb = cdf + n_z;
    const float* w = weights + (uint64_t)r * n;
    const float* z = z_vals + (uint64_t)r * n_z;
    // blurred weights (mip.py:159-168): 0.5 * (max(w[i-1], w[i]) + max(w[i], w[i+1])) + padding, indices clamped
    double part = 0.0;
    for (uint32_t i = lane; i < n; i += 64) {
        const float wc = w[i], wl = w[i > 0 ? i - 1 : 0], wr = w[i + 1 < n ? i + 1 : n - 1];
        const float v = 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + resample_padding;
        wb[i] = v;
        part += (double)v;
    }
    float weight_sum = (float)wave_sum(part);
    const float padding = fmaxf(0.f, 1e-5f - weight_sum);             // mip.py:12-16
    const float pad_each = padding / (float)n;
    weight_sum = weight_sum + padding;
    // cdf[0] = 0, cdf[i+1] = min(1, cumsum(pdf)[i]) for i < n-1, cdf[n] = 1        (mip.py:20-27)
    double carry = 0.0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        double p = 0.0;
        if (i < n) p = (double)((wb[i] + pad_each) / weight_sum);
        const double incl = wave_incl_scan(p);
        if (i + 1 < n) cdf[i + 1] = fminf(1.f, (float)(carry + incl));
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) { cdf[0] = 0.f; cdf[n] = 1.f; }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float one_m_eps = 0.99999988079071044921875f;               // fp32(1 - eps)
    for (uint32_t j = lane; j < n_z; j += 64) {
        float u;
        if (rand != nullptr) {                                          // mip.py:30-38
            u = (float)j * s32;
            u = u + rand[(uint64_t)r * n_z + j] * span32;
            u = fminf(u, one_m_eps);
        } else {
            u = xr_torch_linspace(0.f, one_m_eps, n_z, j);              // mip.py:41-43
        }
        // last index with cdf <= u (cdf[0] = 0 <= u always); cdf is non-decreasing
        uint32_t lo = 0, hi = n;                                        // answer in [lo, hi]
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (cdf[mid] <= u) lo = mid; else hi = mid - 1;
        }
        const uint32_t i1 = lo < n ? lo + 1 : n;
        const float c0 = cdf[lo], c1 = cdf[i1], b0 = z[lo], b1 = z[i1];
        float t = (u - c0) / (c1 - c0);
        if (t != t) t = 0.f;                                            // nan_to_num(., 0)
        t = fminf(fmaxf(t, 0.f), 1.f);
        z_out[(uint64_t)r * n_z + j] = b0 + t * (b1 - b0);
    }
}

// ------------------------------------------------------------------------------------------ C-ABI
extern "C" int xr_mip_zvals(const float* near, const float* far, uint32_t n_rays, uint32_t n_z, int lindisp,
                            const float* z_rand, float* z_out, void* stream) {
    XR_REQUIRE(n_z >= 2, "n_z must be >= 2");
    if (n_rays == 0) return XR_OK;            // empty batches carry null data pointers
    XR_REQUIRE(near && far && z_out, "null pointer");
    const uint64_t n = (uint64_t)n_rays * n_z;
    XR_REQUIRE(n < (1ull << 40), "too many samples");
    hipLaunchKernelGGL(k_mip_zvals, dim3(xr_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, near, far, n_rays, n_z,
                       lindisp, z_rand, z_out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" uint32_t xr_mip_encode_channels(int min_deg, int max_deg, int min_deg_view, int max_deg_view,
                                           int append_identity) {
    if (max_deg < min_deg || max_deg_view < min_deg_view) return 0;
    return 6u * (uint32_t)(max_deg - min_deg) + 6u * (uint32_t)(max_deg_view - min_deg_view) + (append_identity ? 3u : 0u);
}

static int mip_encode_launch(MipEncArgs a, void* stream) {
    a.ch = xr_mip_encode_channels(a.min_deg, a.max_deg, a.min_deg_view, a.max_deg_view, a.append_identity);
    XR_REQUIRE(a.ch > 0, "empty encoding");
    XR_REQUIRE(a.max_deg <= 60 && a.min_deg >= -60 && a.max_deg_view <= 60 && a.min_deg_view >= -60, "degree out of range");
    XR_REQUIRE(a.ld >= a.ch, "ld smaller than the row width");
    const uint64_t n = (uint64_t)a.n_rays * a.n_s;
    if (n == 0) return XR_OK;
    XR_REQUIRE(a.viewdirs && a.out, "null pointer");
    XR_REQUIRE(a.means ? (a.covs != nullptr) : (a.rays_o && a.rays_d && a.radii && a.z_vals), "null pointer");
    XR_REQUIRE(n <= 0xffffffffull * MIP_TILE, "too many samples");
    hipLaunchKernelGGL(k_mip_encode, dim3(xr_div_up(n, MIP_TILE)), dim3(MIP_BLOCK), 0, (hipStream_t)stream, a);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_mip_encode(const float* rays_o, const float* rays_d, const float* viewdirs, const float* radii,
                             const float* z_vals, uint32_t n_rays, uint32_t n_z, int min_deg, int max_deg,
                             int min_deg_view, int max_deg_view, int append_identity, int ray_shape, float* out,
                             uint32_t ld, void* stream) {
    XR_REQUIRE(n_z >= 2, "n_z must be >= 2");
    XR_REQUIRE(ray_shape == 0 || ray_shape == 1, "ray_shape: 0 = cone, 1 = cylinder");
    MipEncArgs a{rays_o, rays_d, viewdirs, radii, z_vals, nullptr, nullptr, n_rays, n_z - 1, min_deg, max_deg,
                 min_deg_view, max_deg_view, append_identity, ray_shape, 0, ld, out};
    return mip_encode_launch(a, stream);
}

extern "C" int xr_mip_encode_gaussians(const float* means, const float* covs, const float* viewdirs, uint32_t n_rays,
                                       uint32_t n_samples, int min_deg, int max_deg, int min_deg_view,
                                       int max_deg_view, int append_identity, float* out, uint32_t ld, void* stream) {
    XR_REQUIRE(n_rays == 0 || n_samples == 0 || means, "null pointer");
    MipEncArgs a{nullptr, nullptr, viewdirs, nullptr, nullptr, means, covs, n_rays, n_samples, min_deg, max_deg,
                 min_deg_view, max_deg_view, append_identity, 0, 0, ld, out};
    return mip_encode_launch(a, stream);
}

static int mip_render_args(MipRenderArgs& a, const float* raw, const float* z_vals, const float* rays_d,
                           uint32_t n_rays, uint32_t n_z, float density_bias, float rgb_padding, int white_bkgd,
                           int density_activation) {
    XR_REQUIRE(n_rays == 0 || (raw && z_vals && rays_d), "null pointer");
    XR_REQUIRE(n_z >= 2, "n_z must be >= 2");
    XR_REQUIRE(density_activation == 0 || density_activation == 1, "density_activation: 0 = softplus, 1 = relu");
    XR_REQUIRE(((uintptr_t)raw & 15) == 0, "raw must be 16-byte aligned");
    a = MipRenderArgs{raw, z_vals, rays_d, n_rays, n_z - 1, density_bias, rgb_padding, white_bkgd, density_activation};
    return XR_OK;
}

extern "C" int xr_mip_render_forward(const float* raw, const float* z_vals, const float* rays_d, uint32_t n_rays,
                                     uint32_t n_z, float density_bias, float rgb_padding, int white_bkgd,
                                     int density_activation, float* rgb, float* distance, float* acc, float* weights,
                                     void* stream) {
    MipRenderArgs a;
    int rc = mip_render_args(a, raw, z_vals, rays_d, n_rays, n_z, density_bias, rgb_padding, white_bkgd, density_activation);
    if (rc) return rc;
    if (n_rays == 0) return XR_OK;
    XR_REQUIRE(rgb && distance && acc && weights, "null pointer");
    hipLaunchKernelGGL(k_mip_render_fwd, dim3(xr_div_up(n_rays, MIP_BLOCK / 64)), dim3(MIP_BLOCK), 0, (hipStream_t)stream,
                       a, rgb, distance, acc, weights);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_mip_render_backward(const float* raw, const float* z_vals, const float* rays_d,
                                      const float* grad_rgb, uint32_t n_rays, uint32_t n_z, float density_bias,
                                      float rgb_padding, int white_bkgd, int density_activation, float* grad_raw,
                                      void* stream) {
    MipRenderArgs a;
    int rc = mip_render_args(a, raw, z_vals, rays_d, n_rays, n_z, density_bias, rgb_padding, white_bkgd, density_activation);
    if (rc) return rc;
    if (n_rays == 0) return XR_OK;
    XR_REQUIRE(grad_rgb && grad_raw, "null pointer");
    XR_REQUIRE(((uintptr_t)grad_raw & 15) == 0, "grad_raw must be 16-byte aligned");
    hipLaunchKernelGGL(k_mip_render_bwd, dim3(xr_div_up(n_rays, MIP_BLOCK / 64)), dim3(MIP_BLOCK), 0, (hipStream_t)stream,
                       a, grad_rgb, grad_raw);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_mip_resample(const float* z_vals, const float* weights, const float* rand, float resample_padding,
                               uint32_t n_rays, uint32_t n_z, float* z_out, void* stream) {
    XR_REQUIRE(n_z >= 2 && n_z <= MIP_MAX_NZ, "n_z must be in [2, 2048]");
    if (n_rays == 0) return XR_OK;
    XR_REQUIRE(z_vals && weights && z_out, "null pointer");
    XR_REQUIRE(z_out != z_vals, "in-place resampling is not supported");
    // mip.py:31-35: s = 1/num_samples and (s - eps) are python doubles that the tensor ops round to fp32
    const float s32 = (float)(1.0 / (double)n_z);
    const float span32 = (float)(1.0 / (double)n_z - (double)1.1920928955078125e-07);
    const size_t lds = (size_t)(MIP_BLOCK / 64) * 2 * n_z * sizeof(float);
    hipLaunchKernelGGL(k_mip_resample, dim3(xr_div_up(n_rays, MIP_BLOCK / 64)), dim3(MIP_BLOCK), lds, (hipStream_t)stream,
                       z_vals, weights, rand, resample_padding, n_rays, n_z, s32, span32, z_out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
