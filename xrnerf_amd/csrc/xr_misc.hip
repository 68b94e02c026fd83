// Library plumbing (error string, version, RNG host helper) and the small kernels on either side
// of the path: ray generation, Huber loss gradient, fused Adam(+EMA).
#include "xr_common.h"
#include "xr_adam.h"
#include <cstdlib>
#include <cstdarg>

static thread_local char g_err[512] = "";
void xr_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* xr_last_error(void) { return g_err; }
extern "C" int xr_version(void) { return 122; }    // 122 (round 6): xr_sum_partials, xr_set_mlp_range_word, xr_ema_update_bitfield, strided linear kernels, the scatter on one stream.  121: xr_nerf_mlp_fwd / _bwd take their arithmetic as an argument; small merges.  120 (round 5): one generation per entry point (xr_rays_sampler, xr_hashgrid_fwd / _bwd, xr_composite_train,
                                                   // xr_live_rows, xr_generate_grid_samples, xr_clip_numsteps take their newest signatures), window march, loop without a handle

extern "C" void xr_pcg32_host_state(uint64_t seed, uint64_t ncalls, uint64_t* state_host, uint64_t* inc_host) {
    xr_pcg32 r; r.seed(seed, 1u);
    r.advance(ncalls << 32);      // (one advance by the summed distance: the generator is a group action, 2^64 its period)
    *state_host = r.state; *inc_host = r.inc;
}

// ------------------------------------------------------------------ ray generation
// get_rays_np_hash (/root/reference/xrnerf/datasets/load_data/get_rays.py:35-69) in fp32:
// pixel centre +0.5, dir = ((i-cx)/fx, (j-cy)/fy, 1), d = R*dir, normalise, o = translation.
struct Pose43 { float m[12]; };
__global__ __launch_bounds__(256) void k_gen_rays(Pose43 p, int W, float fx, float fy, float cx, float cy, int row0,
                                                   uint32_t n, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int r = q / W, c = q % W;
    const float i = (float)c + 0.5f, j = (float)(row0 + r) + 0.5f;
    const float dx = (i - cx) / fx, dy = (j - cy) / fy;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = __fmul_rn(p.m[k], dx);
        a = __fadd_rn(a, __fmul_rn(p.m[3 + k], dy));
        v[k] = __fadd_rn(a, p.m[6 + k]);
    }
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
#pragma unroll
    for (int k = 0; k < 3; ++k) { rays_d[3 * (size_t)q + k] = v[k] / nrm; rays_o[3 * (size_t)q + k] = p.m[9 + k]; }
}
extern "C" int xr_gen_rays(const float* pose43_host, int H, int W, float fx, float fy, float cx, float cy, int row0,
                           int nrows, float* rays_o, float* rays_d, void* stream_) {
    XR_REQUIRE(pose43_host && rays_o && rays_d, "null pointer");
    XR_REQUIRE(H > 0 && W > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= H, "bad image window");
    Pose43 p; memcpy(p.m, pose43_host, sizeof(p.m));
    const uint32_t n = (uint32_t)nrows * (uint32_t)W;
    hipLaunchKernelGGL(k_gen_rays, dim3(xr_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream_, p, W, fx, fy, cx, cy, row0, n,
                       rays_o, rays_d);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ scale * HuberLoss(sum) and its gradient
// (+ optionally the alpha-masked squared error the reference logs as PSNR, networks/hashnerf.py:40-42)
__global__ __launch_bounds__(256) void k_huber(const float* __restrict__ rgb, const float* __restrict__ target,
                                                const float* __restrict__ alpha, uint32_t n, float delta, float scale,
                                                float* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float ws[4], ws2[4];
    float acc = 0.f, mse = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float r = rgb[i] - target[i], a = fabsf(r);
        // HuberLoss of the reference (utils/metrics.py:8-16): rel > delta ? rel - delta/2 : 0.5/delta*rel^2
        if (a > delta) { acc += a - 0.5f * delta; grad[i] = scale * (r > 0.f ? 1.f : -1.f); }
        else { acc += 0.5f / delta * a * a; grad[i] = scale * (r / delta); }
        if (alpha) { const float m = r * alpha[i / 3]; mse += m * m; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); mse += __shfl_xor(mse, d, 64); }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = acc; ws2[threadIdx.x >> 6] = mse; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss, scale * ((ws[0] + ws[1]) + (ws[2] + ws[3])));
        if (alpha) atomicAdd(loss + 1, (ws2[0] + ws2[1]) + (ws2[2] + ws2[3]));
    }
}
extern "C" int xr_huber_loss_grad(const float* rgb, const float* target, const float* alpha, uint32_t n_elems, float delta, float scale,
                                  float* grad, float* loss_out, void* stream_) {
    XR_REQUIRE(rgb && target && grad && loss_out && n_elems > 0, "bad argument");
    XR_REQUIRE(!alpha || n_elems % 3 == 0, "the masked error takes rgb triples");
    hipLaunchKernelGGL(k_huber, dim3(min(xr_div_up(n_elems, 256), 1024u)), dim3(256), 0, (hipStream_t)stream_, rgb, target, alpha, n_elems,
                       delta, scale, grad, loss_out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ training batch assembly
// HashBatchSample + RandomBGColor (/root/reference/xrnerf/datasets/pipelines/create.py:153-191,
// augment.py:290-317) on the device in one launch: slices `n` rows of the [N,11] ray table
// (o3, d3, rgba4, img_id) and draws the random background with PCG32 (stream position = row).
// blockIdx.y = batch c of a series (xr_make_batch_series): table rows from row0[c], the generator of call index (first + c)
// (it moves on by 2^32 per call: xr_pcg32_host_state), outputs at c * ray_stride rows.
struct XrBatchRows { uint64_t row0[XR_NGP_WINDOW]; };
__global__ __launch_bounds__(256) void k_make_batch(const float* __restrict__ rows, uint32_t n, xr_pcg32 rng,
                                                     float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                     float* __restrict__ target, float* __restrict__ alpha,
                                                     float* __restrict__ bg, int32_t* __restrict__ img_ids, XrBatchRows at, uint32_t ray_stride) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* r = rows + 11 * ((size_t)at.row0[blockIdx.y] + i);
    rng.advance(((uint64_t)blockIdx.y << 32) + 3ull * i);
    const size_t q = (size_t)blockIdx.y * ray_stride + i;
    const float a = r[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[3 * q + k] = r[k];
        rays_d[3 * q + k] = r[3 + k];
        const float b = rng.next_float();
        bg[3 * q + k] = b;
        target[3 * q + k] = r[6 + k] * a + b * (1.f - a);
    }
    alpha[q] = a;
    img_ids[q] = (int32_t)r[10];
}
// n_series <= XR_NGP_WINDOW batches in one launch: batch c = rows [row0[c], row0[c] + n) of the table, drawn with the generator of call index
// (first + c) (rng_state / rng_inc = xr_pcg32_host_state(seed, first)), written at rows c * ray_stride of the six outputs
extern "C" int xr_make_batch_series(const float* rays_rgb_rows, const uint64_t* row0_host, uint32_t n, uint32_t n_series, uint32_t ray_stride,
                                    uint64_t rng_state, uint64_t rng_inc, float* rays_o, float* rays_d, float* target, float* alpha, float* bg,
                                    int32_t* img_ids, void* stream_) {
    XR_REQUIRE(rays_rgb_rows && row0_host && rays_o && rays_d && target && alpha && bg && img_ids && n > 0, "bad argument");
    XR_REQUIRE(n_series >= 1 && n_series <= (uint32_t)XR_NGP_WINDOW && ray_stride >= n, "bad series");
    xr_pcg32 rng{rng_state, rng_inc};
    XrBatchRows at = {};
    for (uint32_t c = 0; c < n_series; ++c) at.row0[c] = row0_host[c];
    hipLaunchKernelGGL(k_make_batch, dim3(xr_div_up(n, 256), n_series), dim3(256), 0, (hipStream_t)stream_, rays_rgb_rows, n, rng, rays_o,
                       rays_d, target, alpha, bg, img_ids, at, ray_stride);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ fused Adam (+L2 weight decay, + optional EMA)
// torch.optim.Adam semantics (adam1, xr_adam.h); one pass over p, g, m, v (, ema): 16 B per lane per stream.
// up to 4 parameter tensors in ONE launch (the three tensors of HashNerfMLP: 12.2 M + 3 K + 7 K floats):
// block ranges are assigned proportionally, every tensor gets at least one block
struct AdamTensors { float* p[4]; const float* g[4]; float* m[4]; float* v[4]; float* ema[4]; unsigned long long n[4]; unsigned first_block[5]; };
// NT: the optimiser states (m, v, EMA copy) and the gradient -- 390 of the 439 MB this launch moves, each byte touched once per
// step -- go through non-temporal loads / stores so that the parameters, which the next step's gather reads right away, are
// what the caches keep (build with -DXR_ADAM_NT=0 to switch it off -- tools/build_variant.sh; measured in the training loop, profiles/r03_adam_nontemporal.txt:
// 0.534 -> 0.520 ms per step: the gather 93 -> 89 us, the scatter 113 -> 108 us, this launch 71 -> 70 us)
typedef float am_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 am_ld(const float* p, size_t i) {
    if (NT) { const am_f4 r = __builtin_nontemporal_load(reinterpret_cast<const am_f4*>(p) + i); return make_float4(r.x, r.y, r.z, r.w); }
    return reinterpret_cast<const float4*>(p)[i];
}
template <bool NT> __device__ __forceinline__ void am_st(float* p, size_t i, const float4 v) {
    if (NT) { am_f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; __builtin_nontemporal_store(r, reinterpret_cast<am_f4*>(p) + i); }
    else reinterpret_cast<float4*>(p)[i] = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_adam_multi(AdamTensors t, int nt, float b1, float b2, float step_size, float bc2s,
                                                     float eps, float wd, float mom, float gs) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) if (j < nt && blockIdx.x >= t.first_block[j]) k = j;
    const unsigned nblk = t.first_block[k + 1] - t.first_block[k], blk = blockIdx.x - t.first_block[k];
    float* __restrict__ p = t.p[k]; const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k]; float* __restrict__ ema = t.ema[k];
    const size_t n = t.n[k], n4 = n / 4;
    for (size_t i = blk * 256ull + threadIdx.x; i < n4; i += (size_t)nblk * 256) {
        float4 pp = ((float4*)p)[i], mm = am_ld<NT>(m, i), vv = am_ld<NT>(v, i);
        const float4 gg = am_ld<NT>(g, i);
        adam1(pp.x, gg.x, mm.x, vv.x, b1, b2, step_size, bc2s, eps, wd, gs);
        adam1(pp.y, gg.y, mm.y, vv.y, b1, b2, step_size, bc2s, eps, wd, gs);
        adam1(pp.z, gg.z, mm.z, vv.z, b1, b2, step_size, bc2s, eps, wd, gs);
        adam1(pp.w, gg.w, mm.w, vv.w, b1, b2, step_size, bc2s, eps, wd, gs);
        ((float4*)p)[i] = pp; am_st<NT>(m, i, mm); am_st<NT>(v, i, vv);
        if (ema) {
            float4 e = am_ld<NT>(ema, i);
            e.x = (1.f - mom) * e.x + mom * pp.x; e.y = (1.f - mom) * e.y + mom * pp.y;
            e.z = (1.f - mom) * e.z + mom * pp.z; e.w = (1.f - mom) * e.w + mom * pp.w;
            am_st<NT>(ema, i, e);
        }
    }
    if (blk == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i];
        adam1(pp, g[i], mm, vv, b1, b2, step_size, bc2s, eps, wd, gs);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (ema) ema[i] = (1.f - mom) * ema[i] + mom * pp;
    }
}
extern "C" int xr_adam_step_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                  float* const* ema, const size_t* n, int step, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, float ema_momentum, float grad_scale, void* stream_) {
    XR_REQUIRE(n_tensors >= 1 && n_tensors <= 4 && p && g && m && v && n && step >= 1, "bad argument");
    AdamTensors t; memset(&t, 0, sizeof(t));
    unsigned blocks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        XR_REQUIRE(p[k] && g[k] && m[k] && v[k] && n[k] > 0, "null tensor");
        XR_REQUIRE((((uintptr_t)p[k] | (uintptr_t)g[k] | (uintptr_t)m[k] | (uintptr_t)v[k] | (uintptr_t)(ema ? ema[k] : nullptr)) & 15) == 0,
                   "buffers must be 16-byte aligned");
        t.p[k] = p[k]; t.g[k] = g[k]; t.m[k] = m[k]; t.v[k] = v[k]; t.ema[k] = ema ? ema[k] : nullptr; t.n[k] = n[k];
        t.first_block[k] = blocks;
        blocks += min(xr_div_up(n[k] / 4 + 1, 256), 2048u);
    }
    for (int k = n_tensors; k <= 4; ++k) t.first_block[k] = blocks;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
#ifndef XR_ADAM_NT
#define XR_ADAM_NT 1
#endif
    hipLaunchKernelGGL(k_adam_multi<XR_ADAM_NT != 0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, t, n_tensors, beta1, beta2, lr / bc1,
                       sqrtf(bc2), eps, weight_decay, ema_momentum, grad_scale);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// gradients *= (*scale_dev) * host_factor for up to 4 tensors in one launch; a factor of exactly 1 (what
// loss.backward() hands to a fused-step autograd node on one GPU) returns without touching memory
struct ScaleTensors { float* p[4]; unsigned long long n[4]; };
__global__ __launch_bounds__(256) void k_scale_multi(ScaleTensors t, int nt, const float* __restrict__ scale_dev, float host_factor) {
    const float f = (scale_dev ? *scale_dev : 1.f) * host_factor;
    if (f == 1.f) return;
    for (int k = 0; k < nt; ++k) {
        float* __restrict__ p = t.p[k];
        const size_t n = t.n[k], n4 = n / 4;
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            float4 v = ((float4*)p)[i];
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            ((float4*)p)[i] = v;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] *= f;
    }
}
extern "C" int xr_scale_multi(int n_tensors, float* const* p, const size_t* n, const float* scale_dev, float host_factor, void* stream_) {
    XR_REQUIRE(n_tensors >= 1 && n_tensors <= 4 && p && n, "bad argument");
    ScaleTensors t; memset(&t, 0, sizeof(t));
    for (int k = 0; k < n_tensors; ++k) {
        XR_REQUIRE(p[k] && ((uintptr_t)p[k] & 15) == 0, "tensors must be 16-byte aligned");
        t.p[k] = p[k]; t.n[k] = n[k];
    }
    hipLaunchKernelGGL(k_scale_multi, dim3(2048), dim3(256), 0, (hipStream_t)stream_, t, n_tensors, scale_dev, host_factor);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

