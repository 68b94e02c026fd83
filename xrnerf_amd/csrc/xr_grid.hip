// K6..K11: occupancy-grid maintenance (sample cells -> splat -> EMA-max -> mean -> bitfield + mips).
// Streaming, HBM-bound integer/fp32 work; compiled with -ffp-contract=off (K6/K11 make index
// decisions).  Reference: /root/reference/extensions/ngp_raymarch/src/
// {generate_grid_samples_nerf_nonuniform,mark_untrained_density_grid,
//  splat_grid_samples_nerf_max_nearest_neighbor,ema_grid_samples_nerf,update_bitfield}.cu
#include "xr_common.h"

#define GR_BLOCK 256

// ------------------------------------------------------------------ K6 (generate_grid_samples...cu:11-41)
__global__ __launch_bounds__(GR_BLOCK) void k6_generate(uint32_t n_elements, xr_pcg32 rng, uint32_t step, float lo,
                                                         float hi, const float* __restrict__ grid,
                                                         float* __restrict__ out, uint32_t out_rs, uint32_t out_cs,
                                                         int32_t* __restrict__ indices, uint32_t n_cascades, float thresh) {
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n_elements) return;
    rng.advance((uint64_t)(i * 4u));
    uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < 10; ++j) {
        idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % XR_GRID_CELLS;
        idx += level * XR_GRID_CELLS;
        if (grid[idx] > thresh) break;
    }
    const uint32_t pidx = idx % XR_GRID_CELLS;
    const float x = (float)xr_morton3d_invert(pidx >> 0), y = (float)xr_morton3d_invert(pidx >> 1),
                z = (float)xr_morton3d_invert(pidx >> 2);
    const float sc = scalbnf(1.0f, (int)level), diag = hi - lo;
    const float u0 = rng.next_float(), u1 = rng.next_float(), u2 = rng.next_float();
    const float px = ((x + u0) / 128.0f - 0.5f) * sc + 0.5f;
    const float py = ((y + u1) / 128.0f - 0.5f) * sc + 0.5f;
    const float pz = ((z + u2) / 128.0f - 0.5f) * sc + 0.5f;
    float* o = out + (size_t)i * out_rs;          // rows [n,3] (out_rs = 3, out_cs = 1) or three planes (out_rs = 1, out_cs = plane size)
    o[0] = (px - lo) / diag; o[out_cs] = (py - lo) / diag; o[2 * (size_t)out_cs] = (pz - lo) / diag;
    indices[i] = (int32_t)idx;
}

extern "C" int xr_generate_grid_samples(const float* density_grid, uint32_t ema_step, uint32_t n_elements,
                                         uint32_t n_cascades, float thresh, float aabb0, float aabb1,
                                         uint64_t rng_state, uint64_t rng_inc, float* positions, uint32_t pos_row_stride,
                                         uint32_t pos_comp_stride, int32_t* indices, void* stream_) {
    if (n_elements == 0) return XR_OK;
    XR_REQUIRE(density_grid && positions && indices, "null pointer");
    XR_REQUIRE(n_cascades >= 1 && n_cascades <= XR_NERF_CASCADES, "n_cascades out of range");
    XR_REQUIRE(pos_row_stride >= 1 && pos_comp_stride >= 1, "bad stride");
    xr_pcg32 rng{rng_state, rng_inc};
    hipLaunchKernelGGL(k6_generate, dim3(xr_div_up(n_elements, GR_BLOCK)), dim3(GR_BLOCK), 0, (hipStream_t)stream_,
                       n_elements, rng, ema_step, aabb0, aabb1, density_grid, positions, pos_row_stride, pos_comp_stride, indices,
                       n_cascades, thresh);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K7 (mark_untrained_density_grid.cu:6-52)
// Camera frames are staged in LDS once per block (n_img * 14 floats) instead of being re-read
// from global memory by each of the 16.8 M cells.
__global__ __launch_bounds__(GR_BLOCK) void k7_mark(uint32_t n_elements, float* __restrict__ grid, uint32_t n_img,
                                                     const float* __restrict__ focal, const float* __restrict__ xforms,
                                                     int res0, int res1) {
    extern __shared__ __attribute__((aligned(16))) float cam[];   // [n_img][14]: 12 xform + 2 focal
    for (uint32_t e = threadIdx.x; e < n_img * 14; e += GR_BLOCK) {
        uint32_t j = e / 14, k = e % 14;
        cam[e] = k < 12 ? xforms[12 * j + k] : focal[2 * j + (k - 12)];
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n_elements) return;
    const uint32_t level = i / XR_GRID_CELLS, pidx = i % XR_GRID_CELLS;
    const float sc = scalbnf(1.0f, (int)level);
    const float hx = res0 * 0.5f, hy = res1 * 0.5f;
    const float px = (((float)xr_morton3d_invert(pidx >> 0) + 0.5f) / 128.0f - 0.5f) * sc + 0.5f;
    const float py = (((float)xr_morton3d_invert(pidx >> 1) + 0.5f) / 128.0f - 0.5f) * sc + 0.5f;
    const float pz = (((float)xr_morton3d_invert(pidx >> 2) + 0.5f) / 128.0f - 0.5f) * sc + 0.5f;
    const float radius = 0.5f * XR_SQRT3 * sc / 128.0f;
    bool seen = false;
    for (uint32_t j = 0; j < n_img && !seen; ++j) {
        const float* m = cam + 14 * j;   // Matrix<float,3,4> column major: col c = m[3c..3c+2]
        const float lx = px - m[9], ly = py - m[10], lz = pz - m[11];
        const float x = lx * m[0] + ly * m[1] + lz * m[2];
        const float y = lx * m[3] + ly * m[4] + lz * m[5];
        const float z = lx * m[6] + ly * m[7] + lz * m[8];
        if (z > 0.f && fabsf(x) - radius < z / m[12] * hx && fabsf(y) - radius < z / m[13] * hy) seen = true;
    }
    grid[i] = seen ? 0.f : -1.f;
}

extern "C" int xr_mark_untrained_density_grid(const float* focal_lengths, const float* xforms, uint32_t n_elements,
                                              uint32_t n_images, int resolution0, int resolution1,
                                              float* density_grid, void* stream_) {
    XR_REQUIRE(focal_lengths && xforms && density_grid, "null pointer");
    XR_REQUIRE(n_elements > 0 && n_images > 0, "empty input");
    XR_REQUIRE(n_images * 14 * sizeof(float) <= 64 * 1024, "too many training images for the LDS camera cache");
    hipLaunchKernelGGL(k7_mark, dim3(xr_div_up(n_elements, GR_BLOCK)), dim3(GR_BLOCK), n_images * 14 * sizeof(float),
                       (hipStream_t)stream_, n_elements, density_grid, n_images, focal_lengths, xforms, resolution0,
                       resolution1);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K8 (splat_...cu:7-28)
__global__ __launch_bounds__(GR_BLOCK) void k8_splat(uint32_t n, const int32_t* __restrict__ indices, uint32_t width,
                                                      const float* __restrict__ mlp_out, float* __restrict__ grid_tmp) {
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    // density activation is hard-wired Exponential there (:50); optical thickness at the
    // smallest step.  Positive floats order like their bit patterns -> uint atomicMax.
    const float thick = expf(mlp_out[(size_t)i * width]) * xr_min_step();
    atomicMax((uint32_t*)&grid_tmp[(uint32_t)indices[i]], __float_as_uint(thick));
}
extern "C" int xr_splat_grid_samples(const float* mlp_out, const int32_t* indices, uint32_t padded_output_width,
                                     uint32_t n_samples, float* density_grid_tmp, void* stream_) {
    if (n_samples == 0) return XR_OK;
    XR_REQUIRE(mlp_out && indices && density_grid_tmp && padded_output_width >= 1, "bad argument");
    hipLaunchKernelGGL(k8_splat, dim3(xr_div_up(n_samples, GR_BLOCK)), dim3(GR_BLOCK), 0, (hipStream_t)stream_, n_samples,
                       indices, padded_output_width, mlp_out, density_grid_tmp);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K9 (ema_grid_samples_nerf.cu:4-27)
// pure stream: 16 B per lane per access, grid-stride.
__global__ __launch_bounds__(GR_BLOCK) void k9_ema(uint32_t n4, float decay, float4* __restrict__ grid,
                                                    const float4* __restrict__ tmp) {
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < n4; i += gridDim.x * GR_BLOCK) {
        float4 p = grid[i]; const float4 t = tmp[i];
        p.x = p.x < 0.f ? p.x : fmaxf(p.x * decay, t.x);
        p.y = p.y < 0.f ? p.y : fmaxf(p.y * decay, t.y);
        p.z = p.z < 0.f ? p.z : fmaxf(p.z * decay, t.z);
        p.w = p.w < 0.f ? p.w : fmaxf(p.w * decay, t.w);
        grid[i] = p;
    }
}
extern "C" int xr_ema_grid_samples(const float* density_grid_tmp, uint32_t n_elements, float decay, float* density_grid,
                                   void* stream_) {
    XR_REQUIRE(density_grid_tmp && density_grid, "null pointer");
    XR_REQUIRE(n_elements % 4 == 0 && (((uintptr_t)density_grid_tmp | (uintptr_t)density_grid) & 15) == 0,
               "grids must be 16-byte aligned multiples of 4 elements");
    const uint32_t n4 = n_elements / 4;
    const uint32_t blocks = min(xr_div_up(n4, GR_BLOCK), 2048u);
    hipLaunchKernelGGL(k9_ema, dim3(blocks), dim3(GR_BLOCK), 0, (hipStream_t)stream_, n4, decay, (float4*)density_grid,
                       (const float4*)density_grid_tmp);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K10 mean (update_bitfield.cu:3-22,98-101)
// Fixed-order two-stage tree (wave64 shuffles -> LDS -> one partial per block -> one final
// block): bit-reproducible, unlike the reference's float atomicAdd of block partials.
#define MEAN_BLOCKS 512
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__global__ __launch_bounds__(GR_BLOCK) void k10_partial(const float4* __restrict__ grid, float* __restrict__ partial) {
    __shared__ float ws[GR_BLOCK / 64];
    const uint32_t n4 = XR_GRID_CELLS / 4;
    const float inv = 1.0f / (float)XR_GRID_CELLS;   // exact power of two: v/G^3 == v*inv
    float s = 0.f;
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < n4; i += MEAN_BLOCKS * GR_BLOCK) {
        const float4 v = grid[i];
        s += fmaxf(v.x, 0.f) * inv + fmaxf(v.y, 0.f) * inv + fmaxf(v.z, 0.f) * inv + fmaxf(v.w, 0.f) * inv;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(MEAN_BLOCKS) void k10_final(const float* __restrict__ partial, float* __restrict__ mean) {
    __shared__ float ws[MEAN_BLOCKS / 64];
    float s = wave_sum(partial[threadIdx.x]);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < MEAN_BLOCKS / 64; ++w) t += ws[w];
        mean[0] = t;
    }
}

// ------------------------------------------------------------------ K11 (update_bitfield.cu:24-71)
// one byte per thread = 8 cells = two 16-B loads per lane.
__global__ __launch_bounds__(GR_BLOCK) void k11_bits(uint32_t n_bytes, const float4* __restrict__ grid,
                                                      uint8_t* __restrict__ bitfield, const float* __restrict__ mean) {
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n_bytes) return;
    const float m = mean[0];
    const float thresh = 0.01f < m ? 0.01f : m;                                        // :35
    const float4 a = grid[2 * (size_t)i], b = grid[2 * (size_t)i + 1];
    uint32_t bits = (a.x > thresh ? 1u : 0u) | (a.y > thresh ? 2u : 0u) | (a.z > thresh ? 4u : 0u) | (a.w > thresh ? 8u : 0u) |
                    (b.x > thresh ? 16u : 0u) | (b.y > thresh ? 32u : 0u) | (b.z > thresh ? 64u : 0u) | (b.w > thresh ? 128u : 0u);
    bitfield[i] = (uint8_t)bits;
}
__global__ __launch_bounds__(GR_BLOCK) void k11_pool(uint32_t n, const uint2* __restrict__ prev, uint8_t* __restrict__ next) {
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint2 p = prev[i];                                                            // 8 child bytes
    uint32_t bits = ((p.x & 0xffu) ? 1u : 0u) | ((p.x & 0xff00u) ? 2u : 0u) | ((p.x & 0xff0000u) ? 4u : 0u) |
                    ((p.x & 0xff000000u) ? 8u : 0u) | ((p.y & 0xffu) ? 16u : 0u) | ((p.y & 0xff00u) ? 32u : 0u) |
                    ((p.y & 0xff0000u) ? 64u : 0u) | ((p.y & 0xff000000u) ? 128u : 0u);
    const uint32_t x = xr_morton3d_invert(i >> 0) + 16, y = xr_morton3d_invert(i >> 1) + 16, z = xr_morton3d_invert(i >> 2) + 16;
    next[xr_morton3d(x, y, z)] |= (uint8_t)bits;     // exactly one thread per parent byte (:69)
}

static int launch_bitfield(const float* grid, const float* mean, uint8_t* bitfield, hipStream_t stream) {
    const uint32_t n_bytes = XR_GRID_CELLS / 8 * XR_NERF_CASCADES;
    hipLaunchKernelGGL(k11_bits, dim3(xr_div_up(n_bytes, GR_BLOCK)), dim3(GR_BLOCK), 0, stream, n_bytes,
                       (const float4*)grid, bitfield, mean);
    for (uint32_t level = 1; level < XR_NERF_CASCADES; ++level) {
        hipLaunchKernelGGL(k11_pool, dim3(xr_div_up(XR_GRID_CELLS / 64, GR_BLOCK)), dim3(GR_BLOCK), 0, stream,
                           XR_GRID_CELLS / 64, (const uint2*)(bitfield + (size_t)XR_GRID_CELLS * (level - 1) / 8),
                           bitfield + (size_t)XR_GRID_CELLS * level / 8);
    }
    return XR_OK;
}

extern "C" size_t xr_update_bitfield_workspace_bytes(void) { return MEAN_BLOCKS * sizeof(float); }

extern "C" int xr_update_bitfield(const float* density_grid, float* density_grid_mean, uint8_t* bitfield,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(density_grid && density_grid_mean && bitfield, "null pointer");
    XR_REQUIRE(((uintptr_t)density_grid & 15) == 0 && ((uintptr_t)bitfield & 7) == 0, "misaligned grid / bitfield");
    XR_REQUIRE(workspace && workspace_bytes >= xr_update_bitfield_workspace_bytes(), "workspace too small");
    hipLaunchKernelGGL(k10_partial, dim3(MEAN_BLOCKS), dim3(GR_BLOCK), 0, stream, (const float4*)density_grid, (float*)workspace);
    hipLaunchKernelGGL(k10_final, dim3(1), dim3(MEAN_BLOCKS), 0, stream, (const float*)workspace, density_grid_mean);
    launch_bitfield(density_grid, density_grid_mean, bitfield, stream);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_bitfield_from_mean(const float* density_grid, const float* density_grid_mean, uint8_t* bitfield,
                                     void* stream_) {
    XR_REQUIRE(density_grid && density_grid_mean && bitfield, "null pointer");
    XR_REQUIRE(((uintptr_t)density_grid & 15) == 0 && ((uintptr_t)bitfield & 7) == 0, "misaligned grid / bitfield");
    launch_bitfield(density_grid, density_grid_mean, bitfield, (hipStream_t)stream_);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
