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

// ------------------------------------------------------------------ K9 + K10 + K11 in three launches (round 6)
// The refresh's tail used to be eleven launches of 5-14 us each over data that streams in ~20 us (K9, K10 partial, K10 final, K11 bits,
// seven K11 pools).  Same values, bit for bit, from three:
//   1. k9_ema_mean: K9 over the cascades in use and, where it walks cascade 0, K10's partial sums in K10's own order (MEAN_BLOCKS
//      blocks, the grid-stride of k10_partial, the same expression on the UPDATED value);
//   2. k11_bits_own: every workgroup folds the MEAN_BLOCKS partials itself, in k10_final's order (wave butterflies over 64 partials,
//      then the 8 wave sums serially) -- 2 KB out of the L2 instead of a launch; bits of all cascades as k11_bits; and, from a wave
//      ballot over its 64 bytes, the bytes a cascade's OWN bits pool to (`own pool`, what k11_pool would compute if nothing had been
//      pooled into that cascade from below) into a scratch array S[cascade + 1][32768];
//   3. k11_pool_chain: what reaches cascade L from below is C_L = S[L] | T(C_{L-1}) (C_1 = S[1]), where T pools the 32^3 bytes of
//      C_{L-1} -- embedded at the centre [16, 48)^3 of cascade L - 1's 64^3 byte grid -- into the 16^3 parent bytes at [8, 24)^3:
//      parent morton(X, Y, Z) reads the 8 consecutive bytes 8 m .. 8 m + 7 of C_{L-1}, m = morton(X - 8, Y - 8, Z - 8).  The chain is
//      4 096 eight-byte reads per cascade, so every workgroup recomputes it in LDS up to its own cascade (no inter-workgroup order)
//      and then ORs its 4 096-byte slice of C_L into the bitfield at morton(x + 16, y + 16, z + 16) -- exactly the bytes k11_pool's
//      seven dependent launches wrote.
#define POOL_BYTES (XR_GRID_CELLS / 64)                  // 32768 parent bytes per cascade
__global__ __launch_bounds__(GR_BLOCK) void k9_ema_mean(uint32_t n4, float decay, float4* __restrict__ grid, const float4* __restrict__ tmp,
                                                         float* __restrict__ partial) {
    __shared__ float ws[GR_BLOCK / 64];
    const uint32_t n4_0 = XR_GRID_CELLS / 4;
    const float inv = 1.0f / (float)XR_GRID_CELLS;
    float s = 0.f;
    for (uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x; i < max(n4, n4_0); i += MEAN_BLOCKS * GR_BLOCK) {
        float4 p = grid[i];
        if (i < n4) {                                    // K9 (the cascades in use)
            const float4 t = tmp[i];
            p.x = p.x < 0.f ? p.x : fmaxf(p.x * decay, t.x);
            p.y = p.y < 0.f ? p.y : fmaxf(p.y * decay, t.y);
            p.z = p.z < 0.f ? p.z : fmaxf(p.z * decay, t.z);
            p.w = p.w < 0.f ? p.w : fmaxf(p.w * decay, t.w);
            grid[i] = p;
        }
        if (i < n4_0) s += fmaxf(p.x, 0.f) * inv + fmaxf(p.y, 0.f) * inv + fmaxf(p.z, 0.f) * inv + fmaxf(p.w, 0.f) * inv;   // K10 partial (cascade 0)
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
// the mean from the partials, in k10_final's order, by a 256-thread workgroup: wave w folds partials [64 w, 64 w + 64) and [256 + 64 w, ..)
__device__ inline float mean_from_partials(const float* __restrict__ partial, float* ws /*[MEAN_BLOCKS / 64]*/) {
    static_assert(MEAN_BLOCKS == 2 * GR_BLOCK, "two partials per thread");
    const float a = wave_sum(partial[threadIdx.x]), b = wave_sum(partial[GR_BLOCK + threadIdx.x]);
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = a; ws[GR_BLOCK / 64 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < MEAN_BLOCKS / 64; ++w) t += ws[w];
    return t;
}
__global__ __launch_bounds__(GR_BLOCK) void k11_bits_own(uint32_t n_bytes, const float4* __restrict__ grid, uint8_t* __restrict__ bitfield,
                                                          const float* __restrict__ partial, const float* __restrict__ mean_in,
                                                          float* __restrict__ mean_out, uint8_t* __restrict__ own_pool /*[CASCADES - 1][POOL_BYTES]*/) {
    __shared__ float ws[MEAN_BLOCKS / 64];
    float m;
    if (partial != nullptr) {
        m = mean_from_partials(partial, ws);
        if (blockIdx.x == 0 && threadIdx.x == 0) mean_out[0] = m;
    } else m = mean_in[0];
    const uint32_t i = blockIdx.x * GR_BLOCK + threadIdx.x;                // n_bytes is a multiple of the block: no ragged wave
    const float thresh = 0.01f < m ? 0.01f : m;                                        // :35
    const float4 a = grid[2 * (size_t)i], b = grid[2 * (size_t)i + 1];
    const uint32_t bits = (a.x > thresh ? 1u : 0u) | (a.y > thresh ? 2u : 0u) | (a.z > thresh ? 4u : 0u) | (a.w > thresh ? 8u : 0u) |
                          (b.x > thresh ? 16u : 0u) | (b.y > thresh ? 32u : 0u) | (b.z > thresh ? 64u : 0u) | (b.w > thresh ? 128u : 0u);
    bitfield[i] = (uint8_t)bits;
    // bit k of parent byte q = child byte 8 q + k is not zero: the wave's 64 bytes are 8 parents, its ballot their 8 bytes in order
    const unsigned long long any = __ballot(bits != 0u);
    const uint32_t level = i / (XR_GRID_CELLS / 8), j = i % (XR_GRID_CELLS / 8);
    if ((threadIdx.x & 63) == 0 && level + 1 < XR_NERF_CASCADES)
        *reinterpret_cast<uint2*>(own_pool + (size_t)level * POOL_BYTES + j / 8) = make_uint2((uint32_t)any, (uint32_t)(any >> 32));
}
#define CHAIN_THREADS 1024
#define CHAIN_SLICES 8                                   // workgroups per cascade, POOL_BYTES / CHAIN_SLICES bytes each
__global__ __launch_bounds__(CHAIN_THREADS) void k11_pool_chain(const uint8_t* __restrict__ own_pool, uint8_t* __restrict__ bitfield) {
    __shared__ __attribute__((aligned(16))) uint8_t c[2][POOL_BYTES];
    const uint32_t level = 1 + blockIdx.x / CHAIN_SLICES, slice = blockIdx.x % CHAIN_SLICES;      // cascade 1 .. 7
    static_assert(POOL_BYTES == CHAIN_THREADS * 32, "32 bytes per thread");
    auto load = [&](uint32_t l, uint8_t* dst) {          // S[l] (own pool of cascade l - 1) -> LDS
        const uint4* src = reinterpret_cast<const uint4*>(own_pool + (size_t)(l - 1) * POOL_BYTES);
        reinterpret_cast<uint4*>(dst)[threadIdx.x] = src[threadIdx.x];
        reinterpret_cast<uint4*>(dst)[CHAIN_THREADS + threadIdx.x] = src[CHAIN_THREADS + threadIdx.x];
    };
    load(1, c[0]);
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t l = 2; l <= level; ++l) {
        load(l, c[cur ^ 1]);
        __syncthreads();
        for (uint32_t mm = threadIdx.x; mm < POOL_BYTES / 8; mm += CHAIN_THREADS) {
            const uint2 p = reinterpret_cast<const uint2*>(c[cur])[mm];
            const uint32_t bits = ((p.x & 0xffu) ? 1u : 0u) | ((p.x & 0xff00u) ? 2u : 0u) | ((p.x & 0xff0000u) ? 4u : 0u) |
                                  ((p.x & 0xff000000u) ? 8u : 0u) | ((p.y & 0xffu) ? 16u : 0u) | ((p.y & 0xff00u) ? 32u : 0u) |
                                  ((p.y & 0xff0000u) ? 64u : 0u) | ((p.y & 0xff000000u) ? 128u : 0u);
            const uint32_t x = xr_morton3d_invert(mm >> 0) + 8, y = xr_morton3d_invert(mm >> 1) + 8, z = xr_morton3d_invert(mm >> 2) + 8;
            if (bits) c[cur ^ 1][xr_morton3d(x, y, z)] |= (uint8_t)bits;      // one thread per target byte
        }
        __syncthreads();
        cur ^= 1;
    }
    uint8_t* __restrict__ dst = bitfield + (size_t)XR_GRID_CELLS * level / 8;
    for (uint32_t i = slice * (POOL_BYTES / CHAIN_SLICES) + threadIdx.x; i < (slice + 1) * (POOL_BYTES / CHAIN_SLICES); i += CHAIN_THREADS) {
        const uint32_t bits = c[cur][i];
        if (bits == 0u) continue;
        const uint32_t x = xr_morton3d_invert(i >> 0) + 16, y = xr_morton3d_invert(i >> 1) + 16, z = xr_morton3d_invert(i >> 2) + 16;
        dst[xr_morton3d(x, y, z)] |= (uint8_t)bits;      // exactly one thread per parent byte (:69)
    }
}

extern "C" size_t xr_update_bitfield_workspace_bytes(void) { return MEAN_BLOCKS * sizeof(float) + (size_t)(XR_NERF_CASCADES - 1) * POOL_BYTES; }

static int launch_bits_and_chain(const float* grid, const float* partial, const float* mean_in, float* mean_out, uint8_t* bitfield, void* workspace,
                                 hipStream_t stream) {
    const uint32_t n_bytes = XR_GRID_CELLS / 8 * XR_NERF_CASCADES;
    uint8_t* own_pool = (uint8_t*)workspace + MEAN_BLOCKS * sizeof(float);
    static_assert((XR_GRID_CELLS / 8 * XR_NERF_CASCADES) % GR_BLOCK == 0, "whole workgroups");
    hipLaunchKernelGGL(k11_bits_own, dim3(n_bytes / GR_BLOCK), dim3(GR_BLOCK), 0, stream, n_bytes, (const float4*)grid, bitfield, partial, mean_in, mean_out,
                       own_pool);
    hipLaunchKernelGGL(k11_pool_chain, dim3((XR_NERF_CASCADES - 1) * CHAIN_SLICES), dim3(CHAIN_THREADS), 0, stream, (const uint8_t*)own_pool, bitfield);
    return XR_OK;
}

extern "C" int xr_update_bitfield(const float* density_grid, float* density_grid_mean, uint8_t* bitfield,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(density_grid && density_grid_mean && bitfield, "null pointer");
    XR_REQUIRE(((uintptr_t)density_grid & 15) == 0 && ((uintptr_t)bitfield & 7) == 0, "misaligned grid / bitfield");
    XR_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= xr_update_bitfield_workspace_bytes(), "workspace too small / misaligned");
    hipLaunchKernelGGL(k10_partial, dim3(MEAN_BLOCKS), dim3(GR_BLOCK), 0, stream, (const float4*)density_grid, (float*)workspace);
    launch_bits_and_chain(density_grid, (const float*)workspace, nullptr, density_grid_mean, bitfield, workspace, stream);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

/* K9 + K10 + K11 of one refresh in three launches (the sampler's refresh tail): xr_ema_grid_samples followed by xr_update_bitfield, same
 * grid, mean and bitfield bit for bit */
extern "C" int xr_ema_update_bitfield(const float* density_grid_tmp, uint32_t n_elements, float decay, float* density_grid,
                                      float* density_grid_mean, uint8_t* bitfield, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(density_grid_tmp && density_grid && density_grid_mean && bitfield, "null pointer");
    XR_REQUIRE(n_elements % 4 == 0 && n_elements <= XR_GRID_CELLS * XR_NERF_CASCADES &&
               (((uintptr_t)density_grid_tmp | (uintptr_t)density_grid) & 15) == 0 && ((uintptr_t)bitfield & 7) == 0,
               "grids must be 16-byte aligned multiples of 4 elements");
    XR_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= xr_update_bitfield_workspace_bytes(), "workspace too small / misaligned");
    hipLaunchKernelGGL(k9_ema_mean, dim3(MEAN_BLOCKS), dim3(GR_BLOCK), 0, stream, n_elements / 4, decay, (float4*)density_grid,
                       (const float4*)density_grid_tmp, (float*)workspace);
    launch_bits_and_chain(density_grid, (const float*)workspace, nullptr, density_grid_mean, bitfield, workspace, stream);
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
