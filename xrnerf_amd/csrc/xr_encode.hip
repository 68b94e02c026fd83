// Multiresolution hash-grid encoding (forward gather / backward scatter-add) and SH-4.
// This is the tiny-cuda-nn surface of /root/reference/xrnerf/models/mlps/hashnerf_mlp.py:34-37,
// rebuilt from the published algorithm -- tcnn itself is not in the
// reference tree.  fp32 tables, fp32 interpolation.
//
// MI355X mapping: one thread per (sample, level).  blockIdx -> (XCD, level slot, sample block)
// so that a given level's table slice is only ever touched from ONE XCD (block b is observed to
// run on XCD b % 8 -- a speed assumption only) and, within the XCD, level-major: an XCD finishes
// one level before it starts the next, so the 4 MiB slice it is gathering from stays in its
// private 4 MiB L2 instead of thrashing over all 16 levels.  Features are written FEATURE-MAJOR ([2L][ld]) so that both these stores and the MLP's
// MFMA operand loads are 256-B coalesced rows.
#include "xr_common.h"
#include <cstdlib>

#define EN_BLOCK 256
#define EN_MAX_LEVELS 16

struct GridMeta {
    float scale[EN_MAX_LEVELS];
    uint32_t res[EN_MAX_LEVELS];
    uint32_t off[EN_MAX_LEVELS + 1];
    int n_levels;
    uint32_t n_sblocks;   // sample blocks per level
    int order;            // 0: levels of an XCD interleaved, 1: level-major within the XCD
};

extern "C" void xr_hashgrid_meta(int n_levels, int log2_hashmap_size, int base_resolution, double per_level_scale,
                                 float* scale, uint32_t* resolution, uint32_t* offset) {
    // tcnn keeps per_level_scale (and its log2) as float
    const float log2b = log2f((float)per_level_scale);
    uint32_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float s = exp2f((float)l * log2b) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(s) + 1u;
        scale[l] = s; resolution[l] = res; offset[l] = off;
        const double cube = (double)res * res * res;
        uint32_t n = cube > 2147483647.0 ? 2147483647u : (uint32_t)cube;
        n = (n + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        off += n;
    }
    offset[n_levels] = off;
}

__device__ inline uint32_t grid_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t hsize, bool hashed) {
    uint32_t index;
    if (hashed) index = cx ^ (cy * 2654435761u) ^ (cz * 805459861u);
    else index = cx + cy * res + cz * res * res;
    return index % hsize;
}
// tcnn's stride loop (`for dim while stride <= hashmap_size`) followed by `if hashmap_size < stride`
// reduces, for 3-D inputs, to: hashed iff res^3 > hashmap_size (computed on the host side of the
// launch in 64-bit, passed as a flag bit per level)
__device__ inline void level_of_block(const GridMeta& gm, uint32_t* level, uint32_t* sblock) {
    const uint32_t per_xcd = (gm.n_levels + 7) / 8;
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (gm.order == 0) { *level = xcd + 8 * (j % per_xcd); *sblock = j / per_xcd; }
    else if (gm.order == 2) {
        // balanced: all XCDs share every level (finest first); used by the scatter, whose atomics
        // showed no sensitivity to L2 residency but a per-level request count that differs 20x
        *level = gm.n_levels - 1 - blockIdx.x / gm.n_sblocks; *sblock = blockIdx.x % gm.n_sblocks;
    } else {
        // level-major: an XCD finishes one level before it starts the next (finest first), so that
        // the one 4 MiB table slice it is working on stays resident in its 4 MiB L2
        const uint32_t slot = j / gm.n_sblocks;
        *level = xcd + 8 * (per_xcd - 1 - slot); *sblock = j % gm.n_sblocks;
    }
}

__global__ __launch_bounds__(EN_BLOCK) void k_hashgrid_fwd(GridMeta gm, uint32_t hashed_mask, const float* __restrict__ table,
                                                            const float* __restrict__ x, uint32_t x_stride, uint32_t n,
                                                            const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                            float* __restrict__ enc_t, uint32_t ld) {
    uint32_t l, sb;
    level_of_block(gm, &l, &sb);
    if (l >= (uint32_t)gm.n_levels) return;
    if (n_dev) n = min(n, *n_dev);
    const uint32_t i = sb * EN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float scale = gm.scale[l];
    const uint32_t res = gm.res[l], hsize = gm.off[l + 1] - gm.off[l];
    const bool hashed = (hashed_mask >> l) & 1;
    const float2* __restrict__ tab = (const float2*)table + gm.off[l];
    const float* xp = x + (size_t)(rows ? rows[i] : i) * x_stride;    // optional row indirection (render slices)
    float w[3]; uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = xp[d] * scale + 0.5f;
        const float f = floorf(p);
        g[d] = (uint32_t)(int)f; w[d] = p - f;
    }
    // The two x-neighbours of a (y,z) corner pair are adjacent table entries whenever their indices differ
    // only in bit 0 (dense levels with an even index, hashed levels with an even x: idx ^ 1): one 16-B
    // load then serves both corners -- 6 instead of 8 cache-line lookups per sample-level on average.
    float v0x[4], v0y[4], v1x[4], v1y[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t gy = g[1] + (p & 1), gz = g[2] + (p >> 1);
        const uint32_t i0 = grid_index(g[0], gy, gz, res, hsize, hashed), i1 = grid_index(g[0] + 1, gy, gz, res, hsize, hashed);
        if ((i0 ^ i1) == 1u) {
            const float4 q = *reinterpret_cast<const float4*>(tab + (i0 & ~1u));
            const bool odd = i0 & 1u;
            v0x[p] = odd ? q.z : q.x; v0y[p] = odd ? q.w : q.y;
            v1x[p] = odd ? q.x : q.z; v1y[p] = odd ? q.y : q.w;
        } else {
            const float2 a = tab[i0], b = tab[i1];
            v0x[p] = a.x; v0y[p] = a.y; v1x[p] = b.x; v1y[p] = b.y;
        }
    }
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {        // same accumulation order as the oracle: corner 0..7, x fastest
        const float wt = ((c & 1) ? w[0] : 1.f - w[0]) * ((c & 2) ? w[1] : 1.f - w[1]) * ((c & 4) ? w[2] : 1.f - w[2]);
        const int p = c >> 1;
        const float vx = (c & 1) ? v1x[p] : v0x[p], vy = (c & 1) ? v1y[p] : v0y[p];
        r0 += wt * vx; r1 += wt * vy;
    }
    enc_t[(size_t)(2 * l) * ld + i] = r0;
    enc_t[(size_t)(2 * l + 1) * ld + i] = r1;
}

// Scatter-add of the feature gradients.
//
// Measured on MI355X (tools/atomic_probe.hip): scattered global atomics retire at ~21-24 G
// REQUESTS/s chip-wide whatever the type (f32/u64/f64/pk_f16), the footprint (32 KB .. 128 MB) or
// the XCD partitioning -- but lanes of ONE wave instruction that hit adjacent dwords of the same
// line are merged into one request (pairs 42, quads 84, 16-lane lines 333 G atomics/s).  The
// kernel is therefore bound by the number of atomic REQUESTS, and is organised around that:
//   * 16 lanes own the 16 dwords {cz,cy,cx,f} of one (sample stream, level): the four lanes
//     {cx=0,1} x {f=0,1} of a (cy,cz) corner pair address 16 contiguous bytes for dense levels and,
//     for hashed levels, idx ^ (x ^ (x+1)) -- the same 64-B line 7 times out of 8 -- so one
//     instruction issues 4 requests per sample-level instead of 16;
//   * each 16-lane group walks BW_CH CONSECUTIVE samples (ray order) and keeps the running sum of
//     its dword in a register while the cell does not change: at the coarse levels a whole ray
//     segment collapses into one flush (run-length reduction with no shuffles, any run length).
#define BW_CH 32
#define BW_SAMPLES_PER_BLOCK (BW_CH * (EN_BLOCK / 16))
__global__ __launch_bounds__(EN_BLOCK) void k_hashgrid_bwd(GridMeta gm, uint32_t hashed_mask, const float* __restrict__ x,
                                                            uint32_t x_stride, const float* __restrict__ denc_t, uint32_t ld,
                                                            uint32_t n, const uint32_t* __restrict__ n_dev, float* __restrict__ grad_table) {
    constexpr uint32_t bw_ch = BW_CH;   // compile-time: a runtime chunk length costs 8 % (loop not unrolled)
    uint32_t l, sb;
    level_of_block(gm, &l, &sb);
    if (l >= (uint32_t)gm.n_levels) return;
    if (n_dev) n = min(n, *n_dev);
    const uint32_t q = threadIdx.x & 15, group = threadIdx.x >> 4;
    const uint32_t f = q & 1, cx = (q >> 1) & 1, cy = (q >> 2) & 1, cz = (q >> 3) & 1;
    const uint32_t b0 = sb * BW_SAMPLES_PER_BLOCK;
    if (b0 >= n) return;                                     // uniform for the block
    const uint32_t bn = min((uint32_t)BW_SAMPLES_PER_BLOCK, n - b0);
    // Stage the block's positions and feature gradients in LDS with coalesced loads.  The walk below
    // then touches global memory ONLY through atomics: LDS reads count on lgkmcnt, so no vector-memory
    // wait ever sits between two atomic issues (a global load per sample would order every atomic
    // behind it -- vmcnt retires in order -- and expose the full atomic latency per sample).
    __shared__ float s_x[BW_SAMPLES_PER_BLOCK * 3];
    __shared__ float s_d[2][BW_SAMPLES_PER_BLOCK];
    for (uint32_t e = threadIdx.x; e < bn * 3; e += EN_BLOCK) {
        const uint32_t i = e / 3, k = e - 3 * i;
        s_x[e] = x[(size_t)(b0 + i) * x_stride + k];
    }
    for (uint32_t e = threadIdx.x; e < bn; e += EN_BLOCK) {
        s_d[0][e] = denc_t[(size_t)(2 * l) * ld + b0 + e];
        s_d[1][e] = denc_t[(size_t)(2 * l + 1) * ld + b0 + e];
    }
    __syncthreads();
    const uint32_t i0 = group * bw_ch;
    if (i0 >= bn) return;
    const uint32_t i1 = min(i0 + bw_ch, bn);
    const float scale = gm.scale[l];
    const uint32_t res = gm.res[l], hsize = gm.off[l + 1] - gm.off[l];
    const bool hashed = (hashed_mask >> l) & 1;
    float* __restrict__ tab = grad_table + 2 * (size_t)gm.off[l] + f;
    uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu, c2 = 0xffffffffu;   // current cell
    float acc = 0.f;
    float* dst = tab;
    for (uint32_t i = i0; i < i1; ++i) {
        const float d = s_d[f][i];
        const float p0 = s_x[3 * i] * scale + 0.5f, p1 = s_x[3 * i + 1] * scale + 0.5f, p2 = s_x[3 * i + 2] * scale + 0.5f;
        const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
        const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
        const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
        const float wt = (cx ? w0 : 1.f - w0) * (cy ? w1 : 1.f - w1) * (cz ? w2 : 1.f - w2);
        const float contrib = wt * d;
        if (g0 == c0 && g1 == c1 && g2 == c2) {
            acc += contrib;
        } else {
            if (acc != 0.f) unsafeAtomicAdd(dst, acc);       // hardware fp32 atomic add, no return
            c0 = g0; c1 = g1; c2 = g2;
            dst = tab + 2 * (size_t)grid_index(g0 + cx, g1 + cy, g2 + cz, res, hsize, hashed);
            acc = contrib;
        }
    }
    if (acc != 0.f) unsafeAtomicAdd(dst, acc);
}

static int fill_meta(GridMeta* gm, uint32_t* hashed_mask, int n_levels, const float* scale, const uint32_t* res,
                     const uint32_t* off) {
    if (n_levels < 1 || n_levels > EN_MAX_LEVELS || !scale || !res || !off) return -1;
    gm->n_levels = n_levels;
    *hashed_mask = 0;
    for (int l = 0; l < n_levels; ++l) {
        gm->scale[l] = scale[l]; gm->res[l] = res[l]; gm->off[l] = off[l];
        const uint64_t hsize = off[l + 1] - off[l];
        // tcnn: stride accumulates while stride <= hsize; hashed iff hsize < final stride
        uint64_t stride = 1;
        for (int d = 0; d < 3 && stride <= hsize; ++d) stride *= res[l];
        if (hsize < stride) *hashed_mask |= 1u << l;
    }
    gm->off[n_levels] = off[n_levels];
    gm->order = 1;   // measured: forward gather 0.154 -> 0.115 ms at 2^18 samples
    return 0;
}

extern "C" int xr_hashgrid_fwd(const float* table, const float* x, uint32_t x_stride, uint32_t n, const uint32_t* n_dev,
                               const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                               const uint32_t* offset_host, float* enc_t, uint32_t ld, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(table && x && enc_t, "null pointer");
    XR_REQUIRE(x_stride >= 3 && ld >= n, "bad stride");
    XR_REQUIRE(((uintptr_t)table & 15) == 0, "table must be 16-byte aligned");
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    const uint32_t per_xcd = (n_levels + 7) / 8;
    gm.n_sblocks = xr_div_up(n, EN_BLOCK);
    const uint32_t blocks = 8 * per_xcd * xr_div_up(n, EN_BLOCK);
    hipLaunchKernelGGL(k_hashgrid_fwd, dim3(blocks), dim3(EN_BLOCK), 0, (hipStream_t)stream_, gm, hm, table, x, x_stride, n,
                       n_dev, rows, enc_t, ld);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_hashgrid_bwd(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev, int n_levels,
                               const float* scale_host, const uint32_t* resolution_host, const uint32_t* offset_host,
                               float* grad_table, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(x && denc_t && grad_table, "null pointer");
    XR_REQUIRE(x_stride >= 3 && ld >= n, "bad stride");
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    const uint32_t per_xcd = (n_levels + 7) / 8;
    const uint32_t bw_ch = BW_CH;      // swept 8..128 on MI355X: 0.71-0.75 ms at 2^18 dense-gradient samples, flat
    gm.n_sblocks = xr_div_up(n, bw_ch * (EN_BLOCK / 16));
    const uint32_t blocks = 8 * per_xcd * gm.n_sblocks;
    hipLaunchKernelGGL(k_hashgrid_bwd, dim3(blocks), dim3(EN_BLOCK), 0, (hipStream_t)stream_, gm, hm, x, x_stride, denc_t, ld,
                       n, n_dev, grad_table);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ SH degree 4 (tcnn SphericalHarmonics)
__device__ inline void sh4_eval(float x, float y, float z, float* o) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}
__global__ __launch_bounds__(EN_BLOCK) void k_sh4(const float* __restrict__ dirs, uint32_t stride, uint32_t n,
                                                   float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * EN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* d = dirs + (size_t)i * stride;
    float o[16];
    sh4_eval(d[0] * 2.f - 1.f, d[1] * 2.f - 1.f, d[2] * 2.f - 1.f, o);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[4 * (size_t)i + q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}
extern "C" int xr_sh4(const float* dirs, uint32_t dir_stride, uint32_t n, float* out, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(dirs && out && dir_stride >= 3, "bad argument");
    XR_REQUIRE(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    hipLaunchKernelGGL(k_sh4, dim3(xr_div_up(n, EN_BLOCK)), dim3(EN_BLOCK), 0, (hipStream_t)stream_, dirs, dir_stride, n,
                       (float4*)out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
