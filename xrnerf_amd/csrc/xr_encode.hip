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
#include "xr_hashgrid.h"
#include "xr_scatter.h"
#include "xr_adam.h"
#include <cstdlib>

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

// one (sample, level): 8 corners as 4 x-neighbour pairs -> the level's two features.  The two x-neighbours of a (y,z) corner pair are
// adjacent table entries whenever their indices differ only in bit 0 (dense levels with an even index, hashed levels with an even
// x: idx ^ 1): one 16-B load then serves both corners.  (Rounds 2-3 measured the alternatives -- eight plain 8-B loads, non-temporal
// loads at the hashed levels, the two coarsest levels from LDS, four other block-to-XCD orders -- and kept this form with the
// cost-balanced map below: profiles/r02_microbench_fwd_variants.txt, profiles/r03_microbench_fwd3.txt.)
template <bool POW2> __device__ inline void hg_sample_level(const float2* __restrict__ tab, const float* xp, uint32_t x_cs, float scale,
                                                            uint32_t res, uint32_t hsize, bool hashed, float* r0_, float* r1_) {
    float w[3]; uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = xp[(size_t)d * x_cs] * scale + 0.5f;
        const float f = floorf(p);
        g[d] = (uint32_t)(int)f; w[d] = p - f;
    }
    float v0x[4], v0y[4], v1x[4], v1y[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t gy = g[1] + (p & 1), gz = g[2] + (p >> 1);
        const uint32_t i0 = POW2 ? grid_index_pow2(g[0], gy, gz, hsize - 1u) : grid_index(g[0], gy, gz, res, hsize, hashed);
        const uint32_t i1 = POW2 ? grid_index_pow2(g[0] + 1, gy, gz, hsize - 1u) : grid_index(g[0] + 1, gy, gz, res, hsize, hashed);
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
    *r0_ = r0; *r1_ = r1;
}

#ifdef HG_CELL_MAJOR_PROBE
// TIMING PROBE ONLY (tools/build_variant.sh cellmajor xr_encode.hip -DHG_CELL_MAJOR_PROBE; results are NOT the encoding): what would the
// dense levels cost if the 8 corners of a cell sat side by side (one 64-B chunk = one line instead of four)?  The chunk is read from
// inside the table allocation at (8 off[l] + 8 cell) mod (entries - 8): the footprint and the access pattern of a real cell-major
// copy (21 MB over levels 0-4 of the Lego geometry), without building one.  profiles/r06_lookup_cell_major_probe.txt
__device__ inline void hg_sample_level_cm_probe(const float2* __restrict__ table_all, uint32_t total, uint32_t off_l, const float* xp, uint32_t x_cs,
                                                float scale, uint32_t res, float* r0_, float* r1_) {
    float w[3]; uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float p = xp[(size_t)d * x_cs] * scale + 0.5f; const float f = floorf(p); g[d] = (uint32_t)(int)f; w[d] = p - f; }
    const uint32_t cell = g[0] + g[1] * res + g[2] * res * res;
    const float4* __restrict__ q = reinterpret_cast<const float4*>(table_all + ((8u * off_l + 8u * cell) % (total - 8u) & ~1u));
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float4 v = q[p];
        const float wyz = ((p & 1) ? w[1] : 1.f - w[1]) * ((p >> 1) ? w[2] : 1.f - w[2]);
        r0 += wyz * ((1.f - w[0]) * v.x + w[0] * v.z); r1 += wyz * ((1.f - w[0]) * v.y + w[0] * v.w);
    }
    *r0_ = r0; *r1_ = r1;
}
#endif

// Order 5: an explicit map.  XCD k (= blockIdx % 8, an observation used for speed only) walks its list of segments
// (level, sample blocks [lo, hi)) in order -- whole levels first, then its share of the levels that are split.  The map is
// built on the host from one measured cost per level (hg_build_map): every XCD gets the same cost, the eight most expensive
// levels stay whole on one XCD each (their 4-MiB slices are faulted into ONE L2), only the cheap remainder is shared.
#define HG_MAP_SEGS 6
struct XcdMap {
    uint8_t level[8][HG_MAP_SEGS];
    uint8_t nseg[8];
    uint32_t lo[8][HG_MAP_SEGS], hi[8][HG_MAP_SEGS];
};

__global__ __launch_bounds__(EN_BLOCK) void k_hashgrid_fwd(GridMeta gm, XcdMap xm, uint32_t hashed_mask, const float* __restrict__ table,
                                                            const float* __restrict__ x, uint32_t x_stride, uint32_t x_cs, uint32_t n,
                                                            const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                            float* __restrict__ enc_t, uint32_t ld) {
    uint32_t l, sb;
    if (n_dev) n = min(n, *n_dev);
    if (gm.order == 5) {
        const uint32_t k = blockIdx.x & 7u;
        uint32_t j = blockIdx.x >> 3, sg = 0;
        const uint32_t ns = xm.nseg[k];
        while (sg < ns && j >= xm.hi[k][sg] - xm.lo[k][sg]) { j -= xm.hi[k][sg] - xm.lo[k][sg]; ++sg; }
        if (sg >= ns) return;
        l = xm.level[k][sg]; sb = xm.lo[k][sg] + j;
    } else {
        level_of_block(gm, &l, &sb);
        if (l >= (uint32_t)gm.n_levels) return;
    }
    const uint32_t i = sb * EN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float scale = gm.scale[l];
    const uint32_t res = gm.res[l], hsize = gm.off[l + 1] - gm.off[l];
    const bool hashed = (hashed_mask >> l) & 1;
    const float2* __restrict__ tab = (const float2*)table + gm.off[l];
    const float* xp = x + (size_t)(rows ? rows[i] : i) * x_stride;    // optional row indirection (render slices)
    float r0, r1;
    const bool pow2 = hashed && (hsize & (hsize - 1u)) == 0u;           // uniform for the block
    if (pow2) hg_sample_level<true>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
#ifdef HG_CELL_MAJOR_PROBE
    else if (!hashed) hg_sample_level_cm_probe((const float2*)table, gm.off[gm.n_levels], gm.off[l], xp, x_cs, scale, res, &r0, &r1);
#endif
    else hg_sample_level<false>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
    enc_t[(size_t)(2 * l) * ld + i] = r0;
    enc_t[(size_t)(2 * l + 1) * ld + i] = r1;
}

// Scatter-add of the feature gradients with global atomics: the path for small row counts and for levels the binned scatter
// (xr_scatter.hip) has no layout for.  (Generations 1 and 2 of the bin / accumulate pair lived here until round 4: history.)
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
                                                            uint32_t n, const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                            float* __restrict__ grad_table) {
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
        s_x[e] = x[(size_t)(rows ? rows[b0 + i] : b0 + i) * x_stride + k];
    }
    for (uint32_t e = threadIdx.x; e < bn; e += EN_BLOCK) {
        const uint32_t r = rows ? rows[b0 + e] : b0 + e;
        s_d[0][e] = denc_t[(size_t)(2 * l) * ld + r];
        s_d[1][e] = denc_t[(size_t)(2 * l + 1) * ld + r];
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

// Host side of order 5.  cost[l] = time of level l alone on one XCD (any unit); a level goes to the XCD with the most budget left,
// whole when it fits (within 5 %), otherwise in pieces of sample blocks -- most expensive levels first, so the fine hashed levels
// stay whole and only the cheap remainder is shared between XCDs.  -> blocks per XCD (max over the XCDs), 0 = no map possible.
static uint32_t hg_build_map(XcdMap* xm, const float* cost, int n_levels, uint32_t nsb) {
    memset(xm, 0, sizeof(*xm));
    double T = 0;
    for (int l = 0; l < n_levels; ++l) T += cost[l] > 0.f ? cost[l] : 1e-3;
    T /= 8.0;
    int idx[EN_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) idx[l] = l;
    for (int i = 1; i < n_levels; ++i)                       // insertion sort, cost descending (finer level first on ties)
        for (int j = i; j > 0 && (cost[idx[j]] > cost[idx[j - 1]] || (cost[idx[j]] == cost[idx[j - 1]] && idx[j] > idx[j - 1])); --j) {
            const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t;
        }
    double load[8] = {0};
    uint32_t blocks[8] = {0};
    for (int i = 0; i < n_levels; ++i) {
        const int l = idx[i];
        const double c = cost[l] > 0.f ? cost[l] : 1e-3, w = c / nsb;     // cost per sample block
        uint32_t next = 0;
        while (next < nsb) {
            int k = -1;
            for (int q = 0; q < 8; ++q)
                if (xm->nseg[q] < HG_MAP_SEGS && (k < 0 || T - load[q] > T - load[k])) k = q;
            if (k < 0) return 0;
            const double capk = T - load[k], rem = w * (nsb - next);
            uint32_t take = nsb - next;
            if (capk > 0 && capk < 0.95 * rem) { take = (uint32_t)(capk / w); if (take == 0) take = 1; }
            const uint32_t sg = xm->nseg[k]++;
            xm->level[k][sg] = (uint8_t)l; xm->lo[k][sg] = next; xm->hi[k][sg] = next + take;
            load[k] += w * take; blocks[k] += take; next += take;
        }
    }
    uint32_t mx = 0;
    for (int q = 0; q < 8; ++q) mx = blocks[q] > mx ? blocks[q] : mx;
    return mx;
}
// single-level times (a level alone on one XCD, 2.6e5 ray-ordered samples of the Lego geometry on the MI355X, us; tools/microbench_fwd3.py
// re-measures them, profiles/r03_microbench_fwd3.txt): dense levels 23-24 (bound by the texture-address rate of their 13 lane accesses
// per sample, all cache hits), hashed levels 27.6, 32.0, 38.9, 47.3, 50.5, 51.4, 51.6 ... from the coarsest on (the coarse ones still
// share lines between neighbouring samples)
static void hg_level_costs(float* cost, int n_levels, uint32_t hashed_mask) {
    static const float hashed_cost[] = {27.6f, 32.0f, 38.9f, 47.3f, 50.5f, 51.4f, 51.7f};
    int h = 0;
    for (int l = 0; l < n_levels; ++l) {
        if ((hashed_mask >> l) & 1) { cost[l] = hashed_cost[h < 6 ? h : 6]; ++h; }
        else cost[l] = 23.6f;
    }
}

extern "C" int xr_hashgrid_fwd(const float* table, const float* x, uint32_t x_stride, uint32_t x_comp_stride, uint32_t n, const uint32_t* n_dev,
                                const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                const uint32_t* offset_host, float* enc_t, uint32_t ld, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(table && x && enc_t, "null pointer");
    XR_REQUIRE(ld >= n && x_comp_stride >= 1 && (x_comp_stride > 1 ? x_stride >= 1 : x_stride >= 3), "bad stride");
    XR_REQUIRE(((uintptr_t)table & 15) == 0, "table must be 16-byte aligned");
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    hipStream_t stream = (hipStream_t)stream_;
    // block -> (level, sample block): the cost-balanced XCD map (hg_build_map); level-major when no map fits (more than 6 segments on an XCD)
    const uint32_t nsb = xr_div_up(n, EN_BLOCK);
    gm.n_sblocks = nsb;
    XcdMap xm;
    float cost[EN_MAX_LEVELS];
    hg_level_costs(cost, n_levels, hm);
    const uint32_t per_xcd = hg_build_map(&xm, cost, n_levels, nsb);
    uint32_t blocks;
    if (per_xcd) { gm.order = 5; blocks = 8 * per_xcd; }
    else { memset(&xm, 0, sizeof(xm)); blocks = 8 * ((n_levels + 7) / 8) * nsb; }
    hipLaunchKernelGGL(k_hashgrid_fwd, dim3(blocks), dim3(EN_BLOCK), 0, stream, gm, xm, hm, table, x, x_stride, x_comp_stride, n,
                       n_dev, rows, enc_t, ld);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" size_t xr_hashgrid_bwd_workspace_bytes(uint32_t n, int n_levels, const uint32_t* resolution_host, const uint32_t* offset_host) {
    if (n_levels < 1 || n_levels > EN_MAX_LEVELS || !resolution_host || !offset_host) return 0;
    GridMeta gm; uint32_t hm;
    float dummy[EN_MAX_LEVELS] = {0};
    if (fill_meta(&gm, &hm, n_levels, dummy, resolution_host, offset_host) != 0) return 0;
    return xr_scatter3_workspace_bytes(n, gm, hm);
}

// Levels of `mask` through the atomic kernel (small n, tiny or oddly shaped tables): one launch per run of consecutive levels
static int hashgrid_bwd_atomic_levels(const GridMeta& gm, uint32_t hm, uint32_t mask, const float* x, uint32_t x_stride, const float* denc_t,
                                      uint32_t ld, uint32_t n, const uint32_t* n_dev, const uint32_t* rows, float* grad_table, int overwrite,
                                      hipStream_t stream) {
    int l = 0;
    while (l < gm.n_levels) {
        if (!((mask >> l) & 1)) { ++l; continue; }
        int e = l;
        while (e < gm.n_levels && ((mask >> e) & 1)) ++e;
        GridMeta gd = gm;
        gd.n_levels = e - l;
        for (int i = 0; i < gd.n_levels; ++i) { gd.scale[i] = gm.scale[l + i]; gd.res[i] = gm.res[l + i]; gd.off[i] = gm.off[l + i]; }
        gd.off[gd.n_levels] = gm.off[e];
        if (gd.n_levels & 7) gd.order = 2;
        gd.n_sblocks = xr_div_up(n, BW_CH * (EN_BLOCK / 16));
        const uint32_t blocks = (gd.order == 2 ? (uint32_t)gd.n_levels : 8u * ((gd.n_levels + 7) / 8)) * gd.n_sblocks;
        if (overwrite) XR_HIP(hipMemsetAsync(grad_table + 2 * (size_t)gm.off[l], 0, 2 * (size_t)(gm.off[e] - gm.off[l]) * sizeof(float), stream));
        hipLaunchKernelGGL(k_hashgrid_bwd, dim3(blocks), dim3(EN_BLOCK), 0, stream, gd, hm >> l, x, x_stride, denc_t + (size_t)2 * l * ld, ld,
                           n, n_dev, rows, grad_table);
        XR_LAUNCH_CHECK();
        l = e;
    }
    return XR_OK;
}

extern "C" int xr_hashgrid_bwd(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                                const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                const uint32_t* offset_host, float* grad_table, void* workspace, size_t workspace_bytes, int flags,
                                void* stream_) {
    XR_REQUIRE(grad_table && scale_host && resolution_host && offset_host, "null pointer");
    XR_REQUIRE((flags & ~XR_SCATTER_OVERWRITE) == 0, "unknown flag");
    const int overwrite = (flags & XR_SCATTER_OVERWRITE) ? 1 : 0;
    hipStream_t stream = (hipStream_t)stream_;
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    if (n == 0) {
        if (overwrite) XR_HIP(hipMemsetAsync(grad_table + 2 * (size_t)gm.off[0], 0, 2 * (size_t)(gm.off[n_levels] - gm.off[0]) * sizeof(float), stream));
        return XR_OK;
    }
    XR_REQUIRE(x && denc_t, "null pointer");
    XR_REQUIRE(x_stride >= 3 && ld >= n, "bad stride");
    XR_REQUIRE(!rows || n_dev, "a row list comes with its device-side length (n_dev)");
    // levels without a non-atomic path (tiny / oddly shaped tables, small n) take the atomic kernel -- beside the binned levels on a
    // helper stream when both exist (disjoint table slices)
    uint32_t amask = xr_scatter3_atomic_mask(n, gm, hm, workspace && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)grad_table & 15) == 0);
    const uint32_t all = (1u << n_levels) - 1u;
    const XrHelper* hp = xr_internal_helper();            // the caller's helper stream (xr_set_helper_stream); none: in order on `stream`
    const bool fork = hp && amask != 0 && amask != all;
    const hipStream_t aux = hp ? hp->stream : nullptr;
    const hipEvent_t ev_fork = hp ? hp->fork : nullptr, ev_join = hp ? hp->join : nullptr;
    if (fork) {
        XR_HIP(hipEventRecord(ev_fork, stream));
        XR_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
    }
    if (amask) {
        const int rc = hashgrid_bwd_atomic_levels(gm, hm, amask, x, x_stride, denc_t, ld, n, n_dev, rows, grad_table, overwrite, fork ? aux : stream);
        if (rc != XR_OK) return rc;
        if (fork) XR_HIP(hipEventRecord(ev_join, aux));
    }
    if (amask != all) {
        uint32_t amask2 = 0;
        const int rc = xr_scatter3(x, x_stride, denc_t, ld, n, n_dev, rows, gm, hm, grad_table, workspace, workspace_bytes, overwrite, &amask2, stream);
        if (rc != XR_OK) return rc;
    }
    if (fork) XR_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    return XR_OK;
}

int xr_internal_hashgrid_bwd_adam_supported(uint32_t n, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                            const uint32_t* offset_host) {
    GridMeta gm; uint32_t hm;
    if (!scale_host || !resolution_host || !offset_host || n == 0) return 0;
    if (fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) != 0) return 0;
    return xr_scatter3_atomic_mask(n, gm, hm, true) == 0 ? 1 : 0;
}
extern "C" int xr_hashgrid_bwd_adam(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                                    const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                    const uint32_t* offset_host, void* workspace, size_t workspace_bytes, const xr_adam_fuse* adam,
                                    void* stream_) {
    if (!adam) {                     // dry run: is there a non-atomic path for every level at this row capacity?
        XR_REQUIRE(xr_internal_hashgrid_bwd_adam_supported(n, n_levels, scale_host, resolution_host, offset_host),
                   "a level of this geometry / row count has no non-atomic path: scatter and step separately");
        return XR_OK;
    }
    XR_REQUIRE(x && denc_t && scale_host && resolution_host && offset_host && workspace, "null pointer");
    XR_REQUIRE(adam->param && adam->m && adam->v && adam->step >= 1, "bad optimiser state");
    XR_REQUIRE((((uintptr_t)adam->param | (uintptr_t)adam->m | (uintptr_t)adam->v | (uintptr_t)adam->ema | (uintptr_t)workspace) & 15) == 0,
               "buffers must be 16-byte aligned");
    XR_REQUIRE(n > 0 && x_stride >= 3 && ld >= n, "bad sizes");
    XR_REQUIRE(!rows || n_dev, "a row list comes with its device-side length (n_dev)");
    XR_REQUIRE(xr_internal_hashgrid_bwd_adam_supported(n, n_levels, scale_host, resolution_host, offset_host),
               "a level of this geometry / row count has no non-atomic path: scatter and step separately");
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    // the constants exactly as xr_adam_step_multi hands them to its kernel
    const float bc1 = 1.f - powf(adam->beta1, (float)adam->step), bc2 = 1.f - powf(adam->beta2, (float)adam->step);
    XrAdamArgs A = {adam->param, adam->m, adam->v, adam->ema, adam->beta1, adam->beta2, adam->lr / bc1, sqrtf(bc2), adam->eps,
                    adam->weight_decay, adam->ema_momentum, adam->grad_scale};
    uint32_t amask = 0;
    return xr_scatter3(x, x_stride, denc_t, ld, n, n_dev, rows, gm, hm, adam->param /* alignment check only */, workspace, workspace_bytes,
                       1, &amask, (hipStream_t)stream_, &A);
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
