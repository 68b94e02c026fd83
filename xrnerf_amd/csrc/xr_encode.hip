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

// Order 3.  Measured with order 1 (XCD x owns levels x and x+8): the 11 hashed levels fall 2-2-2-1-1-1-1-1 on the 8
// XCDs, so three XCDs gather twice as long as the other five and the kernel takes 2 hashed-level times instead of 11/8.
// Here the (level, sample block) pairs are laid out finest level first, each weighted with its level's relative cost
// 2^wsh (hashed levels: one L2 line request per corner pair; dense levels: mostly L1 hits), and cut into 8 contiguous
// pieces of equal weight -- XCD k (= blockIdx % 8, an observation used for speed only) walks piece k in order.  A level
// is then shared by at most 2-3 XCDs (its 4 MiB slice is loaded into each of their L2s: +20 MB of fills per launch) and
// every XCD still works on one level at a time.  Sample block s of visiting slot li sits at weighted position
// p = nsb * P_li + s * w_li; it belongs to XCD k iff k*W <= 8p < (k+1)*W, W = nsb * wsum.  All uniform (scalar) maths.
__host__ __device__ inline void hg_xcd_segment(const GridMeta& gm, uint32_t k, uint32_t nsb, uint64_t P, uint32_t sh,
                                               uint32_t* s_lo, uint32_t* s_hi) {
    const uint64_t W = (uint64_t)nsb * gm.wsum, base8 = 8ull * P * nsb, lo = (uint64_t)k * W, hi = lo + W;
    const uint32_t q = 3u + sh;
    const uint64_t a = lo > base8 ? (lo - base8 + ((1ull << q) - 1)) >> q : 0ull;
    const uint64_t b = hi > base8 ? (hi - base8 + ((1ull << q) - 1)) >> q : 0ull;
    *s_lo = (uint32_t)(a < nsb ? a : nsb);
    *s_hi = (uint32_t)(b < nsb ? b : nsb);
}
// -> false when block j of XCD k has no work
__host__ __device__ inline bool hg_balanced_block(const GridMeta& gm, uint32_t k, uint32_t j, uint32_t nsb, uint32_t* level,
                                                  uint32_t* sblock) {
    uint64_t P = 0;
    for (int l = gm.n_levels - 1; l >= gm.l_min; --l) {
        uint32_t s_lo, s_hi;
        hg_xcd_segment(gm, k, nsb, P, gm.wsh[l], &s_lo, &s_hi);
        const uint32_t cnt = s_hi - s_lo;
        if (j < cnt) { *level = (uint32_t)l; *sblock = s_lo + j; return true; }
        j -= cnt;
        P += 1ull << gm.wsh[l];
    }
    return false;
}
// Order 4.  What order 3 measured (profiles/r02_microbench_hash_a.txt): run time = (sample blocks of the busiest XCD) x
// ~44 us per 1012 blocks, the same for dense and hashed blocks -- every block costs the same texture-address time -- while
// the L2 request load (4 line requests per hashed sample-level, next to none at the dense levels) is what differs.  So:
// the same NUMBER of blocks on every XCD, and the hashed levels' blocks spread evenly: the (level, sample block) pairs of
// the hashed levels [n_levels - n_hashed, n_levels), finest first, are cut into 8 equal pieces, the dense levels' pairs
// likewise; XCD k walks its hashed piece, then its dense piece.
__host__ __device__ inline bool hg_twolist_block(const GridMeta& gm, uint32_t k, uint32_t j, uint32_t nsb, uint32_t* level,
                                                 uint32_t* sblock) {
    const uint32_t nl = (uint32_t)(gm.n_levels - gm.l_min), nh = gm.n_hashed < nl ? gm.n_hashed : nl, nd = nl - nh;
    // list of m levels x nsb blocks cut at multiples of m*nsb/8 (rounded up): piece k = [k*T/8, (k+1)*T/8)
    const uint64_t Th = (uint64_t)nh * nsb, Td = (uint64_t)nd * nsb;
    const uint64_t h0 = (Th * k + 7) / 8, h1 = (Th * (k + 1) + 7) / 8;
    uint64_t e;
    if (j < h1 - h0) {
        e = h0 + j;
        *level = (uint32_t)gm.n_levels - 1u - (uint32_t)(e / nsb);
    } else {
        const uint64_t d0 = (Td * k + 7) / 8, d1 = (Td * (k + 1) + 7) / 8;
        const uint64_t jj = j - (h1 - h0);
        if (jj >= d1 - d0) return false;
        e = d0 + jj;
        *level = (uint32_t)gm.n_levels - 1u - nh - (uint32_t)(e / nsb);
    }
    *sblock = (uint32_t)(e % nsb);
    return true;
}
static uint32_t hg_balanced_blocks_per_xcd(const GridMeta& gm, uint32_t nsb) {
    uint32_t mx = 0;
    for (uint32_t k = 0; k < 8; ++k) {
        uint64_t P = 0; uint32_t c = 0;
        for (int l = gm.n_levels - 1; l >= gm.l_min; --l) {
            uint32_t s_lo, s_hi;
            hg_xcd_segment(gm, k, nsb, P, gm.wsh[l], &s_lo, &s_hi);
            c += s_hi - s_lo; P += 1ull << gm.wsh[l];
        }
        mx = c > mx ? c : mx;
    }
    return mx + (uint32_t)gm.n_levels;      // the device recomputes with n_dev <= n: at most one block more per level
}

typedef float hg_f4 __attribute__((ext_vector_type(4)));
typedef float hg_f2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ inline float4 hg_ld4(const float2* p) {
    if (NT) { const hg_f4 r = __builtin_nontemporal_load(reinterpret_cast<const hg_f4*>(p)); return make_float4(r.x, r.y, r.z, r.w); }
    return *reinterpret_cast<const float4*>(p);
}
template <bool NT> __device__ inline float2 hg_ld2(const float2* p) {
    if (NT) { const hg_f2 r = __builtin_nontemporal_load(reinterpret_cast<const hg_f2*>(p)); return make_float2(r.x, r.y); }
    return *p;
}

// one (sample, level): 8 corners as 4 x-neighbour pairs -> the level's two features
template <bool NT, bool POW2, bool PAIRS> __device__ inline void hg_sample_level(const float2* __restrict__ tab, const float* xp, uint32_t x_cs, float scale,
                                                                      uint32_t res, uint32_t hsize, bool hashed, float* r0_, float* r1_) {
    float w[3]; uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = xp[(size_t)d * x_cs] * scale + 0.5f;
        const float f = floorf(p);
        g[d] = (uint32_t)(int)f; w[d] = p - f;
    }
    // PAIRS (the round-1 kernel): the two x-neighbours of a (y,z) corner pair are adjacent table entries whenever their
    // indices differ only in bit 0 (dense levels with an even index, hashed levels with an even x: idx ^ 1), and one
    // 16-B load then serves both corners.  Measured (rocprofv3 PMC, profiles/r02_pmc_hashgrid_fwd_v0_round1_kernel.txt):
    // the kernel is bound by the texture-address path, ~1 lane-access per clock and CU (TCP_TOTAL_ACCESSES 55.8 M over
    // 256 CUs x 219 K cycles) -- an issued vector-memory instruction costs its 16 cycles per dword of width whatever its
    // exec mask, and the adjacent / not-adjacent branch is divergent in nearly every wave, so each pair pays one
    // dwordx4 AND two dwordx2 (128 cycles) instead of two dwordx2 (64).  Default now: eight plain 8-B loads, issued
    // together; the second load of a pair finds its line in L1 (or merges with the pending miss) 15 times out of 16.
    float v0x[4], v0y[4], v1x[4], v1y[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t gy = g[1] + (p & 1), gz = g[2] + (p >> 1);
        const uint32_t i0 = POW2 ? grid_index_pow2(g[0], gy, gz, hsize - 1u) : grid_index(g[0], gy, gz, res, hsize, hashed);
        const uint32_t i1 = POW2 ? grid_index_pow2(g[0] + 1, gy, gz, hsize - 1u) : grid_index(g[0] + 1, gy, gz, res, hsize, hashed);
        if (PAIRS && (i0 ^ i1) == 1u) {
            const float4 q = hg_ld4<NT>(tab + (i0 & ~1u));
            const bool odd = i0 & 1u;
            v0x[p] = odd ? q.z : q.x; v0y[p] = odd ? q.w : q.y;
            v1x[p] = odd ? q.x : q.z; v1y[p] = odd ? q.y : q.w;
        } else {
            const float2 a = hg_ld2<NT>(tab + i0), b = hg_ld2<NT>(tab + i1);
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
    } else if (gm.order == 4) {
        if (!hg_twolist_block(gm, blockIdx.x & 7u, blockIdx.x >> 3, (n + EN_BLOCK - 1) / EN_BLOCK, &l, &sb)) return;
    } else if (gm.order == 3) {
        if (!hg_balanced_block(gm, blockIdx.x & 7u, blockIdx.x >> 3, (n + EN_BLOCK - 1) / EN_BLOCK, &l, &sb)) return;
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
    if (gm.pairs) {
        if (pow2) hg_sample_level<false, true, true>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
        else hg_sample_level<false, false, true>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
    } else if (pow2 && gm.nt) hg_sample_level<true, true, false>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
    else if (pow2) hg_sample_level<false, true, false>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
    else hg_sample_level<false, false, false>(tab, xp, x_cs, scale, res, hsize, hashed, &r0, &r1);
    enc_t[(size_t)(2 * l) * ld + i] = r0;
    enc_t[(size_t)(2 * l + 1) * ld + i] = r1;
}

// The coarsest levels from LDS (north_star: "LDS staging of per-level feature tiles").  Levels [0, n_lds) -- at the
// config's geometry levels 0 and 1, 4096 + 12168 entries = 127 KiB -- are copied into the workgroup's LDS with
// coalesced 16-B loads, then every thread gathers its samples' corners with ds_read_b64 (2 LDS cycles per 64
// conflict-free lanes, against one L1 tag lookup per distinct line).  One workgroup per CU, grid-strided over samples.
#define EN_LDS_THREADS 1024
__global__ __launch_bounds__(EN_LDS_THREADS) void k_hashgrid_fwd_lds(GridMeta gm, int n_lds, const float* __restrict__ table,
                                                                      const float* __restrict__ x, uint32_t x_stride, uint32_t n,
                                                                      const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                                      float* __restrict__ enc_t, uint32_t ld) {
    extern __shared__ __attribute__((aligned(16))) float2 s_tab[];
    if (n_dev) n = min(n, *n_dev);
    if (blockIdx.x * EN_LDS_THREADS >= n) return;
    const uint32_t n_ent = gm.off[n_lds];                       // entries of levels [0, n_lds); a multiple of 8
    {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(table);
        float4* dst = reinterpret_cast<float4*>(s_tab);
        for (uint32_t e = threadIdx.x; e < n_ent / 2; e += EN_LDS_THREADS) dst[e] = src[e];
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * EN_LDS_THREADS + threadIdx.x; i < n; i += gridDim.x * EN_LDS_THREADS) {
        const float* xp = x + (size_t)(rows ? rows[i] : i) * x_stride;
        const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
        for (int l = 0; l < n_lds; ++l) {
            const float scale = gm.scale[l];
            const uint32_t res = gm.res[l];
            const float2* tab = s_tab + gm.off[l];
            const float p0 = x0 * scale + 0.5f, p1 = x1 * scale + 0.5f, p2 = x2 * scale + 0.5f;
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            const uint32_t hsize = gm.off[l + 1] - gm.off[l];
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {    // the oracle's order: corner 0..7, x fastest
                const uint32_t idx = grid_index(g0 + (c & 1), g1 + ((c >> 1) & 1), g2 + (c >> 2), res, hsize, false);
                const float2 v = tab[idx];
                const float wt = ((c & 1) ? w0 : 1.f - w0) * ((c & 2) ? w1 : 1.f - w1) * ((c & 4) ? w2 : 1.f - w2);
                r0 += wt * v.x; r1 += wt * v.y;
            }
            enc_t[(size_t)(2 * l) * ld + i] = r0;
            enc_t[(size_t)(2 * l + 1) * ld + i] = r1;
        }
    }
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
                                                            uint32_t n, const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                            float* __restrict__ grad_table, float* __restrict__ rep, uint32_t n_rep, uint32_t rep_stride,
                                                            uint32_t rep_levels) {
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
    // coarse levels: thousands of flushes land on the few hundred entries around the object and same-address
    // atomics serialise (level 0 alone: 63 us).  With `rep`, workgroup sb adds into replica sb % n_rep of the
    // slice; k_reduce_replicas folds the replicas into the table afterwards.
    float* __restrict__ tab = (rep && l < rep_levels ? rep + (size_t)(sb % n_rep) * rep_stride : grad_table) + 2 * (size_t)gm.off[l] + f;
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

// ---- scatter without global atomics at the hashed levels: bin, then accumulate in LDS ---------------
// The atomic scatter above is bound by the chip-wide rate of scattered L2 atomic REQUESTS (measured
// 21-24 G/s whatever the type, footprint or XCD placement): at the fine, hashed levels nothing merges
// and every sample costs 4-8 requests per level (0.46 ms for the 8 finest levels at 2^18 samples).
// Here a hashed level's gradient slice (2^19 entries x 2 floats = 4 MiB) is cut into 2^13-entry
// PARTITIONS (128 KiB of fp64 accumulators) -- what one workgroup can hold in LDS -- and the work is split in two streaming
// kernels with no global atomic at all:
//   A  k_scatter_bin    one thread per (sample, level), 4096 samples per workgroup: the 8 corners are
//      4 x-neighbour PAIRS; the partition of a pair is bits [13,19) of  y*P1 ^ z*P2  (x < 2^13 never
//      reaches them), so both corners of a pair live in the same partition.  Each pair becomes one
//      16-byte item {idx0 | idx1 << 13, wy*wz*d0, wy*wz*d1, wx} stored in the workgroup's PRIVATE
//      sub-bin of its (level, partition); ranks come from an LDS histogram (returning LDS atomics),
//      the sub-bin's fill count is stored next to it.  Streaming: reads 28 + 8 B, writes 64 B per
//      (sample, level).
//   B  k_scatter_accum  one workgroup per (level, partition): walks the sub-bins of all sample blocks
//      with coalesced 16-B loads, 4 LDS fp64 atomic adds per item, then adds its partition to the table with
//      plain 16-B accesses -- it is the only writer of that slice.
// A sub-bin holds 2x its expected load; items beyond that (adversarially clustered inputs) are
// scattered by A directly with global atomics, so any input stays correct.
// Measured at 2^18 samples, 11 hashed levels: A 79 us (16 us without its 176 MB of item stores), B 105 us.
// Measured dead ends, for the record: (1) every partition-owning workgroup scanning ALL samples and
// filtering by partition: 0.22 ms per 8 levels (32x redundant filter work at 4 cycles per wave64 VALU
// instruction, latency-bound gathers); (2) shared bins with one returning global atomic per
// (workgroup, partition): 0.19 ms even for ONE level -- 64 same-address returning atomics serialise
// at ~3 us each.
#define SC_THREADS 1024                            // binning kernel
#ifndef SC_LOG2
#define SC_LOG2 13
#endif
#ifndef SC_ACC_THREADS
#define SC_ACC_THREADS 1024                        // accumulate kernel (measured: 2^12-entry partitions with two 512-thread
#endif                                             // workgroups per CU, 0.287 ms, do not beat 2^13 / 1024 / one per CU, 0.282 ms)
#define SC_ENTRIES (1u << SC_LOG2)
#define SC_LDS_BYTES (SC_ENTRIES * 2 * sizeof(double))
#define SC_SPT 4                                   // samples per thread in the binning kernel
#define SC_BLOCK_SAMPLES (SC_THREADS * SC_SPT)
#define SC_MAX_PARTS 256
#define SC_MAX_SB 1024                               // sample blocks per call: n <= 2^22 on the binned path
#define SC_SUB_ITEMS (2u * 4u * SC_BLOCK_SAMPLES)  // items of all sub-bins of one (workgroup, level): 2x the 4 pairs per sample

__global__ __launch_bounds__(SC_THREADS) void k_scatter_bin(GridMeta gm, uint32_t l_lo, uint32_t l_hi, uint32_t parts, uint32_t nsb,
                                                            const float* __restrict__ x, uint32_t x_stride,
                                                            const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                                            const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                            uint32_t* __restrict__ counts,
                                                            float4* __restrict__ bins, float* __restrict__ grad_table) {
    __shared__ uint32_t s_cnt[SC_MAX_PARTS];
    const uint32_t nl = l_hi - l_lo;
    const uint32_t l = l_hi - 1 - blockIdx.x % nl, sb = blockIdx.x / nl;   // finest first; the levels of one sample block are neighbours
    if (n_dev) n = min(n, *n_dev);
    const uint32_t b0 = sb * SC_BLOCK_SAMPLES;
    const uint32_t cap = SC_SUB_ITEMS / parts;                              // capacity of one sub-bin
    uint32_t* __restrict__ cnt_out = counts + ((size_t)(l - l_lo) * parts) * nsb + sb;      // [level][part][sample block]
    if (b0 >= n) {                                                          // uniform: an empty sample block has empty sub-bins
        for (uint32_t p = threadIdx.x; p < parts; p += SC_THREADS) cnt_out[(size_t)p * nsb] = 0;
        return;
    }
    for (uint32_t p = threadIdx.x; p < parts; p += SC_THREADS) s_cnt[p] = 0;
    __syncthreads();
    const float scale = gm.scale[l];
    const uint32_t hsize = gm.off[l + 1] - gm.off[l], hmask = hsize - 1;    // power of two (checked on the host)
    const float* __restrict__ d0p = denc_t + (size_t)(2 * l) * ld;
    const float* __restrict__ d1p = d0p + ld;
    float* __restrict__ tab = grad_table + 2 * (size_t)gm.off[l];
    // sub-bin (level, part, sample block) at [level][part][sample block][cap]: the reader of one (level, part)
    // streams nsb * cap contiguous items
    float4* __restrict__ out = bins + (size_t)(l - l_lo) * nsb * SC_SUB_ITEMS + (size_t)sb * cap;
    // all 5 x SC_SPT loads of the thread are issued before anything is consumed (one exposed memory latency
    // per workgroup instead of one per sample)
    float ld0[SC_SPT], ld1[SC_SPT], lx0[SC_SPT], lx1[SC_SPT], lx2[SC_SPT];
#pragma unroll
    for (uint32_t s = 0; s < SC_SPT; ++s) {
        uint32_t i = min(b0 + s * SC_THREADS + threadIdx.x, n - 1);
        if (rows) i = rows[i];
        const float* xp = x + (size_t)i * x_stride;
        ld0[s] = d0p[i]; ld1[s] = d1p[i]; lx0[s] = xp[0]; lx1[s] = xp[1]; lx2[s] = xp[2];
    }
#pragma unroll
    for (uint32_t s = 0; s < SC_SPT; ++s) {
        const uint32_t i = b0 + s * SC_THREADS + threadIdx.x;
        if (i >= n) continue;
        const float d0 = ld0[s], d1 = ld1[s], x0 = lx0[s], x1 = lx1[s], x2 = lx2[s];
        if (d0 == 0.f && d1 == 0.f) continue;                               // rows behind the compositor's early stop add nothing
        const float p0 = x0 * scale + 0.5f, p1 = x1 * scale + 0.5f, p2 = x2 * scale + 0.5f;
        const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
        const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
        const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
        const uint32_t a0 = g1 * 2654435761u, b0h = g2 * 805459861u;
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t cy = c & 1u, cz = c >> 1;
            const uint32_t h = (a0 + (cy ? 2654435761u : 0u)) ^ (b0h + (cz ? 805459861u : 0u));
            const uint32_t i0 = (g0 ^ h) & hmask, i1 = ((g0 + 1u) ^ h) & hmask;
            const uint32_t part = i0 >> SC_LOG2;                            // == i1 >> SC_LOG2
            const float wyz = (cy ? w1 : 1.f - w1) * (cz ? w2 : 1.f - w2);
            const float va = wyz * d0, vb = wyz * d1;
            const uint32_t rank = atomicAdd(&s_cnt[part], 1u);
            if (rank < cap) {
                const uint32_t pr = (i0 & (SC_ENTRIES - 1)) | ((i1 & (SC_ENTRIES - 1)) << SC_LOG2);
                out[(size_t)part * nsb * cap + rank] = make_float4(__uint_as_float(pr), va, vb, w0);
            } else {                                                        // overfull sub-bin: scatter this pair directly
                unsafeAtomicAdd(tab + 2 * (size_t)i0, (1.f - w0) * va);
                unsafeAtomicAdd(tab + 2 * (size_t)i0 + 1, (1.f - w0) * vb);
                unsafeAtomicAdd(tab + 2 * (size_t)i1, w0 * va);
                unsafeAtomicAdd(tab + 2 * (size_t)i1 + 1, w0 * vb);
            }
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < parts; p += SC_THREADS) cnt_out[(size_t)p * nsb] = min(s_cnt[p], cap);
}

// LDS accumulation.  MEASURED on MI355X (tools/lds_probe.hip), ns per wave64 instruction per CU, random addresses:
//   ds_add_f32 81 (!)   ds_add_f64 8.6   ds_add_u32 3.0   ds_add_u64 4.7   ds_cmpst_rtn_b64 8.9   8-byte read+write 7.2
// The fp32 LDS atomic add is serialised per lane (~3 cycles each); the fp64 one is not.  (A 64-bit compare-and-
// swap of the (f0, f1) pair has the throughput but needs the returned value: two exposed LDS latencies per add,
// 62 us per partition.)  So the partition is accumulated in DOUBLE with returnless ds_add_f64 -- more accurate
// than the fp32 atomics of the other path -- and rounded to fp32 once, when it is added to the table.
#ifdef SC_TIMING
__device__ long long g_sc_t[24];
#define SC_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sc_t[k] = wall_clock64(); } while (0)
#else
#define SC_T(k)
#endif
__global__ __launch_bounds__(SC_ACC_THREADS) void k_scatter_accum(GridMeta gm, uint32_t l_lo, uint32_t l_hi, uint32_t parts, uint32_t nsb,
                                                              const uint32_t* __restrict__ counts, const float4* __restrict__ bins,
                                                              float* __restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) double s_acc[];       // [SC_ENTRIES][2]
    const uint32_t li = blockIdx.x / parts, part = blockIdx.x % parts, l = l_lo + li;
    const uint32_t hsize = gm.off[l + 1] - gm.off[l];
    if (part >= (hsize >> SC_LOG2)) return;
    const uint32_t cap = SC_SUB_ITEMS / parts;
    const uint32_t* __restrict__ cnt = counts + ((size_t)li * parts + part) * nsb;
    double2* acc2 = reinterpret_cast<double2*>(s_acc);
    SC_T(0);
    __shared__ uint32_t s_fill[SC_MAX_SB];                                  // fill counts of this unit's sub-bins
    for (uint32_t e = threadIdx.x; e < nsb; e += SC_ACC_THREADS) s_fill[e] = cnt[e];
    for (uint32_t e = threadIdx.x; e < SC_ENTRIES; e += SC_ACC_THREADS) acc2[e] = make_double2(0.0, 0.0);
    __syncthreads();
    // The unit's sub-bins are contiguous ([sample block][cap], cap a power of two): tile k = positions
    // [k * SC_ACC_THREADS, (k+1) * SC_ACC_THREADS), a position is live when its offset in its sub-bin is below the fill
    // count.  The loads of the next U tiles are in flight while the current U are accumulated.
    const uint32_t cap_log2 = 31 - __builtin_clz(cap), npos = nsb << cap_log2;
    const float4* __restrict__ src = bins + (size_t)li * nsb * SC_SUB_ITEMS + (size_t)part * npos;
    constexpr uint32_t U = 8;
    float4 nx[U];
    bool non[U];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t pos = (k0 + u) * SC_ACC_THREADS + threadIdx.x;
            non[u] = pos < npos && (pos & (cap - 1)) < s_fill[min(pos >> cap_log2, nsb - 1)];
            if (non[u]) nx[u] = src[pos];
        }
    };
    const uint32_t ntiles = (npos + SC_ACC_THREADS - 1) / SC_ACC_THREADS;
    SC_T(1);
    fetch(0);
    for (uint32_t k0 = 0; k0 < ntiles; k0 += U) {
        float4 it[U];
        bool on[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { it[u] = nx[u]; on[u] = non[u]; }
        if (k0 + U < ntiles) fetch(k0 + U);
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const uint32_t pr = __float_as_uint(it[u].x), i0 = pr & (SC_ENTRIES - 1), i1 = pr >> SC_LOG2;
            const float w0 = it[u].w, a = it[u].y, b = it[u].z;
            atomicAdd(&s_acc[2 * i0], (double)((1.f - w0) * a));
            atomicAdd(&s_acc[2 * i0 + 1], (double)((1.f - w0) * b));
            atomicAdd(&s_acc[2 * i1], (double)(w0 * a));
            atomicAdd(&s_acc[2 * i1 + 1], (double)(w0 * b));
        }
    }
    SC_T(2);
    __syncthreads();
    float2* __restrict__ dst = reinterpret_cast<float2*>(grad_table + 2 * ((size_t)gm.off[l] + (size_t)part * SC_ENTRIES));
    SC_T(3);
    constexpr uint32_t F = SC_ENTRIES / SC_ACC_THREADS;                         // all F loads in flight before the first add
    float2 t[F];
#pragma unroll
    for (uint32_t k = 0; k < F; ++k) t[k] = dst[k * SC_ACC_THREADS + threadIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < F; ++k) {
        const double2 a = acc2[k * SC_ACC_THREADS + threadIdx.x];
        t[k].x += (float)a.x; t[k].y += (float)a.y;
        dst[k * SC_ACC_THREADS + threadIdx.x] = t[k];
    }
    SC_T(4);
}

// ---- second generation of the bin / accumulate pair (XR_SC_MODE=1, default) ---------------------------------------
// Measured on the first pair (rocprofv3, 2^18 samples, 11 hashed levels): k_scatter_bin 90 us, of which 16 us without
// its item stores -- 11.5 M scattered 16-B stores, each its own L2 write request into a partially written line -- and
// k_scatter_accum 140 us reading sub-bins that are half empty by construction (capacity 2x the expected fill, lanes
// past the fill idle in the loads AND in the LDS atomics).  Here:
//   A' k_scatter_bin2   512 threads, the workgroup's 4096 samples in 4 rounds of 1024: the round's <= 4096 items are
//      ranked per partition with LDS counters, placed in LDS in partition order (64 KiB), and copied out as contiguous
//      runs (~64 items = 1 KiB per partition and round): full 16-B-per-lane coalesced stores.  Two workgroups per CU.
//   B' k_scatter_accum2 sub-bin capacity 1.25x the expected fill; a wave walks whole sub-bins in 64-item chunks,
//      so only a sub-bin's last chunk has idle lanes; 8 chunk loads in flight per lane while the previous 8 are
//      accumulated.
#ifndef SB_THREADS
#define SB_THREADS 512
#endif
#ifndef SB_SPT
#define SB_SPT 2
#endif
#define SB_ROUND_SAMPLES (SB_THREADS * SB_SPT)
#define SB_ROUND_ITEMS (4 * SB_ROUND_SAMPLES)
#define SB_ROUNDS (SC_BLOCK_SAMPLES / SB_ROUND_SAMPLES)
#define SC_SUB_ITEMS2 (5u * SC_BLOCK_SAMPLES)      // items of all sub-bins of one (workgroup, level): 1.25x the 4 pairs per sample

__global__ __launch_bounds__(SB_THREADS) void k_scatter_bin2(GridMeta gm, uint32_t l_lo, uint32_t l_hi, uint32_t parts, uint32_t nsb,
                                                             const float* __restrict__ x, uint32_t x_stride,
                                                             const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                                             const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                             uint32_t* __restrict__ counts,
                                                             float4* __restrict__ bins, float* __restrict__ grad_table) {
    __shared__ float4 s_items[SB_ROUND_ITEMS];
    __shared__ uint8_t s_ipart[SB_ROUND_ITEMS];
    __shared__ uint32_t s_cnt[SC_MAX_PARTS], s_off[SC_MAX_PARTS + 1], s_base[SC_MAX_PARTS];
    const uint32_t nl = l_hi - l_lo;
    const uint32_t l = l_hi - 1 - blockIdx.x % nl, sb = blockIdx.x / nl;   // finest first; the levels of one sample block are neighbours
    if (n_dev) n = min(n, *n_dev);
    const uint32_t b0 = sb * SC_BLOCK_SAMPLES;
    const uint32_t cap = SC_SUB_ITEMS2 / parts;                             // capacity of one sub-bin
    uint32_t* __restrict__ cnt_out = counts + ((size_t)(l - l_lo) * parts) * nsb + sb;      // [level][part][sample block]
    if (b0 >= n) {                                                          // uniform: an empty sample block has empty sub-bins
        for (uint32_t p = threadIdx.x; p < parts; p += SB_THREADS) cnt_out[(size_t)p * nsb] = 0;
        return;
    }
    for (uint32_t p = threadIdx.x; p < parts; p += SB_THREADS) s_base[p] = 0;
    const float scale = gm.scale[l];
    const uint32_t hsize = gm.off[l + 1] - gm.off[l], hmask = hsize - 1;    // power of two (checked on the host)
    const float* __restrict__ d0p = denc_t + (size_t)(2 * l) * ld;
    const float* __restrict__ d1p = d0p + ld;
    float* __restrict__ tab = grad_table + 2 * (size_t)gm.off[l];
    float4* __restrict__ out = bins + (size_t)(l - l_lo) * nsb * SC_SUB_ITEMS2 + (size_t)sb * cap;
    const uint32_t per = (parts + 63u) / 64u;                               // partitions per lane in the offset scan (<= 4)
    for (uint32_t r = 0; r < SB_ROUNDS; ++r) {
        const uint32_t rb0 = b0 + r * SB_ROUND_SAMPLES;
        if (rb0 >= n) break;                                                // uniform
        if (r == 0) SC_T(8);
        for (uint32_t p = threadIdx.x; p < parts; p += SB_THREADS) s_cnt[p] = 0;
        __syncthreads();
        float ld0[SB_SPT], ld1[SB_SPT], lx0[SB_SPT], lx1[SB_SPT], lx2[SB_SPT];
#pragma unroll
        for (uint32_t s = 0; s < SB_SPT; ++s) {
            uint32_t i = min(rb0 + s * SB_THREADS + threadIdx.x, n - 1);
            if (rows) i = rows[i];
            const float* xp = x + (size_t)i * x_stride;
            ld0[s] = d0p[i]; ld1[s] = d1p[i]; lx0[s] = xp[0]; lx1[s] = xp[1]; lx2[s] = xp[2];
        }
        uint32_t ipart[SB_SPT * 4], irank[SB_SPT * 4], ipr[SB_SPT * 4];
        float iva[SB_SPT * 4], ivb[SB_SPT * 4], iw0[SB_SPT];
        bool live[SB_SPT];
#pragma unroll
        for (uint32_t s = 0; s < SB_SPT; ++s) {
            const uint32_t i = rb0 + s * SB_THREADS + threadIdx.x;
            const float d0 = ld0[s], d1 = ld1[s];
            live[s] = i < n && !(d0 == 0.f && d1 == 0.f);                   // rows behind the compositor's early stop add nothing
            if (!live[s]) continue;
            const float p0 = lx0[s] * scale + 0.5f, p1 = lx1[s] * scale + 0.5f, p2 = lx2[s] * scale + 0.5f;
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            const uint32_t a0 = g1 * 2654435761u, b0h = g2 * 805459861u;
            iw0[s] = w0;
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t cy = c & 1u, cz = c >> 1, k = s * 4 + c;
                const uint32_t h = (a0 + (cy ? 2654435761u : 0u)) ^ (b0h + (cz ? 805459861u : 0u));
                const uint32_t i0 = (g0 ^ h) & hmask, i1 = ((g0 + 1u) ^ h) & hmask;
                const uint32_t part = i0 >> SC_LOG2;                        // == i1 >> SC_LOG2
                const float wyz = (cy ? w1 : 1.f - w1) * (cz ? w2 : 1.f - w2);
                iva[k] = wyz * d0; ivb[k] = wyz * d1;
                ipart[k] = part;
                ipr[k] = (i0 & (SC_ENTRIES - 1)) | ((i1 & (SC_ENTRIES - 1)) << SC_LOG2);
                irank[k] = atomicAdd(&s_cnt[part], 1u);
            }
        }
        if (r == 0) SC_T(9);
        __syncthreads();
        if (r == 0) SC_T(10);
        if (threadIdx.x < 64) {                                             // exclusive scan of the round's partition counts
            uint32_t loc[4], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t idx = threadIdx.x * per + q;
                loc[q] = sum;
                if (q < per && idx < parts) sum += s_cnt[idx];
            }
            uint32_t incl = sum;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (threadIdx.x >= d) incl += o;
            }
            const uint32_t excl = incl - sum;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t idx = threadIdx.x * per + q;
                if (q < per && idx < parts) s_off[idx] = excl + loc[q];
            }
            if (threadIdx.x == 63) s_off[parts] = incl;
        }
        __syncthreads();
        if (r == 0) SC_T(11);
#pragma unroll
        for (uint32_t s = 0; s < SB_SPT; ++s) {
            if (!live[s]) continue;
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t k = s * 4 + c, q = s_off[ipart[k]] + irank[k];
                s_items[q] = make_float4(__uint_as_float(ipr[k]), iva[k], ivb[k], iw0[s]);
                s_ipart[q] = (uint8_t)ipart[k];
            }
        }
        __syncthreads();
        if (r == 0) SC_T(12);
        const uint32_t total = s_off[parts];
        for (uint32_t q = threadIdx.x; q < total; q += SB_THREADS) {
            const uint32_t p = s_ipart[q], rk = q - s_off[p] + s_base[p];
            const float4 it = s_items[q];
            if (rk < cap) {
                out[(size_t)p * nsb * cap + rk] = it;
            } else {                                                        // overfull sub-bin: scatter this pair directly
                const uint32_t pr = __float_as_uint(it.x);
                const uint32_t i0 = (pr & (SC_ENTRIES - 1)) | (p << SC_LOG2), i1 = (pr >> SC_LOG2) | (p << SC_LOG2);
                unsafeAtomicAdd(tab + 2 * (size_t)i0, (1.f - it.w) * it.y);
                unsafeAtomicAdd(tab + 2 * (size_t)i0 + 1, (1.f - it.w) * it.z);
                unsafeAtomicAdd(tab + 2 * (size_t)i1, it.w * it.y);
                unsafeAtomicAdd(tab + 2 * (size_t)i1 + 1, it.w * it.z);
            }
        }
        if (r == 0) SC_T(13);
        __syncthreads();
        if (r == 0) SC_T(14);
        for (uint32_t p = threadIdx.x; p < parts; p += SB_THREADS) s_base[p] += s_cnt[p];
    }
    SC_T(15);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < parts; p += SB_THREADS) cnt_out[(size_t)p * nsb] = min(s_base[p], cap);
}

__global__ __launch_bounds__(SC_ACC_THREADS) void k_scatter_accum2(GridMeta gm, uint32_t l_lo, uint32_t l_hi, uint32_t parts, uint32_t nsb,
                                                               const uint32_t* __restrict__ counts, const float4* __restrict__ bins,
                                                               float* __restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) double s_acc[];       // [SC_ENTRIES][2]
    const uint32_t li = blockIdx.x / parts, part = blockIdx.x % parts, l = l_lo + li;
    const uint32_t hsize = gm.off[l + 1] - gm.off[l];
    if (part >= (hsize >> SC_LOG2)) return;
    const uint32_t cap = SC_SUB_ITEMS2 / parts;
    const uint32_t* __restrict__ cnt = counts + ((size_t)li * parts + part) * nsb;
    double2* acc2 = reinterpret_cast<double2*>(s_acc);
    SC_T(16);
    __shared__ uint32_t s_fill[SC_MAX_SB];                                  // fill counts of this unit's sub-bins
    for (uint32_t e = threadIdx.x; e < nsb; e += SC_ACC_THREADS) s_fill[e] = cnt[e];
    for (uint32_t e = threadIdx.x; e < SC_ENTRIES; e += SC_ACC_THREADS) acc2[e] = make_double2(0.0, 0.0);
    __syncthreads();
    SC_T(17);
    // the unit's sub-bins: [sample block][cap] contiguous; wave w takes sub-bins w, w + W, ... in 64-item chunks
    const float4* __restrict__ src = bins + (size_t)li * nsb * SC_SUB_ITEMS2 + (size_t)part * nsb * cap;
    constexpr uint32_t U = 8, WAVES = SC_ACC_THREADS / 64;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = 0;   // next (sub-bin, chunk) of this wave: scalar
    while (s < nsb && s_fill[s] == 0) s += WAVES;
    float4 nx[U];
    bool non[U];
    auto fetch = [&]() {
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            non[u] = false;
            if (s < nsb) {
                const uint32_t fill = s_fill[s], off = c * 64u + lane;
                non[u] = off < fill;
#ifdef SC_ABL_NOLOAD
                if (non[u]) nx[u] = make_float4(__uint_as_float((off * 2654435761u) & ((1u << (2 * SC_LOG2)) - 1u)), 1.f, 2.f, 0.25f);
#else
                if (non[u]) nx[u] = src[(size_t)s * cap + off];
#endif
                ++c;
                if (c * 64u >= fill) {
                    c = 0; s += WAVES;
                    while (s < nsb && s_fill[s] == 0) s += WAVES;
                }
            }
        }
    };
    fetch();
    for (;;) {
        float4 it[U];
        bool on[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { it[u] = nx[u]; on[u] = non[u]; }
        const bool more = s < nsb;                                           // uniform per wave
        if (more) fetch();
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const uint32_t pr = __float_as_uint(it[u].x), i0 = pr & (SC_ENTRIES - 1), i1 = pr >> SC_LOG2;
            const float w0 = it[u].w, a = it[u].y, b = it[u].z;
#ifdef SC_ABL_NOATOMIC
            if (w0 * a + b == 1234.5f) s_acc[2 * i0 + i1] = 1.0;
#else
            atomicAdd(&s_acc[2 * i0], (double)((1.f - w0) * a));
            atomicAdd(&s_acc[2 * i0 + 1], (double)((1.f - w0) * b));
            atomicAdd(&s_acc[2 * i1], (double)(w0 * a));
            atomicAdd(&s_acc[2 * i1 + 1], (double)(w0 * b));
#endif
        }
        if (!more) break;
    }
    SC_T(18);
    __syncthreads();
    SC_T(19);
    float2* __restrict__ dst = reinterpret_cast<float2*>(grad_table + 2 * ((size_t)gm.off[l] + (size_t)part * SC_ENTRIES));
    constexpr uint32_t F = SC_ENTRIES / SC_ACC_THREADS;                         // all F loads in flight before the first add
    float2 t[F];
#pragma unroll
    for (uint32_t k = 0; k < F; ++k) t[k] = dst[k * SC_ACC_THREADS + threadIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < F; ++k) {
        const double2 a = acc2[k * SC_ACC_THREADS + threadIdx.x];
        t[k].x += (float)a.x; t[k].y += (float)a.y;
        dst[k * SC_ACC_THREADS + threadIdx.x] = t[k];
    }
    SC_T(20);
}

__global__ __launch_bounds__(256) void k_reduce_replicas(const float4* __restrict__ rep, uint32_t n_rep, uint32_t stride4, uint32_t count4,
                                                         float4* __restrict__ grad_table) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= count4) return;
    float4 t = grad_table[e];
    for (uint32_t r = 0; r < n_rep; ++r) {                                  // fixed order
        const float4 a = rep[(size_t)r * stride4 + e];
        t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    grad_table[e] = t;
}

static int scatter_env(const char* name, int dflt);
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
// XR_HG_COST="c0,c1,...": measured single-level times (tools/microbench_fwd3.py prints the line); fewer values than levels: the
// last one repeats.  Default: the Lego geometry's figures at 2.6e5 ray-ordered samples on the MI355X.
static void hg_level_costs(float* cost, int n_levels, uint32_t hashed_mask) {
    static float env_cost[EN_MAX_LEVELS];
    static int n_env = -1;
    if (n_env < 0) {
        n_env = 0;
        const char* e = getenv("XR_HG_COST");
        while (e && *e && n_env < EN_MAX_LEVELS) {
            char* end = nullptr;
            const float v = strtof(e, &end);
            if (end == e) break;
            env_cost[n_env++] = v;
            e = *end == ',' ? end + 1 : end;
        }
    }
    // measured (profiles/r03_microbench_fwd3.txt): a level alone on one XCD, 2.6e5 ray-ordered samples, us: dense levels 23-24
    // (bound by the texture-address rate of their 13 lane accesses per sample, all cache hits), hashed levels 27.6, 32.0, 38.9,
    // 47.3, 50.5, 51.4, 51.6 ... from the coarsest on (the coarse ones still share lines between neighbouring samples)
    static const float hashed_cost[] = {27.6f, 32.0f, 38.9f, 47.3f, 50.5f, 51.4f, 51.7f};
    int h = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (n_env > 0) cost[l] = env_cost[l < n_env ? l : n_env - 1];
        else if ((hashed_mask >> l) & 1) { cost[l] = hashed_cost[h < 6 ? h : 6]; ++h; }
        else cost[l] = 23.6f;
    }
}

extern "C" int xr_hashgrid_fwd2(const float* table, const float* x, uint32_t x_stride, uint32_t x_comp_stride, uint32_t n, const uint32_t* n_dev,
                                const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                const uint32_t* offset_host, float* enc_t, uint32_t ld, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(table && x && enc_t, "null pointer");
    XR_REQUIRE(ld >= n && x_comp_stride >= 1 && (x_comp_stride > 1 ? x_stride >= 1 : x_stride >= 3), "bad stride");
    XR_REQUIRE(((uintptr_t)table & 15) == 0, "table must be 16-byte aligned");
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    hipStream_t stream = (hipStream_t)stream_;
    // XR_HG_FWD_MODE (measurement switches, read once): bit 5 (32) = explicit cost-balanced XCD map (order 5), bit 4 (16) = two-list
    // balanced XCD mapping (order 4), bit 0 = cost-weighted mapping (order 3), none of them = level-major (order 1); bit 1 =
    // non-temporal table loads at the hashed levels, bit 2 = coarsest levels from LDS, bit 3 = 16-B pair gathers; XR_HG_WSH =
    // "a,b,c": order 3's log2 cost of a sample block at dense levels < 2^16 entries, larger dense levels, hashed levels.
    // Round 2 (profiles/r02_microbench_fwd_variants.txt, 2^18 ray-ordered samples): 8: 88.2 us, 24: 88.6, 0: 92.8, 16: 96.8,
    // 20: 101.6, 18: 254.0.
    // Round 3 (profiles/r03_microbench_fwd3.txt): 8: 91.2 us, 0: 96.8, 40 (map from the measured per-level costs + pair gathers):
    // 83.3, 32 (map, plain gathers): 86.5; with positions as three planes (x_comp_stride > 1): 82.9 / 85.0 / 76.1 / 77.6.
    // Default 40.
    static const int mode = scatter_env("XR_HG_FWD_MODE", 40);
    static int wsh3[3] = {-1, 0, 0};
    if (wsh3[0] < 0) {
        int a = 0, b = 1, c = 2;
        const char* e = getenv("XR_HG_WSH");
        if (e) sscanf(e, "%d,%d,%d", &a, &b, &c);
        wsh3[1] = b; wsh3[2] = c; wsh3[0] = a;
    }
    const uint32_t nsb = xr_div_up(n, EN_BLOCK);
    gm.n_sblocks = nsb;
    gm.nt = (mode >> 1) & 1;
    gm.pairs = (mode >> 3) & 1;
    gm.l_min = 0;
    XcdMap xm;
    memset(&xm, 0, sizeof(xm));
    static const uint32_t lds_min_n = (uint32_t)scatter_env("XR_HG_LDS_MIN_N", 32768);
    if ((mode & 4) && n >= lds_min_n && x_comp_stride == 1 && !(mode & 32)) {
        // levels whose slices fit the LDS together (dense, contiguous from level 0): at most 144 KiB
        int n_lds = 0;
        while (n_lds < n_levels && !((hm >> n_lds) & 1) && (size_t)gm.off[n_lds + 1] * 8 <= 144u * 1024u) ++n_lds;
        if (n_lds > 0 && n_lds < n_levels) {
            const size_t lds = (size_t)gm.off[n_lds] * 8;
            static size_t attr_lds = 0;
            if (lds > attr_lds) {
                XR_HIP(hipFuncSetAttribute((const void*)k_hashgrid_fwd_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_lds = lds;
            }
            const uint32_t g = xr_div_up(n, EN_LDS_THREADS);
            hipLaunchKernelGGL(k_hashgrid_fwd_lds, dim3(g < 256u ? g : 256u), dim3(EN_LDS_THREADS), lds, stream, gm, n_lds, table, x,
                               x_stride, n, n_dev, rows, enc_t, ld);
            XR_LAUNCH_CHECK();
            gm.l_min = n_lds;
        }
    }
    uint32_t blocks = 0;
    if (mode & 32) {
        float cost[EN_MAX_LEVELS];
        hg_level_costs(cost, n_levels, hm);
        const uint32_t per_xcd = hg_build_map(&xm, cost, n_levels, nsb);
        if (per_xcd) { gm.order = 5; blocks = 8 * per_xcd; }
    }
    if (blocks) {
    } else if (mode & 16) {
        // hashed levels must be the top of the level range (they are, for a growing resolution)
        uint32_t nh = 0;
        while (nh < (uint32_t)(n_levels - gm.l_min) && ((hm >> (n_levels - 1 - nh)) & 1)) ++nh;
        gm.order = 4;
        gm.n_hashed = nh;
        const uint32_t nl = (uint32_t)(n_levels - gm.l_min);
        blocks = 8 * (xr_div_up((uint64_t)nh * nsb, 8) + xr_div_up((uint64_t)(nl - nh) * nsb, 8) + 2);
    } else if (mode & 1) {
        gm.order = 3;
        gm.wsum = 0;
        for (int l = 0; l < n_levels; ++l) {
            const uint32_t hsize = gm.off[l + 1] - gm.off[l];
            gm.wsh[l] = (uint8_t)(((hm >> l) & 1) ? wsh3[2] : (hsize < 65536u ? wsh3[0] : wsh3[1]));
            if (l >= gm.l_min) gm.wsum += 1u << gm.wsh[l];
        }
        blocks = 8 * hg_balanced_blocks_per_xcd(gm, nsb);
    } else {
        XR_REQUIRE(gm.l_min == 0, "the LDS path needs a balanced mapping");
        blocks = 8 * ((n_levels + 7) / 8) * nsb;
    }
    hipLaunchKernelGGL(k_hashgrid_fwd, dim3(blocks), dim3(EN_BLOCK), 0, stream, gm, xm, hm, table, x, x_stride, x_comp_stride, n,
                       n_dev, rows, enc_t, ld);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_hashgrid_fwd(const float* table, const float* x, uint32_t x_stride, uint32_t n, const uint32_t* n_dev,
                               const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                               const uint32_t* offset_host, float* enc_t, uint32_t ld, void* stream_) {
    XR_REQUIRE(x_stride >= 3, "bad stride");
    return xr_hashgrid_fwd2(table, x, x_stride, 1, n, n_dev, rows, n_levels, scale_host, resolution_host, offset_host, enc_t, ld, stream_);
}

static int scatter_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int sc_mode() {          // XR_SC_MODE: 2 = third generation (xr_scatter.hip, default); 1 / 0 = the second / first bin /
    static const int m = scatter_env("XR_SC_MODE", 2);   // accumulate pair with the atomic kernel for the dense levels (measurement)
    return m;
}
// Which levels take which path, and the workspace layout:  [fill counts][bins][replicas of the dense slices]
#define SC_REPLICAS 8
struct ScatterPlan {
    int l_bin;            // levels [l_bin, n_levels) are binned, [0, l_bin) take the atomic kernel
    uint32_t parts, nsb;
    size_t counts_bytes, bins_bytes, rep_bytes;
    uint32_t rep_stride;  // floats per replica
    uint32_t rep_levels;  // levels [0, rep_levels) of the dense remainder go through the replicas
};
static ScatterPlan scatter_plan(uint32_t n, int n_levels, const uint32_t* res, const uint32_t* off, uint32_t hashed_mask) {
    static const int cap_levels = scatter_env("XR_SCAN_LEVELS", EN_MAX_LEVELS);
    static const int use_rep = scatter_env("XR_REPLICAS", 1);
    ScatterPlan p{};
    p.l_bin = n_levels; p.parts = 1; p.nsb = xr_div_up(n, SC_BLOCK_SAMPLES);
    if (n >= 16384u && p.nsb <= SC_MAX_SB) {
        while (p.l_bin > 0 && n_levels - p.l_bin < cap_levels) {
            const int l = p.l_bin - 1;
            const uint32_t hsize = off[l + 1] - off[l];
            if (!((hashed_mask >> l) & 1) || (hsize & (hsize - 1)) || hsize < SC_ENTRIES || (hsize >> SC_LOG2) > SC_MAX_PARTS ||
                res[l] >= SC_ENTRIES || (off[l] & 1)) break;
            p.parts = p.parts > (hsize >> SC_LOG2) ? p.parts : (hsize >> SC_LOG2);
            --p.l_bin;
        }
    }
    const uint32_t nl = (uint32_t)(n_levels - p.l_bin);
    p.counts_bytes = (((size_t)nl * p.parts * p.nsb * sizeof(uint32_t)) + 255) & ~(size_t)255;
    p.bins_bytes = (size_t)nl * p.nsb * (sc_mode() ? SC_SUB_ITEMS2 : SC_SUB_ITEMS) * sizeof(float4);
    // replicas only for a dense remainder next to a binned range (small tables: <= 2^14-entry... up to 2^19 each)
    // XR_REPLICA_LEVELS: how many of the coarsest levels are replicated (default: all of the dense remainder).  Measured in the
    // bench's training loop (entry span): all 5 -> 151 us, 3 -> 149, 2 -> 147, 1 -> 156, 0 -> 157: flat, the default stays
    static const int rep_lv = scatter_env("XR_REPLICA_LEVELS", EN_MAX_LEVELS);
    p.rep_levels = (uint32_t)(rep_lv < p.l_bin ? (rep_lv < 0 ? 0 : rep_lv) : p.l_bin);
    p.rep_stride = (use_rep && n >= 16384u && p.rep_levels > 0) ? (2u * off[p.rep_levels] + 3u) & ~3u : 0u;
    p.rep_bytes = (size_t)SC_REPLICAS * p.rep_stride * sizeof(float);
    return p;
}

extern "C" size_t xr_hashgrid_bwd_workspace_bytes(uint32_t n, int n_levels, const uint32_t* resolution_host, const uint32_t* offset_host) {
    if (n_levels < 1 || n_levels > EN_MAX_LEVELS || !resolution_host || !offset_host) return 0;
    GridMeta gm; uint32_t hm;
    float dummy[EN_MAX_LEVELS] = {0};
    if (fill_meta(&gm, &hm, n_levels, dummy, resolution_host, offset_host) != 0) return 0;
    if (sc_mode() == 2) return xr_scatter3_workspace_bytes(n, gm, hm);
    const ScatterPlan p = scatter_plan(n, n_levels, gm.res, gm.off, hm);
    return p.counts_bytes + p.bins_bytes + p.rep_bytes;
}

static int hashgrid_bwd_gen12(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                               const uint32_t* rows, int n_levels,
                               const float* scale_host, const uint32_t* resolution_host, const uint32_t* offset_host,
                               float* grad_table, void* workspace, size_t workspace_bytes, void* stream_) {
    GridMeta gm; uint32_t hm;
    XR_REQUIRE(fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) == 0, "bad level metadata");
    hipStream_t stream = (hipStream_t)stream_;
    // Levels [l_bin, n_levels): hashed, power-of-two slices of 2^14..2^21 entries, resolution < 2^14 -> bin +
    // LDS accumulate (no global atomics).  Levels [0, l_bin): dense / small -> atomic scatter with run-length
    // merging, into SC_REPLICAS replicas of the slice when a workspace is there.  Without a workspace, or for
    // small n (fixed cost: 2 x 128 KiB of LDS traffic per partition), every level takes the plain atomic kernel.
    // XR_SCAN_LEVELS=0 / XR_REPLICAS=0 switch the two mechanisms off (measurement).
    ScatterPlan p = scatter_plan(n, n_levels, gm.res, gm.off, hm);
    const bool ws_ok = workspace && ((uintptr_t)grad_table & 15) == 0 && ((uintptr_t)workspace & 15) == 0;
    if (!ws_ok) { p.l_bin = n_levels; p.rep_stride = 0; p.rep_levels = 0; }
    else XR_REQUIRE(workspace_bytes >= p.counts_bytes + p.bins_bytes + p.rep_bytes, "workspace too small");
    // The dense remainder and the binned range are independent (disjoint table slices, read-only inputs) and bound
    // by different things (atomics vs item stores / LDS): when both exist the remainder runs on an internal helper
    // stream, forked from and joined back into the caller's stream with events (created once per process;
    // XR_SCATTER_OVERLAP=0 keeps everything on the caller's stream).  Measured and rejected: also splitting the
    // binned levels into 2-3 concurrent bin/accumulate groups (0.283 -> 0.30 ms).
    static const int overlap = scatter_env("XR_SCATTER_OVERLAP", 1);
    static hipStream_t aux = nullptr;
    static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool forked = false;
    // XR_SCATTER_ORDER=1: the dense remainder is forked AFTER the binning kernel, i.e. it runs beside the LDS-bound accumulate
    // kernel instead of beside the store-bound binning kernel.  Measured in the bench's training loop: 150 vs 153 us entry span,
    // no difference -- the default stays 0 (forked first).
    static const int order = scatter_env("XR_SCATTER_ORDER", 0);
    auto launch_dense = [&]() -> int {
    if (p.l_bin > 0) {
        hipStream_t ds = stream;
        if (overlap && p.l_bin < n_levels) {
            if (!aux) {
                XR_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
                XR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
                XR_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
            }
            XR_HIP(hipEventRecord(ev_fork, stream));
            XR_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
            ds = aux;
            forked = true;
        }
        GridMeta gd = gm;
        gd.n_levels = p.l_bin;
        if (p.l_bin < n_levels || (p.l_bin & 7)) gd.order = 2;   // a partial level range is spread over all XCDs
        const uint32_t per_xcd = (p.l_bin + 7) / 8;
        const uint32_t bw_ch = BW_CH;      // swept 8..128 on MI355X: 0.71-0.75 ms at 2^18 dense-gradient samples, flat
        gd.n_sblocks = xr_div_up(n, bw_ch * (EN_BLOCK / 16));
        const uint32_t blocks = (gd.order == 2 ? (uint32_t)p.l_bin : 8 * per_xcd) * gd.n_sblocks;
        float* rep = p.rep_stride ? (float*)((char*)workspace + p.counts_bytes + p.bins_bytes) : nullptr;
        if (rep) XR_HIP(hipMemsetAsync(rep, 0, p.rep_bytes, ds));
        hipLaunchKernelGGL(k_hashgrid_bwd, dim3(blocks), dim3(EN_BLOCK), 0, ds, gd, hm, x, x_stride, denc_t, ld,
                           n, n_dev, rows, grad_table, rep, (uint32_t)SC_REPLICAS, p.rep_stride, p.rep_levels);
        XR_LAUNCH_CHECK();
        if (rep) {
            const uint32_t count4 = p.rep_stride / 4;
            hipLaunchKernelGGL(k_reduce_replicas, dim3(xr_div_up(count4, 256)), dim3(256), 0, ds, (const float4*)rep,
                               (uint32_t)SC_REPLICAS, count4, count4, (float4*)grad_table);
            XR_LAUNCH_CHECK();
        }
        if (forked) XR_HIP(hipEventRecord(ev_join, ds));
    }
    return XR_OK;
    };
    if (order == 0 || p.l_bin >= n_levels) { const int rc = launch_dense(); if (rc != XR_OK) return rc; }
    if (p.l_bin < n_levels) {
        const uint32_t nl = (uint32_t)(n_levels - p.l_bin);
        static bool attr_set = false;
        if (!attr_set) {
            XR_HIP(hipFuncSetAttribute((const void*)k_scatter_accum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES));
            XR_HIP(hipFuncSetAttribute((const void*)k_scatter_accum2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES));
            attr_set = true;
        }
        uint32_t* counts = (uint32_t*)workspace;
        float4* bins = (float4*)((char*)workspace + p.counts_bytes);
        if (sc_mode()) {
            hipLaunchKernelGGL(k_scatter_bin2, dim3(nl * p.nsb), dim3(SB_THREADS), 0, stream, gm, (uint32_t)p.l_bin, (uint32_t)n_levels,
                               p.parts, p.nsb, x, x_stride, denc_t, ld, n, n_dev, rows, counts, bins, grad_table);
            XR_LAUNCH_CHECK();
            if (order != 0) { const int rc = launch_dense(); if (rc != XR_OK) return rc; }
            hipLaunchKernelGGL(k_scatter_accum2, dim3(nl * p.parts), dim3(SC_ACC_THREADS), SC_LDS_BYTES, stream, gm, (uint32_t)p.l_bin,
                               (uint32_t)n_levels, p.parts, p.nsb, counts, bins, grad_table);
            XR_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(k_scatter_bin, dim3(nl * p.nsb), dim3(SC_THREADS), 0, stream, gm, (uint32_t)p.l_bin, (uint32_t)n_levels,
                               p.parts, p.nsb, x, x_stride, denc_t, ld, n, n_dev, rows, counts, bins, grad_table);
            XR_LAUNCH_CHECK();
            if (order != 0) { const int rc = launch_dense(); if (rc != XR_OK) return rc; }
            hipLaunchKernelGGL(k_scatter_accum, dim3(nl * p.parts), dim3(SC_ACC_THREADS), SC_LDS_BYTES, stream, gm, (uint32_t)p.l_bin,
                               (uint32_t)n_levels, p.parts, p.nsb, counts, bins, grad_table);
            XR_LAUNCH_CHECK();
        }
    }
    if (forked) XR_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    return XR_OK;
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
                           n, n_dev, rows, grad_table, (float*)nullptr, 0u, 0u, 0u);
        XR_LAUNCH_CHECK();
        l = e;
    }
    return XR_OK;
}

extern "C" int xr_hashgrid_bwd2(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
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
    if (sc_mode() != 2) {
        if (overwrite) XR_HIP(hipMemsetAsync(grad_table + 2 * (size_t)gm.off[0], 0, 2 * (size_t)(gm.off[n_levels] - gm.off[0]) * sizeof(float), stream));
        return hashgrid_bwd_gen12(x, x_stride, denc_t, ld, n, n_dev, rows, n_levels, scale_host, resolution_host, offset_host, grad_table,
                                  workspace, workspace_bytes, stream_);
    }
    // levels without a non-atomic path (tiny / oddly shaped tables, small n) take the atomic kernel -- beside the binned levels on a
    // helper stream when both exist (disjoint table slices)
    uint32_t amask = xr_scatter3_atomic_mask(n, gm, hm, workspace && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)grad_table & 15) == 0);
    const uint32_t all = (1u << n_levels) - 1u;
    static hipStream_t aux = nullptr;
    static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const bool fork = amask != 0 && amask != all;
    if (fork) {
        if (!aux) {
            XR_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
            XR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            XR_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
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

extern "C" int xr_hashgrid_bwd_adam_supported(uint32_t n, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                              const uint32_t* offset_host) {
    GridMeta gm; uint32_t hm;
    if (!scale_host || !resolution_host || !offset_host || n == 0 || sc_mode() != 2) return 0;
    if (fill_meta(&gm, &hm, n_levels, scale_host, resolution_host, offset_host) != 0) return 0;
    return xr_scatter3_atomic_mask(n, gm, hm, true) == 0 ? 1 : 0;
}
extern "C" int xr_hashgrid_bwd_adam(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                                    const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                                    const uint32_t* offset_host, void* workspace, size_t workspace_bytes, const xr_adam_fuse* adam,
                                    void* stream_) {
    XR_REQUIRE(x && denc_t && scale_host && resolution_host && offset_host && workspace && adam, "null pointer");
    XR_REQUIRE(adam->param && adam->m && adam->v && adam->step >= 1, "bad optimiser state");
    XR_REQUIRE((((uintptr_t)adam->param | (uintptr_t)adam->m | (uintptr_t)adam->v | (uintptr_t)adam->ema | (uintptr_t)workspace) & 15) == 0,
               "buffers must be 16-byte aligned");
    XR_REQUIRE(n > 0 && x_stride >= 3 && ld >= n, "bad sizes");
    XR_REQUIRE(!rows || n_dev, "a row list comes with its device-side length (n_dev)");
    XR_REQUIRE(xr_hashgrid_bwd_adam_supported(n, n_levels, scale_host, resolution_host, offset_host),
               "a level of this geometry / row count has no non-atomic path (or XR_SC_MODE != 2): scatter and step separately");
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

extern "C" int xr_hashgrid_bwd(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                               const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                               const uint32_t* offset_host, float* grad_table, void* workspace, size_t workspace_bytes, void* stream_) {
    return xr_hashgrid_bwd2(x, x_stride, denc_t, ld, n, n_dev, rows, n_levels, scale_host, resolution_host, offset_host, grad_table, workspace,
                            workspace_bytes, 0, stream_);
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
