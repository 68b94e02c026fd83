// Third generation of the hash-grid scatter (backward of tcnn's HashGrid encoding as called from
// /root/reference/xrnerf/models/mlps/hashnerf_mlp.py:34-37,59-61): EVERY level without a global atomic, three launches on
// the caller's stream (round 6: no helper stream, no event), results independent of the launch schedule BIT FOR BIT (round 6:
// the LDS sums are 64-bit fixed point, see S3_FIX below; fp64 until round 5 -- schedule-dependent in the last bit).
//
// What round 2 measured on the second generation (profiles/r02_scatter_phase_timing.txt, profiles/NOTES_r01_r03.md 5b):
//   * the dense levels' atomic kernel (55 us alone) beside the bin / accumulate pair stretches that pair from 39 + 56 to
//     73 + 64 us -- the overlap costs what it hides;
//   * inside k_scatter_accum2 the item stream (11.5 us per workgroup) and the LDS atomics (12 us) do NOT overlap (23.8 us
//     together): the loop reads the sub-bin fill counts from LDS, LDS operations retire in order, so the read -- and the
//     global loads of the next items behind it -- waits for every queued atomic;
//   * the accumulate kernel reads the 46 MB of the hashed levels' gradient slices only to add to the zeros a 48.8-MB
//     zero-fill wrote a few kernels earlier.
// Here:
//   kind H  hashed levels (power-of-two slices >= 2^13 entries): bin by bits [13, ..) of y*P1 ^ z*P2 as before;
//   kind D  dense levels above 2^16 entries: the same bin / accumulate pair, partition = (grid row y + z res) mod P with P = 32
//           or 64 -- neighbouring rows land in different partitions, so the load is as even as a hash's, and the two
//           x-neighbours of a pair always share a row (pairs that leave the res^3 lattice at the domain's upper faces, where
//           tcnn's linear index runs into the next row or wraps, become two single-entry items);
//   kind R  dense levels up to 2^16 entries (levels 0-2 of the Lego geometry: 4 096 + 12 168 + 29 792 entries, where a
//           ray's 20 samples fall into two or three cells): one THREAD walks 16 consecutive rows with the 16 corner sums of
//           the current cell in registers and flushes them into a workgroup-private LDS copy of its 2^13-entry
//           partition on a cell change (run-length reduction: ~10x fewer LDS atomics, lanes of a wave sit on different
//           rays, so no same-address serialisation); per-chunk partials are folded in fixed order by k_scatter_fold;
//   items that do not fit their sub-bin (capacity 1.5x the expected fill) go to the workgroup's private overflow list and
//   are picked up by the accumulate workgroup of their partition -- any input stays correct, nothing ever touches the
//   table with an atomic, and the table slice can therefore be WRITTEN instead of added to (flag XR_SCATTER_OVERWRITE:
//   the training step drops its 48.8-MB zero-fill);
//   the accumulate loop keeps its sub-bin fill counts in a VGPR (lane i of wave w: sub-bin w + 16 i, read with
//   v_readlane): no LDS read sits between the item loads and the returnless LDS atomics.
#include "xr_hashgrid.h"
#include "xr_scatter.h"
#include "xr_adam.h"
#include "xr_aux.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- indices that name
// registers of a per-thread array must be constants for the array to stay out of scratch memory
template <typename F, int... I>
__device__ __forceinline__ void s3_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void s3_static_for(F&& f) { s3_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef S3_LOG2
#define S3_LOG2 13      // (tools/build_variant.sh lg12 xr_scatter.hip "-DS3_LOG2=12": measured, slower)
#endif
#define S3_ENTRIES (1u << S3_LOG2)
#define S3_LDS_BYTES (S3_ENTRIES * 2 * sizeof(double))
#define S3_MAX_PARTS 256
#define S3_MAX_SB 1024
#ifndef S3_BIN_THREADS
#define S3_BIN_THREADS 512
#endif
#define S3_ACC_THREADS 1024
#define S3_ACC_WAVES (S3_ACC_THREADS / 64)
#ifndef S3_SPT
#define S3_SPT 2                                     // samples per thread and binning round
#endif
#define S3_ROUND_ITEMS (4 * S3_BIN_THREADS * S3_SPT)   // items staged in LDS per binning round (4 per sample)
#ifndef S3_R_MAX_ENTRIES
#define S3_R_MAX_ENTRIES 65536u                      // dense levels up to this size take the run-length kernel
#endif
#define S3_R_ROWS 16                                 // consecutive rows per thread there

enum { S3_H = 0, S3_D = 1 };
#ifndef S3_PERMUTE
#define S3_PERMUTE 1
#endif
#ifndef S3_MERGE
#define S3_MERGE S3_PERMUTE
#endif

#ifdef S3_TIMING          // tools/scatter3_timing.hip: wall_clock64 (100 MHz) stamps of a few accumulate workgroups
#define S3_T_BLOCKS 8
__device__ long long g_s3_t[S3_T_BLOCKS][8];
#define S3_T(k) do { if (threadIdx.x == 0 && (blockIdx.x % 96u) == 0u && blockIdx.x / 96u < S3_T_BLOCKS) g_s3_t[blockIdx.x / 96u][k] = wall_clock64(); } while (0)
#else
#define S3_T(k)
#endif

struct S3Level {
    float scale;
    uint32_t res, hsize, toff;       // table offset in entries
    uint32_t kind, parts, plog2;     // partitions (kind D: a power of two, plog2 its log2)
    uint32_t cap;                    // items per sub-bin
    uint32_t drow;                   // row of the level's feature 0 in denc_t
    uint32_t counts_off;             // words into counts: [part][nsb]
    uint32_t acc_block0;             // first workgroup of this level in the accumulate grid
    uint32_t pad_;
    uint64_t bins_off;               // items into bins: [part][sb][cap]
    uint64_t ovf_off;                // records into the overflow area: [sb][ovf_cap]
};
struct S3Plan {
    S3Level lv[EN_MAX_LEVELS];
    uint32_t n_lv, nsb, ovf_cap, overwrite;
    uint32_t ovfcnt_off;             // words into counts: [lv][nsb]
    uint32_t acc_blocks;
    uint32_t lg;                     // log2 of the entries of one partition (13, or 12: two accumulate workgroups per CU)
    uint32_t fuse;                   // != 0: the accumulate kernel applies Adam to the entries instead of writing their gradient
    uint32_t amax_off, ramax_off;    // words into counts: [lv][nsb] / [run-length lv][nsb] -- max |dL/d feature| per (level, sample block), as bits
    uint32_t n_rl, n_bound;          // run-length levels (their feature-0 rows in rl_drow), an upper bound of the row count
    uint32_t rl_slices;              // slots per (run-length level, sample block) in the maxima: ceil(n_lv / n_rl) (level r uses the first kr of them)
    uint32_t rl_drow[EN_MAX_LEVELS];
    XrAdamArgs ad;
};
struct S3RLevel {
    float scale;
    uint32_t res, hsize, toff, parts, drow;
    uint32_t poff;                   // entries before this level in a partial slab
    uint32_t block0;                 // first workgroup of this level: [part][chunk]
};
struct S3RPlan {
    S3RLevel lv[EN_MAX_LEVELS];
    uint32_t n_lv, chunks, slab_entries, overwrite, blocks;
    uint32_t nsb, ramax_off, n_bound;  // as in S3Plan (the binning launch writes the maxima, the run-length workgroups read them)
    uint32_t rl_slices, n_bin_lv;      // slots per (level, sample block) and the number of binned levels (0: one slot, k_scatter_rl_absmax)
    uint32_t fuse;                   // != 0: the fold kernel applies Adam instead of writing the gradient
    XrAdamArgs ad;
};

// index % hsize of a dense level: inside the unit cube the index is below 2 hsize (x + y res + z res^2 with coordinates <= res)
__device__ __forceinline__ uint32_t s3_wrap(uint32_t i, uint32_t hsize) {
    if (i >= hsize) { i -= hsize; if (i >= hsize) i %= hsize; }
    return i;
}
// kind D: entry idx of a dense level -> (partition, local entry).  Row r = idx / res of the res^3 lattice belongs to partition
// r mod P and sits at local row r / P; the (at most 7) padding entries behind the lattice close partition 0.
__device__ __forceinline__ uint32_t s3_dense_rows(uint32_t res, uint32_t plog2, uint32_t part) {
    return (res * res + (1u << plog2) - 1u - part) >> plog2;                // rows r < res^2 with r mod P == part
}
__device__ __forceinline__ void s3_dense_local(uint32_t idx, uint32_t res, uint32_t plog2, uint32_t pmask, uint32_t* part, uint32_t* e) {
    const uint32_t cells = res * res * res;
    if (idx < cells) {
        const uint32_t r = idx / res, xx = idx - r * res;
        *part = r & pmask; *e = (r >> plog2) * res + xx;
    } else {
        *part = 0u; *e = s3_dense_rows(res, plog2, 0u) * res + (idx - cells);
    }
}
#ifdef __HIP_EMU__
__device__ inline uint32_t s3_readlane(uint32_t v, uint32_t lane) { return __shfl(v, (int)lane); }
#else
__device__ inline uint32_t s3_readlane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
#endif

// ---- Fixed-point accumulators (round 6).  The LDS sums were fp64 (ds_add_f64) until round 5: exact enough, but their last bit depended
// on the order the atomics arrived in (one ulp, rarely).  Now every contribution is rounded ONCE to a multiple of 2^-S and summed as a
// 64-bit integer (ds_add_u64: 4.7 ns per wave instruction against 8.6, tools/lds_probe.hip): integer addition is associative, so a
// launch's result is the same bits whatever the schedule.  S is chosen per level from max |dL/d feature| of the launch (the binning
// workgroups record it per sample block; every contribution is a weight in [0, 1] times such a value) and the row count n: an entry
// receives at most 8 n contributions, so with max < 2^e and 8 n <= 2^L, S = min(62 - L, 46) - e keeps every sum below 2^63 -- for
// n = 2^18 that is a quantum of 2^-41 of the level's largest gradient (fp32 carries 2^-24 of each value).  A non-finite gradient
// makes the level's entries NaN (what the fp64 sums did to the entries it touched).
#ifndef S3_FIX
#define S3_FIX 1
#endif
struct S3Scale { double mul, inv; bool bad; };
__device__ __forceinline__ S3Scale s3_scale(uint32_t max_bits, uint32_t n_bound) {
    S3Scale sc;
    float m; __builtin_memcpy(&m, &max_bits, 4);
    sc.bad = !(m <= 3.4028235e38f);
    const int e = (m > 0.f && !sc.bad) ? ilogbf(m) + 1 : 0;                       // m < 2^e
    const int L = 64 - __builtin_clzll(8ull * (unsigned long long)(n_bound ? n_bound : 1u));   // 8 n < 2^L
    const int S = (62 - L < 46 ? 62 - L : 46) - e;                                // (<= 46 - e: a value handed to s3_q -- a contribution, or a run-length thread's sum of <= 16 of them -- stays below 2^50)
    sc.mul = ldexp(1.0, S); sc.inv = ldexp(1.0, -S);
    return sc;
}
#if S3_FIX
typedef long long s3_acc_t;
// round(x 2^S) as an integer: |x 2^S| < 2^50 (s3_scale), so adding 1.5 * 2^52 leaves the (round-to-nearest-even) integer in the sum's low mantissa
// bits -- one FMA and a 64-bit subtraction instead of the ~12 instructions of a double -> int64 conversion
__device__ __forceinline__ s3_acc_t s3_q(float x, const S3Scale& sc) {
    const double t = fma((double)x, sc.mul, 6755399441055744.0);
    long long b; __builtin_memcpy(&b, &t, 8);
    return b - 0x4338000000000000LL;
}
__device__ __forceinline__ void s3_lds_add(double* s_acc, uint32_t i, s3_acc_t v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(s_acc) + i, (unsigned long long)v);
}
__device__ __forceinline__ float s3_val(double cell, const S3Scale& sc) {
    long long v; __builtin_memcpy(&v, &cell, 8);
    return sc.bad ? __builtin_nanf("") : (float)((double)v * sc.inv);
}
#else
typedef double s3_acc_t;
__device__ __forceinline__ s3_acc_t s3_q(float x, const S3Scale&) { return (double)x; }
__device__ __forceinline__ void s3_lds_add(double* s_acc, uint32_t i, s3_acc_t v) { atomicAdd(&s_acc[i], v); }
__device__ __forceinline__ float s3_val(double cell, const S3Scale&) { return (float)cell; }
#endif
// max over the nsb per-sample-block maxima of one level: every wave on its own (a few words; positive floats order like their bits)
__device__ __forceinline__ uint32_t s3_level_max(const uint32_t* __restrict__ slots, uint32_t nsb) {
    uint32_t vm = 0;
    for (uint32_t q = threadIdx.x & 63u; q < nsb; q += 64u) vm = max(vm, slots[q]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vm = max(vm, (uint32_t)__shfl_xor((int)vm, d, 64));
    return vm;
}
__device__ __forceinline__ uint32_t s3_abs_bits(float v) { uint32_t b; __builtin_memcpy(&b, &v, 4); return b & 0x7fffffffu; }

// ------------------------------------------------------------------------------------------------ binning
// One workgroup = (level, block of BS samples).  Per round of <= 1024 samples: items ranked per partition with LDS counters,
// placed in LDS in partition order, copied out as contiguous runs (full-line stores).  The inputs of round r + 1 are
// fetched before round r is ranked and copied out.
template <int KIND> struct S3Bin {
    static constexpr int SPT = S3_SPT;                          // samples per thread and round
    static constexpr int IPS = KIND == S3_H ? 4 : 8;            // items per sample, worst case (kind D: every pair split)
    static constexpr int ROUND = S3_BIN_THREADS * SPT;
};

template <int KIND, int BS>
__device__ __forceinline__ void s3_bin_block(const S3Level& L, uint32_t lg, uint32_t nsb, uint32_t ovf_cap, uint32_t sb, const float* __restrict__ x,
                                             uint32_t x_stride, const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                             const uint32_t* __restrict__ rows, uint32_t* __restrict__ cnt_out,
                                             uint32_t* __restrict__ ovfcnt_out, uint32_t* __restrict__ amax_out, float4* __restrict__ bins, float4* __restrict__ ovf,
                                             float4* s_items, uint8_t* s_ipart, uint32_t* s_cnt, uint32_t* s_off, uint32_t* s_base,
                                             uint32_t* s_ovf) {
    using B = S3Bin<KIND>;
    constexpr int SPT = B::SPT, IPS = B::IPS, NI = SPT * IPS, ROUND = B::ROUND, ROUNDS = BS / ROUND;
    static_assert(ROUND * 4 <= S3_ROUND_ITEMS, "a round's expected items must fit the LDS staging area");
    const uint32_t parts = L.parts, cap = L.cap, b0 = sb * BS;
    for (uint32_t p = threadIdx.x; p < parts; p += S3_BIN_THREADS) s_base[p] = 0;
    if (threadIdx.x == 0) *s_ovf = 0;
    const float scale = L.scale;
    const uint32_t res = L.res, hsize = L.hsize, hmask = hsize - 1u, plog2 = L.plog2, pmask = parts - 1u, emask = (1u << lg) - 1u;
    const float* __restrict__ d0p = denc_t + (size_t)L.drow * ld;
    const float* __restrict__ d1p = d0p + ld;
    float4* __restrict__ out = bins + L.bins_off + (size_t)sb * cap;         // [part][sb][cap]
    float4* __restrict__ ovo = ovf + L.ovf_off + (size_t)sb * ovf_cap;
    const uint32_t per = (parts + 63u) / 64u;                                // partitions per lane in the offset scan (<= 4)
    // inputs of the current round and, in flight, of the next one
    float ld0[SPT], ld1[SPT], lx0[SPT], lx1[SPT], lx2[SPT], nd0[SPT], nd1[SPT], nx0[SPT], nx1[SPT], nx2[SPT];
    auto fetch = [&](uint32_t rb0) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            uint32_t i = min(rb0 + s * S3_BIN_THREADS + threadIdx.x, n - 1);
            if (rows) i = rows[i];
            const float* xp = x + (size_t)i * x_stride;
#ifdef S3_PROBE_LOAD_ORDER      // (probe builds, tools/build_variant.sh: does the two-process deviation follow the FIRST load of the fetch?)
            nx0[s] = xp[0]; nx1[s] = xp[1]; nx2[s] = xp[2]; nd1[s] = d1p[i]; nd0[s] = d0p[i];
#else
            nd0[s] = d0p[i]; nd1[s] = d1p[i]; nx0[s] = xp[0]; nx1[s] = xp[1]; nx2[s] = xp[2];
#endif
        }
    };
    fetch(b0);
    uint32_t amax = 0;                                                      // max |dL/d feature| of this thread's rows, as bits
    for (uint32_t r = 0; r < (uint32_t)ROUNDS; ++r) {
        const uint32_t rb0 = b0 + r * ROUND;
        if (rb0 >= n) break;                                                // uniform
#pragma unroll
        for (int s = 0; s < SPT; ++s) { ld0[s] = nd0[s]; ld1[s] = nd1[s]; lx0[s] = nx0[s]; lx1[s] = nx1[s]; lx2[s] = nx2[s]; }
#pragma unroll
        for (int s = 0; s < SPT; ++s)
            if (rb0 + s * S3_BIN_THREADS + threadIdx.x < n) amax = max(amax, max(s3_abs_bits(ld0[s]), s3_abs_bits(ld1[s])));
        for (uint32_t p = threadIdx.x; p < parts; p += S3_BIN_THREADS) s_cnt[p] = 0;
        __syncthreads();
        // The thread's items are GENERATED twice -- once to rank them per partition, once to place them -- instead of being
        // kept in registers in between: 16 item slots (kind D, every pair split) x 6 values cost 181 VGPRs, i.e. one workgroup
        // per CU instead of two.  f(k, partition, packed local entries, va, vb, w): slot k is a compile-time constant.
        auto items = [&](auto&& f) __attribute__((always_inline)) {
            s3_static_for<SPT>([&](auto S_) __attribute__((always_inline)) {
                constexpr int s = decltype(S_)::value;
                const uint32_t i = rb0 + s * S3_BIN_THREADS + threadIdx.x;
                const float d0 = ld0[s], d1 = ld1[s];
                if (!(i < n) || (d0 == 0.f && d1 == 0.f)) return;           // rows with a zero gradient add nothing
                const float p0 = lx0[s] * scale + 0.5f, p1 = lx1[s] * scale + 0.5f, p2 = lx2[s] * scale + 0.5f;
                const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
                const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
                const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
                s3_static_for<4>([&](auto C_) __attribute__((always_inline)) {
                    constexpr int c = decltype(C_)::value;
                    constexpr uint32_t cy = c & 1u, cz = c >> 1;
                    const float wyz = (cy ? w1 : 1.f - w1) * (cz ? w2 : 1.f - w2);
                    const float va = wyz * d0, vb = wyz * d1;
                    if constexpr (KIND == S3_H) {
                        const uint32_t h = ((g1 + cy) * 2654435761u) ^ ((g2 + cz) * 805459861u);
                        const uint32_t i0 = (g0 ^ h) & hmask, i1 = ((g0 + 1u) ^ h) & hmask;
                        // partition = i0 >> lg == i1 >> lg: x < 2^lg never reaches these bits
                        f(std::integral_constant<int, s * IPS + c>{}, i0 >> lg, (i0 & emask) | ((i1 & emask) << S3_LOG2), va, vb, w0);
                    } else {
                        constexpr int k = s * IPS + 2 * c;
                        const uint32_t rr = (g1 + cy) + (g2 + cz) * res;    // grid row of the pair
                        // slot k: the pair (or its first half), slot k + 1: the second half of a split pair -- each slot is
                        // filled at ONE place in the code (a slot index merged over two branches would become a dynamic
                        // register-array index, i.e. scratch memory)
                        const bool whole = g0 + 1u < res && rr < res * res;   // both x-neighbours in row rr of the res^3 lattice
                        uint32_t pa = rr & pmask, ea = (rr >> plog2) * res + g0, eb = ea + 1u, pb = 0u;
                        if (!whole) {                                       // upper faces of the domain / positions outside it
                            const uint32_t base = g0 + rr * res;            // == g0 + y res + z res^2 (mod 2^32), tcnn's linear index
                            s3_dense_local(s3_wrap(base, hsize), res, plog2, pmask, &pa, &ea);
                            s3_dense_local(s3_wrap(base + 1u, hsize), res, plog2, pmask, &pb, &eb);
                        }
                        f(std::integral_constant<int, k>{}, pa, ea | ((whole ? eb : ea) << S3_LOG2), whole ? va : (1.f - w0) * va,
                          whole ? vb : (1.f - w0) * vb, whole ? w0 : 0.f);
                        if (!whole) f(std::integral_constant<int, k + 1>{}, pb, eb | (eb << S3_LOG2), w0 * va, w0 * vb, 0.f);
                    }
                });
            });
        };
        uint32_t irank[NI];
        items([&](auto K_, uint32_t part, uint32_t, float, float, float) __attribute__((always_inline)) { irank[decltype(K_)::value] = atomicAdd(&s_cnt[part], 1u); });
#ifndef S3_PROBE_NO_PREFETCH    // (probe build: no load in flight across the round's barriers -- fetched at the round's end instead)
        if (r + 1 < (uint32_t)ROUNDS && rb0 + ROUND < n) fetch(rb0 + ROUND);   // next round's inputs under this round's ranking
#endif
        __syncthreads();
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
        items([&](auto K_, uint32_t part, uint32_t pr, float va, float vb, float w) __attribute__((always_inline)) {
            const uint32_t q = s_off[part] + irank[decltype(K_)::value];
            if (KIND == S3_H || q < S3_ROUND_ITEMS) {
                s_items[q] = make_float4(__uint_as_float(pr), va, vb, w);
                s_ipart[q] = (uint8_t)part;
            } else {          // kind D, more than 4 items per sample on average (samples piled on the domain's upper faces): overflow records
                const uint32_t e0 = pr & (S3_ENTRIES - 1), e1 = pr >> S3_LOG2;
                const uint32_t o = atomicAdd(s_ovf, e0 == e1 ? 1u : 2u);
                ovo[o] = make_float4(__uint_as_float(e0 | (part << S3_LOG2)), (1.f - w) * va, (1.f - w) * vb, 0.f);
                if (e0 != e1) ovo[o + 1] = make_float4(__uint_as_float(e1 | (part << S3_LOG2)), w * va, w * vb, 0.f);
            }
        });
#ifdef S3_PROBE_REREAD           // (probe build: is the value this round worked with what the row holds?  read it again, uncached, and say so if not)
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            uint32_t i = min(rb0 + s * S3_BIN_THREADS + threadIdx.x, n - 1);
            if (rows) i = rows[i];
            const float a0 = __builtin_nontemporal_load(d0p + i), a1 = __builtin_nontemporal_load(d1p + i);
            if (__float_as_uint(a0) != __float_as_uint(ld0[s]) || __float_as_uint(a1) != __float_as_uint(ld1[s]))
                printf("REREAD level-slot %u drow %u block %u round %u thread %u s %d sample %u: d0 held %08x now %08x | d1 held %08x now %08x | neighbours now %08x %08x\n",
                       (unsigned)(&L - (const S3Level*)nullptr) & 0u, L.drow, sb, r, threadIdx.x, s, i, __float_as_uint(ld0[s]), __float_as_uint(a0),
                       __float_as_uint(ld1[s]), __float_as_uint(a1), __float_as_uint(d0p[i > 0 ? i - 1 : 0]), __float_as_uint(d0p[min(i + 1, n - 1)]));
        }
#endif
        __syncthreads();
        const uint32_t total = KIND == S3_H ? s_off[parts] : min(s_off[parts], (uint32_t)S3_ROUND_ITEMS);
        for (uint32_t q = threadIdx.x; q < total; q += S3_BIN_THREADS) {
            const uint32_t p = s_ipart[q], rk = q - s_off[p] + s_base[p];
            const float4 it = s_items[q];
            if (rk < cap) {
                out[(size_t)p * nsb * cap + rk] = it;
            } else {                                                        // overfull sub-bin: two single-entry records
                const uint32_t pr = __float_as_uint(it.x), e0 = pr & (S3_ENTRIES - 1), e1 = pr >> S3_LOG2;
                if (e0 == e1) {                                             // a single-entry item (weight 0): one record
                    const uint32_t o = atomicAdd(s_ovf, 1u);
                    ovo[o] = make_float4(__uint_as_float(e0 | (p << S3_LOG2)), (1.f - it.w) * it.y, (1.f - it.w) * it.z, 0.f);
                } else {
                    const uint32_t o = atomicAdd(s_ovf, 2u);
                    ovo[o] = make_float4(__uint_as_float(e0 | (p << S3_LOG2)), (1.f - it.w) * it.y, (1.f - it.w) * it.z, 0.f);
                    ovo[o + 1] = make_float4(__uint_as_float(e1 | (p << S3_LOG2)), it.w * it.y, it.w * it.z, 0.f);
                }
            }
        }
        __syncthreads();
        // (kind D: items of a partition that went straight to the overflow list do not take sub-bin slots)
        for (uint32_t p = threadIdx.x; p < parts; p += S3_BIN_THREADS)
            s_base[p] += KIND == S3_H ? s_cnt[p] : min(s_off[p] + s_cnt[p], (uint32_t)S3_ROUND_ITEMS) - min(s_off[p], (uint32_t)S3_ROUND_ITEMS);
#ifdef S3_PROBE_NO_PREFETCH
        if (r + 1 < (uint32_t)ROUNDS && rb0 + ROUND < n) fetch(rb0 + ROUND);
#endif
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < parts; p += S3_BIN_THREADS) cnt_out[(size_t)p * nsb] = min(s_base[p], cap);
    if (threadIdx.x == 0) *ovfcnt_out = *s_ovf;
    // the block's maximum: waves through s_cnt (free again behind the barrier above)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, d, 64));
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (uint32_t w = 0; w < S3_BIN_THREADS / 64; ++w) m = max(m, s_cnt[w]);
        *amax_out = m;
    }
}

// max |dL/d feature| of the rows [b0, b0 + bs) of one feature-row pair (the run-length levels' maxima: their workgroups run in the
// accumulate launch, behind this one)
__device__ __forceinline__ void s3_rows_absmax(const float* __restrict__ d0p, uint32_t ld, uint32_t b0, uint32_t bs, uint32_t n,
                                               const uint32_t* __restrict__ rows, uint32_t* s_red, uint32_t* __restrict__ out) {
    uint32_t amax = 0;
    for (uint32_t q = b0 + threadIdx.x; q < min(b0 + bs, n); q += S3_BIN_THREADS) {
        const uint32_t i = rows ? rows[q] : q;
        amax = max(amax, max(s3_abs_bits(d0p[i]), s3_abs_bits(d0p[(size_t)ld + i])));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, d, 64));
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (uint32_t w = 0; w < S3_BIN_THREADS / 64; ++w) m = max(m, s_red[w]);
        *out = m;
    }
    __syncthreads();                                                        // (s_red is the caller's to reuse)
}

template <int BS>
__global__ __launch_bounds__(S3_BIN_THREADS) void k_scatter_bin3(S3Plan pl, const float* __restrict__ x, uint32_t x_stride,
                                                                 const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                                                 const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                                 uint32_t* __restrict__ counts, float4* __restrict__ bins,
                                                                 float4* __restrict__ ovf, uint32_t n_aux_blocks, XrAuxWork aux) {
    __shared__ float4 s_items[S3_ROUND_ITEMS];
    __shared__ uint8_t s_ipart[S3_ROUND_ITEMS];
    __shared__ uint32_t s_cnt[S3_MAX_PARTS], s_off[S3_MAX_PARTS + 1], s_base[S3_MAX_PARTS], s_ovf;
    // the workgroups in front of the binning ones (dispatched first, they run beside them): the sum of a training step's MLP gradient partials + those tensors' update (xr_aux.h) --
    // they depend on what ran before this launch only, and they have to be finished when the scatter is
    if (blockIdx.x < n_aux_blocks) { xr_aux_reduce_block<S3_BIN_THREADS>(aux, blockIdx.x, reinterpret_cast<float (*)[64]>(s_items)); return; }
    const uint32_t bidx = blockIdx.x - n_aux_blocks;
    const uint32_t e = pl.n_lv - 1u - bidx % pl.n_lv, sb = bidx / pl.n_lv;               // the levels of one sample block are neighbours
    const S3Level& L = pl.lv[e];
    if (n_dev) n = min(n, *n_dev);
    uint32_t* __restrict__ cnt_out = counts + L.counts_off + sb;            // [part][sample block]
    uint32_t* __restrict__ ovfcnt_out = counts + pl.ovfcnt_off + e * pl.nsb + sb;
    uint32_t* __restrict__ amax_out = counts + pl.amax_off + e * pl.nsb + sb;
    if (sb * BS >= n) {                                                     // uniform: an empty sample block has empty sub-bins
        for (uint32_t p = threadIdx.x; p < L.parts; p += S3_BIN_THREADS) cnt_out[(size_t)p * pl.nsb] = 0;
        if (threadIdx.x == 0) {
            *ovfcnt_out = 0; *amax_out = 0;
            if (pl.n_rl && pl.n_lv >= pl.n_rl) counts[pl.ramax_off + ((e % pl.n_rl) * pl.nsb + sb) * pl.rl_slices + e / pl.n_rl] = 0;
            else for (uint32_t r = e; r < pl.n_rl; r += pl.n_lv) counts[pl.ramax_off + (r * pl.nsb + sb) * pl.rl_slices] = 0;
        }
        return;
    }
    // the run-length levels' maxima: the workgroup of binned level e looks at run-length level e mod n_rl over slice e / n_rl of its
    // sample block (a quarter of it with 13 binned and 3 run-length levels: every workgroup scans a little instead of three scanning a lot)
    // (fewer binned than run-length levels -- odd geometries: level e, e + n_lv, ... over the whole block, one slot each)
    if (pl.n_rl && pl.n_lv < pl.n_rl) {
        for (uint32_t r = e; r < pl.n_rl; r += pl.n_lv)
            s3_rows_absmax(denc_t + (size_t)pl.rl_drow[r] * ld, ld, sb * BS, BS, n, rows, s_cnt, counts + pl.ramax_off + (r * pl.nsb + sb) * pl.rl_slices);
    } else if (pl.n_rl) {
        const uint32_t r = e % pl.n_rl, k = e / pl.n_rl, kr = (pl.n_lv - r + pl.n_rl - 1) / pl.n_rl;     // kr slices for level r
        const uint32_t per = (BS + kr - 1) / kr;
        s3_rows_absmax(denc_t + (size_t)pl.rl_drow[r] * ld, ld, sb * BS + k * per, min(per, BS - min(BS, k * per)), n, rows, s_cnt,
                       counts + pl.ramax_off + (r * pl.nsb + sb) * pl.rl_slices + k);
    }
    if (L.kind == S3_H)
        s3_bin_block<S3_H, BS>(L, pl.lg, pl.nsb, pl.ovf_cap, sb, x, x_stride, denc_t, ld, n, rows, cnt_out, ovfcnt_out, amax_out, bins, ovf, s_items, s_ipart,
                               s_cnt, s_off, s_base, &s_ovf);
    else
        s3_bin_block<S3_D, BS>(L, pl.lg, pl.nsb, pl.ovf_cap, sb, x, x_stride, denc_t, ld, n, rows, cnt_out, ovfcnt_out, amax_out, bins, ovf, s_items, s_ipart,
                               s_cnt, s_off, s_base, &s_ovf);
}

// (geometries with run-length levels but no binned level: nobody else records their maxima)
__global__ __launch_bounds__(S3_BIN_THREADS) void k_scatter_rl_absmax(S3Plan pl, uint32_t bs, const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                                                      const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                                      uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_red[S3_BIN_THREADS / 64];
    if (n_dev) n = min(n, *n_dev);
    const uint32_t r = blockIdx.x % pl.n_rl, sb = blockIdx.x / pl.n_rl;
    s3_rows_absmax(denc_t + (size_t)pl.rl_drow[r] * ld, ld, sb * bs, bs, n, rows, s_red, counts + pl.ramax_off + (r * pl.nsb + sb) * pl.rl_slices);
}

// ------------------------------------------------------------------------------------------------ accumulate
// One workgroup = (level, partition): 8-byte LDS accumulators -- 64-bit fixed point (returnless ds_add_u64: 4.7 ns per wave instruction;
// ds_add_f64, the accumulators of rounds 3-5 and of -DS3_FIX=0: 8.6 ns; ds_add_f32: 81 ns -- tools/lds_probe.hip), rounded to fp32
// once when the partition is written to the table.
// ---- Adam applied where the gradient of an entry is complete (XrAdamArgs, xr_adam.h): the scatter owns every table entry of
// its levels exactly once per launch (what XR_SCATTER_OVERWRITE relies on), so instead of writing 48.8 MB of gradient that the
// optimiser launch reads back next, the accumulate / fold kernels run the optimiser's update on (p, m, v, ema) themselves:
// 98 MB and one 70-us HBM-bound launch less per step, its streams spread under the LDS-bound phases of other workgroups.
// Same adam1 / ema1 as k_adam_multi, same operand order: parameters bit for bit those of scatter + xr_adam_step_multi.
typedef float s3_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 s3_ldnt(const float* base, size_t e) {
    const s3_f2 r = __builtin_nontemporal_load(reinterpret_cast<const s3_f2*>(base) + e);
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ void s3_stnt(float* base, size_t e, float2 v) {
    s3_f2 r; r.x = v.x; r.y = v.y;
    __builtin_nontemporal_store(r, reinterpret_cast<s3_f2*>(base) + e);
}
typedef float s3_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 s3_ld4nt(const float* base, size_t i4) {
    const s3_f4 r = __builtin_nontemporal_load(reinterpret_cast<const s3_f4*>(base) + i4);
    return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void s3_st4nt(float* base, size_t i4, float4 v) {
    s3_f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    __builtin_nontemporal_store(r, reinterpret_cast<s3_f4*>(base) + i4);
}
template <int N>
__device__ __forceinline__ void s3_adam_entries(const XrAdamArgs& A, const size_t (&e)[N], const float2 (&g)[N], const bool (&on)[N]) {
    float2 p[N], m[N], v[N], q[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (on[k]) {
            p[k] = reinterpret_cast<const float2*>(A.p)[e[k]]; m[k] = s3_ldnt(A.m, e[k]); v[k] = s3_ldnt(A.v, e[k]);
            if (A.ema) q[k] = s3_ldnt(A.ema, e[k]);
        }
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (on[k]) {
            adam1(p[k].x, g[k].x, m[k].x, v[k].x, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
            adam1(p[k].y, g[k].y, m[k].y, v[k].y, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
            reinterpret_cast<float2*>(A.p)[e[k]] = p[k]; s3_stnt(A.m, e[k], m[k]); s3_stnt(A.v, e[k], v[k]);
            if (A.ema) { q[k].x = ema1(q[k].x, p[k].x, A.mom); q[k].y = ema1(q[k].y, p[k].y, A.mom); s3_stnt(A.ema, e[k], q[k]); }
        }
}

// TH: threads per workgroup.  1024 (16 waves) or 512 (8 waves: the same time, profiles/r03_accumulate_512_threads_ab.txt, with half
// the register file left to whatever else is resident -- the side-stream march, for one).
template <int LG, int TH>
__device__ __forceinline__ void s3_accum_block(const S3Plan& pl, const uint32_t blk, const uint32_t* __restrict__ counts,
                                               const float4* __restrict__ bins, const float4* __restrict__ ovf,
                                               float* __restrict__ grad_table, double* s_acc /* [ENTRIES][2] */) {
    constexpr uint32_t ENTRIES = 1u << LG, THREADS = (uint32_t)TH, WAVES = THREADS / 64;
    uint32_t e = 0;
    while (e + 1 < pl.n_lv && blk >= pl.lv[e + 1].acc_block0) ++e;          // levels in accumulate order (heaviest first)
    const S3Level& L = pl.lv[e];
    const uint32_t part = blk - L.acc_block0, nsb = pl.nsb, cap = L.cap;
    double2* acc2 = reinterpret_cast<double2*>(s_acc);
    S3_T(0);
    // entries this partition owns
    uint32_t n_loc;
    if (L.kind == S3_H) n_loc = ENTRIES;
    else n_loc = s3_dense_rows(L.res, L.plog2, part) * L.res + (part == 0u ? L.hsize - L.res * L.res * L.res : 0u);
    // this wave's sub-bins w, w + 16, ...: their fill counts in one VGPR, the overflow counts of the level's sample blocks too
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t* __restrict__ cnt = counts + L.counts_off + (size_t)part * nsb;
    const uint32_t my_sb = wave + WAVES * lane;
    const uint32_t vfill = my_sb < nsb ? cnt[my_sb] : 0u;
    const uint32_t vovf = my_sb < nsb ? counts[pl.ovfcnt_off + e * nsb + my_sb] : 0u;
    const S3Scale sc = s3_scale(s3_level_max(counts + pl.amax_off + e * nsb, nsb), pl.n_bound);
    const float4* __restrict__ src = bins + L.bins_off + (size_t)part * nsb * cap;
    constexpr uint32_t U = 8;
    const uint32_t n_mine = nsb > wave ? (nsb - wave + WAVES - 1u) / WAVES : 0u;   // sub-bins of this wave
    uint32_t si = 0, c = 0;                                                 // next (sub-bin slot, 64-item chunk): wave-uniform
    uint32_t fill = n_mine ? s3_readlane(vfill, 0) : 0u;
    auto skip_empty = [&]() {
        while (si < n_mine && fill == 0u) { ++si; fill = si < n_mine ? s3_readlane(vfill, si) : 0u; }
    };
    skip_empty();
    S3_T(1);
    float4 nx[U];
    bool non[U];
    auto fetch = [&]() {
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            non[u] = false;
            if (si < n_mine) {
                // lane l of chunk c takes item l * nch + c of the sub-bin (nch = its number of chunks), not c * 64 + l: consecutive
                // items come from consecutive samples of a ray, which at the coarse levels sit in the same cell -- side by side
                // in one wave instruction their LDS atomics hit the same addresses and serialise (measured with
                // tools/scatter3_timing.hip: 18 us of atomics per workgroup at levels 5-6 against 7 us at level 15)
                const uint32_t nch = (fill + 63u) >> 6, off = S3_PERMUTE ? lane * nch + c : c * 64u + lane;
                non[u] = off < fill;
                if (non[u]) nx[u] = src[(size_t)(wave + WAVES * si) * cap + off];
                ++c;
                if (c * 64u >= fill) {
                    c = 0; ++si;
                    fill = si < n_mine ? s3_readlane(vfill, si) : 0u;
                    skip_empty();
                }
            }
        }
    };
    fetch();
    S3_T(2);
    // the accumulators are cleared while the first items are in flight
    for (uint32_t q = threadIdx.x; q < n_loc; q += THREADS) acc2[q] = make_double2(0.0, 0.0);
    __syncthreads();
    S3_T(3);
    uint32_t run_pr = 0xffffffffu;                                          // entry pair of the lane's current run (no item packs to all ones)
    s3_acc_t r00 = 0, r01 = 0, r10 = 0, r11 = 0;
    for (;;) {
        float4 it[U];
        bool on[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { it[u] = nx[u]; on[u] = non[u]; }
        const bool more = si < n_mine;                                      // uniform per wave
        if (more) fetch();
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const uint32_t pr = __float_as_uint(it[u].x);
            const float w0 = it[u].w, a = it[u].y, b = it[u].z;
            if (S3_MERGE) {
                // a lane's successive items are successive items of its sub-bins (the permuted mapping): while they name the same
                // entry pair -- samples of one ray inside one cell -- their four products are summed in registers (as fixed-point integers, like
                // the LDS accumulators) and go to the LDS once per run
                if (pr != run_pr) {
                    if (run_pr != 0xffffffffu) {
                        const uint32_t i0 = run_pr & (S3_ENTRIES - 1), i1 = run_pr >> S3_LOG2;
                        s3_lds_add(s_acc, 2 * i0, r00); s3_lds_add(s_acc, 2 * i0 + 1, r01);
                        s3_lds_add(s_acc, 2 * i1, r10); s3_lds_add(s_acc, 2 * i1 + 1, r11);
                    }
                    run_pr = pr; r00 = r01 = r10 = r11 = 0;
                }
                r00 += s3_q((1.f - w0) * a, sc); r01 += s3_q((1.f - w0) * b, sc);
                r10 += s3_q(w0 * a, sc); r11 += s3_q(w0 * b, sc);
            } else {
                const uint32_t i0 = pr & (S3_ENTRIES - 1), i1 = pr >> S3_LOG2;
                s3_lds_add(s_acc, 2 * i0, s3_q((1.f - w0) * a, sc));
                s3_lds_add(s_acc, 2 * i0 + 1, s3_q((1.f - w0) * b, sc));
                s3_lds_add(s_acc, 2 * i1, s3_q(w0 * a, sc));
                s3_lds_add(s_acc, 2 * i1 + 1, s3_q(w0 * b, sc));
            }
        }
        if (!more) break;
    }
    if (S3_MERGE && run_pr != 0xffffffffu) {
        const uint32_t i0 = run_pr & (S3_ENTRIES - 1), i1 = run_pr >> S3_LOG2;
        s3_lds_add(s_acc, 2 * i0, r00); s3_lds_add(s_acc, 2 * i0 + 1, r01);
        s3_lds_add(s_acc, 2 * i1, r10); s3_lds_add(s_acc, 2 * i1 + 1, r11);
    }
    S3_T(4);
    // overflow records of the level's sample blocks (none unless the samples cluster): every partition scans them all
    if (__ballot(vovf != 0u) != 0ull) {
        for (uint32_t k = 0; k < n_mine; ++k) {
            const uint32_t cntk = s3_readlane(vovf, k);
            const float4* __restrict__ ov = ovf + L.ovf_off + (size_t)(wave + WAVES * k) * pl.ovf_cap;
            for (uint32_t q = lane; q < cntk; q += 64u) {
                const float4 rcd = ov[q];
                const uint32_t key = __float_as_uint(rcd.x);
                if ((key >> S3_LOG2) != part) continue;
                s3_lds_add(s_acc, 2 * (key & (S3_ENTRIES - 1)), s3_q(rcd.y, sc));
                s3_lds_add(s_acc, 2 * (key & (S3_ENTRIES - 1)) + 1, s3_q(rcd.z, sc));
            }
        }
    }
    // fused optimiser update, hashed levels: this thread's 8 entries are 4 pairs of neighbours (16 B of p / m / v / ema each), taken
    // in two rounds of two pairs; the first round's loads go out before the barrier that waits for every wave's LDS atomics
    // (round 6: with 8 waves the whole partition's optimiser state -- 8 pairs x (p, m, v, ema) = 128 VGPRs -- is requested before the
    // barrier, one round instead of four dependent ones: 254 VGPRs, no spill, -2..3 us per launch in the loop; with 16 waves that
    // would spill, so they keep the rounds of two pairs.  profiles/r06_scatter_counters.txt)
    constexpr uint32_t FP = ENTRIES / (2 * THREADS), FH = THREADS <= 512 ? FP : 2, ROUNDS = FP / FH;
    static_assert(FP % FH == 0 && ROUNDS >= 1, "rounds of two pairs");
    const bool fuse_h = pl.fuse != 0u && L.kind == S3_H;
    float4 fp_[FH], fm_[FH], fv_[FH], fq_[FH];
    const size_t f4_0 = ((size_t)L.toff + (size_t)part * ENTRIES) / 2;      // float4 index of the partition's first pair (offsets are even)
    auto adam_load = [&](uint32_t k0) {
#pragma unroll
        for (uint32_t k = 0; k < FH; ++k) {
            const size_t i4 = f4_0 + (k0 + k) * THREADS + threadIdx.x;
            fp_[k] = reinterpret_cast<const float4*>(pl.ad.p)[i4]; fm_[k] = s3_ld4nt(pl.ad.m, i4); fv_[k] = s3_ld4nt(pl.ad.v, i4);
            if (pl.ad.ema) fq_[k] = s3_ld4nt(pl.ad.ema, i4);
        }
    };
    if (fuse_h) adam_load(0);
    __syncthreads();
    S3_T(5);
    float2* __restrict__ tab = reinterpret_cast<float2*>(grad_table) + L.toff;
    const bool add = pl.overwrite == 0u;
    if (fuse_h) {
        const XrAdamArgs& A = pl.ad;
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            double2 a0[FH], a1[FH];
#pragma unroll
            for (uint32_t k = 0; k < FH; ++k) { const uint32_t q = 2u * ((r * FH + k) * THREADS + threadIdx.x); a0[k] = acc2[q]; a1[k] = acc2[q + 1]; }
#pragma unroll
            for (uint32_t k = 0; k < FH; ++k) {
                adam1(fp_[k].x, s3_val(a0[k].x, sc), fm_[k].x, fv_[k].x, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
                adam1(fp_[k].y, s3_val(a0[k].y, sc), fm_[k].y, fv_[k].y, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
                adam1(fp_[k].z, s3_val(a1[k].x, sc), fm_[k].z, fv_[k].z, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
                adam1(fp_[k].w, s3_val(a1[k].y, sc), fm_[k].w, fv_[k].w, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
                const size_t i4 = f4_0 + (r * FH + k) * THREADS + threadIdx.x;
                reinterpret_cast<float4*>(A.p)[i4] = fp_[k]; s3_st4nt(A.m, i4, fm_[k]); s3_st4nt(A.v, i4, fv_[k]);
                if (A.ema) {
                    fq_[k].x = ema1(fq_[k].x, fp_[k].x, A.mom); fq_[k].y = ema1(fq_[k].y, fp_[k].y, A.mom);
                    fq_[k].z = ema1(fq_[k].z, fp_[k].z, A.mom); fq_[k].w = ema1(fq_[k].w, fp_[k].w, A.mom);
                    s3_st4nt(A.ema, i4, fq_[k]);
                }
            }
            if (r + 1 < ROUNDS) adam_load((r + 1) * FH);
        }
    } else if (pl.fuse) {
        // (dense levels: strided rows) the partition's entries, 4 per thread and round: (p, m, v, ema) of all four in flight together
        constexpr int AB = 4;
        const uint32_t res = L.res, lat = L.kind == S3_H ? 0u : s3_dense_rows(res, L.plog2, part) * res;
        for (uint32_t q0 = 0; q0 < n_loc; q0 += AB * THREADS) {
            size_t e[AB]; float2 g[AB]; bool on[AB];
#pragma unroll
            for (int k = 0; k < AB; ++k) {
                const uint32_t q = q0 + k * THREADS + threadIdx.x;
                on[k] = q < n_loc;
                uint32_t idx = 0;
                if (on[k]) {
                    if (L.kind == S3_H) idx = part * ENTRIES + q;
                    else if (q < lat) { const uint32_t rl = q / res; idx = ((rl << L.plog2) | part) * res + (q - rl * res); }
                    else idx = res * res * res + (q - lat);
                    const double2 a = acc2[q];
                    g[k] = make_float2(s3_val(a.x, sc), s3_val(a.y, sc));
                }
                e[k] = (size_t)L.toff + idx;
            }
            s3_adam_entries<AB>(pl.ad, e, g, on);
        }
    } else if (L.kind == S3_H) {
        float2* __restrict__ dst = tab + (size_t)part * ENTRIES;
        constexpr uint32_t F = ENTRIES / THREADS;
        float2 t[F];
#pragma unroll
        for (uint32_t k = 0; k < F; ++k) t[k] = add ? dst[k * THREADS + threadIdx.x] : make_float2(0.f, 0.f);
#pragma unroll
        for (uint32_t k = 0; k < F; ++k) {
            const double2 a = acc2[k * THREADS + threadIdx.x];
            t[k].x += s3_val(a.x, sc); t[k].y += s3_val(a.y, sc);
            dst[k * THREADS + threadIdx.x] = t[k];
        }
    } else {
        const uint32_t res = L.res, lat = s3_dense_rows(res, L.plog2, part) * res;       // lattice entries of this partition
        for (uint32_t q = threadIdx.x; q < n_loc; q += THREADS) {
            uint32_t idx;
            if (q < lat) { const uint32_t rl = q / res; idx = ((rl << L.plog2) | part) * res + (q - rl * res); }
            else idx = res * res * res + (q - lat);                          // padding entries behind the lattice (partition 0)
            const double2 a = acc2[q];
            float2 t = add ? tab[idx] : make_float2(0.f, 0.f);
            t.x += s3_val(a.x, sc); t.y += s3_val(a.y, sc);
            tab[idx] = t;
        }
    }
    S3_T(6);
}

// (Round 4 built this kernel as a PERSISTENT grid too -- one workgroup per CU walking its partitions, next partition's counts and
// first items fetched under the update phase, accumulators cleared by the pass that reads them: 102 us against 106 alone, 191-200
// against 172 in the training loop, where the march of iteration i + 2 shares ~50 CUs with it and a static or ticketed assignment
// ends with its slowest workgroup.  profiles/r04_persistent_accumulate_kernel_ab.txt; the kernel is in the history, commit "Scatter:
// persistent accumulate kernel".)

// ------------------------------------------------------------------------------------------------ small dense levels
// kind R.  Workgroup = (level, 2^13-entry partition, chunk of the rows): thread t walks rows [16 t, 16 t + 16) of its chunk
// keeping the 8 corners x 2 features of the CURRENT cell in registers; on a cell change the 16 sums go to the workgroup's
// 8-byte-per-sum LDS copy of the partition (only the corners that fall into it).  The chunk's partition is then written as fp32 into
// slab `chunk` of the workspace; k_scatter_fold adds the slabs in fixed order.
// Measured and dropped (round 3): a variant that reads the rows with coalesced loads (lane = row) and transposes them through
// a per-wave LDS tile so that lane l can walk rows 16 l .. 16 l + 15 -- 46.6 us against 36.8: with any lane of a wave flushing at
// nearly every step, the kernel issues its 16 LDS atomic instructions per step either way, and those, not the loads, set its time.
__device__ __forceinline__ void s3_r_flush(double* s_acc, const float (&acc)[16], uint32_t g0, uint32_t g1, uint32_t g2, uint32_t res,
                                           uint32_t hsize, uint32_t part, const S3Scale& sc) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t idx = s3_wrap((g0 + (c & 1)) + (g1 + ((c >> 1) & 1)) * res + (g2 + (c >> 2)) * res * res, hsize);
        if ((idx >> S3_LOG2) != part) continue;
        const uint32_t q = idx & (S3_ENTRIES - 1);
        if (acc[2 * c] != 0.f) s3_lds_add(s_acc, 2 * q, s3_q(acc[2 * c], sc));
        if (acc[2 * c + 1] != 0.f) s3_lds_add(s_acc, 2 * q + 1, s3_q(acc[2 * c + 1], sc));
    }
}

template <int TH>
__device__ __forceinline__ void s3_rl_block(const S3RPlan& pl, const uint32_t blk, const float* __restrict__ x, uint32_t x_stride,
                                            const float* __restrict__ denc_t, uint32_t ld, uint32_t n, const uint32_t* __restrict__ rows,
                                            const uint32_t* __restrict__ counts, float2* __restrict__ slabs, double* s_acc /* [<= S3_ENTRIES][2] */) {
    constexpr uint32_t S3_R_THREADS = (uint32_t)TH;
    uint32_t e = 0;
    while (e + 1 < pl.n_lv && blk >= pl.lv[e + 1].block0) ++e;
    const S3RLevel& L = pl.lv[e];
    const uint32_t rel = blk - L.block0, part = rel / pl.chunks, chunk = rel % pl.chunks;
    // (a thread's run sums are fp32 sums of <= 16 rows' contributions in row order: partial sums of the terms the bound counts)
    uint32_t vm = 0;
    {
        const uint32_t kr = pl.n_bin_lv >= pl.n_lv ? (pl.n_bin_lv - e + pl.n_lv - 1) / pl.n_lv : 1u;     // slots of this level that are written
        const uint32_t* __restrict__ slots = counts + pl.ramax_off + (size_t)e * pl.nsb * pl.rl_slices;
        for (uint32_t q = threadIdx.x & 63u; q < pl.nsb * kr; q += 64u) vm = max(vm, slots[(q / kr) * pl.rl_slices + q % kr]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) vm = max(vm, (uint32_t)__shfl_xor((int)vm, d, 64));
    }
    const S3Scale sc = s3_scale(vm, pl.n_bound);
    const uint32_t p_lo = part << S3_LOG2, n_loc = min(S3_ENTRIES, L.hsize - p_lo);
    double2* acc2 = reinterpret_cast<double2*>(s_acc);
    for (uint32_t q = threadIdx.x; q < n_loc; q += S3_R_THREADS) acc2[q] = make_double2(0.0, 0.0);
    __syncthreads();
    // chunk = rows [lo, hi) in units of S3_R_ROWS-row segments, the same number of segments per chunk
    const uint32_t n_seg = (n + S3_R_ROWS - 1) / S3_R_ROWS, seg_per = (n_seg + pl.chunks - 1) / pl.chunks;
    const uint32_t seg_lo = min(chunk * seg_per, n_seg), seg_hi = min(seg_lo + seg_per, n_seg);
    const float scale = L.scale;
    const uint32_t res = L.res, hsize = L.hsize;
    const float* __restrict__ d0p = denc_t + (size_t)L.drow * ld;
    const float* __restrict__ d1p = d0p + ld;
    for (uint32_t seg = seg_lo + threadIdx.x; seg < seg_hi; seg += S3_R_THREADS) {
        const uint32_t r0 = seg * S3_R_ROWS;
        uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu, c2 = 0xffffffffu;     // current cell
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
        for (int h = 0; h < S3_R_ROWS / 8; ++h) {                           // 8 rows' loads in flight
            float vx[8][3], vd0[8], vd1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t i = min(r0 + h * 8 + u, n - 1);
                if (rows) i = rows[i];
                const float* xp = x + (size_t)i * x_stride;
                vx[u][0] = xp[0]; vx[u][1] = xp[1]; vx[u][2] = xp[2]; vd0[u] = d0p[i]; vd1[u] = d1p[i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r0 + h * 8 + u >= n) continue;
                const float d0 = vd0[u], d1 = vd1[u];
                if (d0 == 0.f && d1 == 0.f) continue;
                const float p0 = vx[u][0] * scale + 0.5f, p1 = vx[u][1] * scale + 0.5f, p2 = vx[u][2] * scale + 0.5f;
                const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
                const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
                const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
                if (!(g0 == c0 && g1 == c1 && g2 == c2)) {
                    if (c0 != 0xffffffffu) s3_r_flush(s_acc, acc, c0, c1, c2, res, hsize, part, sc);
                    c0 = g0; c1 = g1; c2 = g2;
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float wt = ((c & 1) ? w0 : 1.f - w0) * ((c & 2) ? w1 : 1.f - w1) * ((c & 4) ? w2 : 1.f - w2);
                    acc[2 * c] += wt * d0; acc[2 * c + 1] += wt * d1;
                }
            }
        }
        if (c0 != 0xffffffffu) s3_r_flush(s_acc, acc, c0, c1, c2, res, hsize, part, sc);
    }
    __syncthreads();
    float2* __restrict__ dst = slabs + (size_t)chunk * pl.slab_entries + L.poff + p_lo;
    for (uint32_t q = threadIdx.x; q < n_loc; q += S3_R_THREADS) {
        const double2 a = acc2[q];
        dst[q] = make_float2(s3_val(a.x, sc), s3_val(a.y, sc));
    }
}

// The accumulate launch: the run-length workgroups of the small dense levels first (the longest ones: each walks 1 / chunks of the
// rows), then one workgroup per (binned level, partition).  Rounds 3-5 ran the two as separate kernels on two streams; as ONE launch
// they share the chip the same way without a fork and a join on the step's queue.
template <int LG, int TH>
__global__ __launch_bounds__(TH) void k_scatter_acc(S3Plan pl, S3RPlan rpl, const float* __restrict__ x, uint32_t x_stride,
                                                    const float* __restrict__ denc_t, uint32_t ld, uint32_t n,
                                                    const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ rows,
                                                    const uint32_t* __restrict__ counts, const float4* __restrict__ bins,
                                                    const float4* __restrict__ ovf, float2* __restrict__ slabs, float* __restrict__ grad_table,
                                                    uint32_t n_tail, XrAuxWork aux) {
    extern __shared__ __attribute__((aligned(16))) double s_acc[];       // [1 << LG][2]
    // (n_tail = 1: workgroup 0 is a training step's loss scalars + a clear, xr_aux.h -- a chain of dependent loads as long as the
    // binning launch, hidden in this one)
    if (blockIdx.x < n_tail) { xr_aux_tail_block<TH>(aux, reinterpret_cast<float*>(s_acc)); return; }
    const uint32_t blk = blockIdx.x - n_tail;
    if (blk < rpl.blocks) {
        if (n_dev) n = min(n, *n_dev);
        s3_rl_block<TH>(rpl, blk, x, x_stride, denc_t, ld, n, rows, counts, slabs, s_acc);
    } else
        s3_accum_block<LG, TH>(pl, blk - rpl.blocks, counts, bins, ovf, grad_table, s_acc);
}

// slabs of the run-length workgroups -> the table slice (or the optimiser update), in chunk order; 8 slab loads in flight
__global__ __launch_bounds__(256) void k_scatter_fold(S3RPlan pl, const float2* __restrict__ slabs, float* __restrict__ grad_table) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= pl.slab_entries) return;
    uint32_t e = 0;
    while (e + 1 < pl.n_lv && q >= pl.lv[e + 1].poff) ++e;
    float2* __restrict__ dst = reinterpret_cast<float2*>(grad_table) + pl.lv[e].toff + (q - pl.lv[e].poff);
    float2 t = (pl.overwrite || pl.fuse) ? make_float2(0.f, 0.f) : *dst;
    uint32_t c = 0;
    for (; c + 8 <= pl.chunks; c += 8) {                                    // fixed order
        float2 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = slabs[(size_t)(c + u) * pl.slab_entries + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) { t.x += a[u].x; t.y += a[u].y; }
    }
    for (; c < pl.chunks; ++c) {
        const float2 a = slabs[(size_t)c * pl.slab_entries + q];
        t.x += a.x; t.y += a.y;
    }
    if (pl.fuse) {
        const size_t ee[1] = {(size_t)pl.lv[e].toff + (q - pl.lv[e].poff)};
        const float2 gg[1] = {t};
        const bool on[1] = {true};
        s3_adam_entries<1>(pl.ad, ee, gg, on);
    } else *dst = t;
}

// ------------------------------------------------------------------------------------------------ host side
// XR_SC_TEST="min_n=256,block=1024,rl_chunks=3,rl=0" (read once per process): the layout parameters the tests vary -- the row
// threshold below which every level takes the atomic kernel (tests run the binned path at sizes the host emulation finishes
// quickly), samples per binning workgroup (1024 | 2048 | 4096), row chunks per partition of the run-length kernel, rl=0 the small
// dense levels through the bins.  Measurement-only alternatives of earlier rounds are compile-time now (S3_LOG2, S3_ACC_THREADS,
// S3_RL_ASYNC: tools/build_variant.sh) or gone with their records under profiles/.
struct S3Test { int min_n = 16384, block = 4096, rl_chunks = 8, rl = 1; };
static const S3Test& s3_test() {
    static const S3Test t = []() {
        S3Test v;
        const char* e = getenv("XR_SC_TEST");
        while (e && *e) {
            int val = 0;
            if (sscanf(e, "min_n=%d", &val) == 1) v.min_n = val;
            else if (sscanf(e, "block=%d", &val) == 1) v.block = (val == 1024 || val == 2048) ? val : 4096;
            else if (sscanf(e, "rl_chunks=%d", &val) == 1) v.rl_chunks = (val >= 1 && val <= 64) ? val : 8;
            else if (sscanf(e, "rl=%d", &val) == 1) v.rl = val != 0;
            const char* c = strchr(e, ',');
            e = c ? c + 1 : nullptr;
        }
        return v;
    }();
    return t;
}
static uint32_t s3_block_samples() { return (uint32_t)s3_test().block; }     // round 3 (beside the run-length kernel): 152 / 139 / 155 us for 4096 / 2048 / 1024; round 6 (one stream): 148-150 / 152-153 / 166
static uint32_t s3_chunks() { return (uint32_t)s3_test().rl_chunks; }     // (round 6, inside the accumulate launch with 8 waves: 147-155 us at 8, 158-160 at 16 / 6, 170 at 4)
#ifndef S3_ACC_THREADS_DEFAULT
#define S3_ACC_THREADS_DEFAULT 512      // 512 | 1024 threads per workgroup of the accumulate launch: the same time alone
#endif                                  // (profiles/r03_accumulate_512_threads_ab.txt); 8 waves can hold a partition's optimiser state in registers

struct S3Layout {
    S3Plan bin; S3RPlan rl;
    uint32_t atomic_mask;            // levels that take the atomic kernel (xr_encode.hip)
    size_t counts_bytes, bins_bytes, ovf_bytes, slabs_bytes;
};

static bool s3_layout(uint32_t n, const GridMeta& gm, uint32_t hashed_mask, int overwrite, S3Layout* out) {
    S3Layout& P = *out;
    memset(&P, 0, sizeof(P));
    const uint32_t bs = s3_block_samples();
    const uint32_t nsb = xr_div_up(n, bs);
    // below min_n rows (default 16384; XR_SC_TEST) the fixed costs -- 128 KiB of LDS per partition zeroed and written back -- exceed
    // what the atomic kernel needs; the tests lower it to run this path at sizes the host emulation finishes quickly
    const uint32_t min_n = (uint32_t)s3_test().min_n;
    if (n < min_n || nsb > S3_MAX_SB) { P.atomic_mask = (1u << gm.n_levels) - 1u; return false; }
    // (measured and dropped: 2^12-entry partitions accumulated by 512-thread workgroups in 64 KiB of LDS, two per CU:
    // 172.8 us against 138.7 for the whole entry point, profiles/r03_scatter3_defaults.txt)
    const uint32_t lg = S3_LOG2, entries = S3_ENTRIES;
    P.bin.nsb = nsb; P.bin.overwrite = (uint32_t)overwrite; P.bin.ovf_cap = 8u * bs; P.bin.lg = lg;
    P.rl.chunks = s3_chunks(); P.rl.overwrite = (uint32_t)overwrite;
    // rl=0 (XR_SC_TEST): the small dense levels through the binned path too (277 us instead of 41 before the accumulate kernel merged
    // runs in registers; kept as a test of the kind-D layout at small resolutions)
    const int use_rl = s3_test().rl;
    uint64_t bins_off = 0, ovf_off = 0;
    uint32_t counts_off = 0;
    // accumulate order: dense binned levels first (32 partitions carry twice a hashed partition's items)
    for (int pass = 0; pass < 2; ++pass)
        for (int l = gm.n_levels - 1; l >= 0; --l) {
            const uint32_t hsize = gm.off[l + 1] - gm.off[l], res = gm.res[l];
            const bool hashed = (hashed_mask >> l) & 1;
            const bool kindH = hashed && (hsize & (hsize - 1)) == 0 && hsize >= entries && (hsize >> lg) <= S3_MAX_PARTS &&
                               res < entries && (gm.off[l] & 1) == 0;
            const bool kindR = !hashed && hsize <= S3_R_MAX_ENTRIES && use_rl;
            const bool kindD = !hashed && !kindR && res >= 8u && res <= 128u && (uint64_t)res * res * res <= hsize;
            if (pass == 0 && !kindH && !kindD && !kindR) P.atomic_mask |= 1u << l;
            if (pass == 0 && kindR) {
                S3RLevel& R = P.rl.lv[P.rl.n_lv++];
                R.scale = gm.scale[l]; R.res = res; R.hsize = hsize; R.toff = gm.off[l]; R.parts = xr_div_up(hsize, S3_ENTRIES);
                R.drow = 2u * (uint32_t)l; R.poff = P.rl.slab_entries; R.block0 = P.rl.blocks;
                P.rl.slab_entries += hsize; P.rl.blocks += R.parts * P.rl.chunks;
            }
            if (!((pass == 0 && kindD) || (pass == 1 && kindH))) continue;
            S3Level& L = P.bin.lv[P.bin.n_lv++];
            L.scale = gm.scale[l]; L.res = res; L.hsize = hsize; L.toff = gm.off[l]; L.drow = 2u * (uint32_t)l;
            L.kind = kindH ? S3_H : S3_D;
            if (kindH) { L.parts = hsize >> lg; L.plog2 = 0; }
            else {
                // a partition holds ceil(res^2 / P) grid rows of res entries (+ the padding entries behind the lattice) <= 2^13
                L.parts = 32u; L.plog2 = 5u;
                while (((res * res + L.parts - 1u) / L.parts) * res + 8u > entries && L.parts < S3_MAX_PARTS) { L.parts *= 2u; ++L.plog2; }
            }
            const uint32_t ips = 4u;                                          // items per sample
            L.cap = ((3u * ips * bs / 2u + L.parts - 1u) / L.parts + 15u) & ~15u;   // 1.5x the expected fill
            L.counts_off = counts_off; counts_off += L.parts * nsb;
            L.bins_off = bins_off; bins_off += (uint64_t)L.parts * nsb * L.cap;
            L.ovf_off = ovf_off; ovf_off += (uint64_t)nsb * P.bin.ovf_cap;
            L.acc_block0 = P.bin.acc_blocks; P.bin.acc_blocks += L.parts;
        }
    P.bin.ovfcnt_off = counts_off; counts_off += P.bin.n_lv * nsb;
    P.bin.amax_off = counts_off; counts_off += P.bin.n_lv * nsb;
    P.bin.n_rl = P.rl.n_lv; P.rl.nsb = nsb; P.rl.n_bin_lv = P.bin.n_lv;
    P.bin.rl_slices = P.rl.rl_slices = (P.rl.n_lv && P.bin.n_lv >= P.rl.n_lv) ? (P.bin.n_lv + P.rl.n_lv - 1) / P.rl.n_lv : 1u;
    P.bin.ramax_off = P.rl.ramax_off = counts_off; counts_off += P.rl.n_lv * nsb * P.bin.rl_slices;
    for (uint32_t r = 0; r < P.rl.n_lv; ++r) P.bin.rl_drow[r] = P.rl.lv[r].drow;
    P.bin.n_bound = P.rl.n_bound = (nsb * (uint64_t)bs > 0xffffffffull) ? 0xffffffffu : nsb * bs;
    P.counts_bytes = (((size_t)counts_off * sizeof(uint32_t)) + 255) & ~(size_t)255;
    P.bins_bytes = (size_t)bins_off * sizeof(float4);
    P.ovf_bytes = (size_t)ovf_off * sizeof(float4);
    P.rl.blocks = P.rl.blocks;
    P.slabs_bytes = (size_t)P.rl.chunks * P.rl.slab_entries * sizeof(float2);
    return true;
}

size_t xr_scatter3_workspace_bytes(uint32_t n, const GridMeta& gm, uint32_t hashed_mask) {
    S3Layout P;
    if (!s3_layout(n, gm, hashed_mask, 0, &P)) return 0;
    return P.counts_bytes + P.bins_bytes + P.ovf_bytes + P.slabs_bytes;
}

uint32_t xr_scatter3_atomic_mask(uint32_t n, const GridMeta& gm, uint32_t hashed_mask, bool workspace_ok) {
    S3Layout P;
    if (!workspace_ok || !s3_layout(n, gm, hashed_mask, 0, &P)) return (1u << gm.n_levels) - 1u;
    return P.atomic_mask;
}

// A training step's small sums and updates ride inside the binning launch (xr_aux.h); the caller hands them over here
static thread_local XrAuxWork* g_aux_work = nullptr;
void xr_internal_scatter_aux_work(XrAuxWork* w) { g_aux_work = w; }
// The helper stream of earlier rounds (xr_set_helper_stream: a stream + its fork and join events, the caller's) is still what the
// atomic-level fallback of xr_hashgrid_bwd forks onto (xr_encode.hip); the binned scatter below no longer uses it: one in-order stream.
static thread_local XrHelper g_helper = {nullptr, nullptr, nullptr};
const XrHelper* xr_internal_helper() { return g_helper.stream ? &g_helper : nullptr; }
extern "C" int xr_set_helper_stream(void* stream, void* fork_event, void* join_event) {
    XR_REQUIRE(!stream || (fork_event && join_event), "a helper stream comes with its fork and join events");
    g_helper.stream = (hipStream_t)stream; g_helper.fork = (hipEvent_t)fork_event; g_helper.join = (hipEvent_t)join_event;
    return XR_OK;
}

int xr_scatter3(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                const uint32_t* rows, const GridMeta& gm, uint32_t hashed_mask, float* grad_table, void* workspace,
                size_t workspace_bytes, int overwrite, uint32_t* atomic_mask, hipStream_t stream, const XrAdamArgs* adam) {
    S3Layout P;
    if (!s3_layout(n, gm, hashed_mask, overwrite, &P) || !workspace || ((uintptr_t)workspace & 15) || ((uintptr_t)grad_table & 15)) {
        XR_REQUIRE(!adam, "the fused optimiser update needs every level on the non-atomic path");
        *atomic_mask = (1u << gm.n_levels) - 1u;
        return XR_OK;
    }
    if (adam) {
        XR_REQUIRE(P.atomic_mask == 0, "the fused optimiser update needs every level on the non-atomic path");
        P.bin.fuse = P.rl.fuse = 1u;
        P.bin.ad = P.rl.ad = *adam;
    }
    XR_REQUIRE(workspace_bytes >= P.counts_bytes + P.bins_bytes + P.ovf_bytes + P.slabs_bytes, "workspace too small");
    *atomic_mask = P.atomic_mask;
    uint32_t* counts = (uint32_t*)workspace;
    float4* bins = (float4*)((char*)workspace + P.counts_bytes);
    float4* ovf = (float4*)((char*)workspace + P.counts_bytes + P.bins_bytes);
    float2* slabs = (float2*)((char*)workspace + P.counts_bytes + P.bins_bytes + P.ovf_bytes);
    static thread_local bool attr_set = false;        // (idempotent: a second thread setting it again is harmless)
    if (!attr_set) {
        XR_HIP(hipFuncSetAttribute((const void*)k_scatter_acc<S3_LOG2, S3_ACC_THREADS_DEFAULT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS_BYTES));
        attr_set = true;
    }
    // Three launches on the caller's stream, no event (rounds 3-5: the small dense levels' two kernels on a helper stream beside the
    // bin / accumulate pair -- a fork before the bin kernel and a join behind the accumulate kernel, ~14 + ~20 us of the step's queue
    // per iteration, profiles/r06_trace_single_stream.txt):
    //   1. binning of the hashed / large dense levels  (+ the caller's gradient-partial sums and MLP update as extra workgroups, xr_aux.h)
    //   2. (the caller's loss scalars +) run-length workgroups of the small dense levels + one accumulate workgroup per (binned level, partition)
    //   3. fold of the run-length slabs (fixed order)
    XrAuxWork aux; memset(&aux, 0, sizeof(aux));
    uint32_t aux_blocks = 0;
    uint32_t n_tail = 0;
    if (g_aux_work && !g_aux_work->done && P.bin.n_lv) {
        aux = *g_aux_work; g_aux_work->done = true;
        aux_blocks = aux.partial ? xr_aux_reduce_blocks<S3_BIN_THREADS>(aux.gw) : 0u;
        n_tail = (aux.rgb || aux.clear) ? 1u : 0u;
    }
    if (P.bin.n_lv) {
        const uint32_t bs = s3_block_samples(), nb = P.bin.n_lv * P.bin.nsb;
        const dim3 grid(nb + aux_blocks), block(S3_BIN_THREADS);
        if (bs == 4096) hipLaunchKernelGGL(k_scatter_bin3<4096>, grid, block, 0, stream, P.bin, x, x_stride, denc_t, ld, n, n_dev, rows, counts, bins, ovf, aux_blocks, aux);
        else if (bs == 2048) hipLaunchKernelGGL(k_scatter_bin3<2048>, grid, block, 0, stream, P.bin, x, x_stride, denc_t, ld, n, n_dev, rows, counts, bins, ovf, aux_blocks, aux);
        else hipLaunchKernelGGL(k_scatter_bin3<1024>, grid, block, 0, stream, P.bin, x, x_stride, denc_t, ld, n, n_dev, rows, counts, bins, ovf, aux_blocks, aux);
        XR_LAUNCH_CHECK();
    }
    if (!P.bin.n_lv && P.rl.n_lv) {                   // nobody else records the run-length levels' maxima
        hipLaunchKernelGGL(k_scatter_rl_absmax, dim3(P.rl.n_lv * P.bin.nsb), dim3(S3_BIN_THREADS), 0, stream, P.bin, s3_block_samples(), denc_t, ld, n, n_dev, rows, counts);
        XR_LAUNCH_CHECK();
    }
    if (P.rl.blocks + P.bin.acc_blocks) {
        hipLaunchKernelGGL((k_scatter_acc<S3_LOG2, S3_ACC_THREADS_DEFAULT>), dim3(n_tail + P.rl.blocks + P.bin.acc_blocks), dim3(S3_ACC_THREADS_DEFAULT),
                           S3_LDS_BYTES, stream, P.bin, P.rl, x, x_stride, denc_t, ld, n, n_dev, rows, (const uint32_t*)counts, (const float4*)bins,
                           (const float4*)ovf, slabs, grad_table, n_tail, aux);
        XR_LAUNCH_CHECK();
    }
    if (P.rl.n_lv) {
        hipLaunchKernelGGL(k_scatter_fold, dim3(xr_div_up(P.rl.slab_entries, 256)), dim3(256), 0, stream, P.rl, (const float2*)slabs, grad_table);
        XR_LAUNCH_CHECK();
    }
    return XR_OK;
}
