// fp32 linear layers of the 8x256 NeRF MLP (configs #1 and #3: xrnerf/models/mlps/nerf_mlp.py:27-94): the fp32-MFMA kernel (k_gemm_f32, round 1)
// and, below it, the split-operand kernels that serve the products by default (k_gemm_split, k_gemm_split_kt: fp32 results on the 16-bit matrix cores).
//
// The reference runs them as nn.Linear; on MI355X that dispatches to hipBLASLt, which reaches 29 TFLOP/s on the shapes
// of this MLP (M = 32768..131072 samples, N = K = 256: `profiles/r01_mip_step_kernel_stats.csv`, 146 us per
// 32768x256x256) -- 18 % of the 157.3 TFLOP/s fp32 MFMA peak -- plus separate bias-gradient reductions and relu passes.
// One LDS-tiled kernel covers the three products of a layer:
//   forward          y  = act(x w^T + b)        C[M,N]  = A[M,K] . B[N,K]^T        (A row-major, B row-major = nn.Linear.weight)
//   backward, input  dx = dy w                  C[M,K'] = A[M,N'] . B[N',K']       (B already [contraction, output])
//   backward, weight dw = dy^T x                C[N',K'] = A[M,N']^T . B[M,K']     (both operands [contraction, output]; split over M)
// Workgroup = 128 x 128 output tile, 4 waves in 2 x 2, each 64 x 64 = 2 x 2 accumulators of v_mfma_f32_32x32x2_f32
// (exact fp32, an fmaf chain); operands staged through LDS as [k][m] / [k][n] panels of 32 k (32 KB), the next panel's
// global loads in flight while the current one is multiplied.  Bias, relu and the relu mask of the incoming gradient are
// fused into the epilogue / the A-panel load.
#include "xr_common.h"
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define GBM 128
#define GBN 128
#ifndef GBK
#define GBK 16                                   // k per panel (multiple of 8); 2 x 2 x GBK x 128 x 4 B of LDS per workgroup
#endif
#define GNJ (GBK / 8)                             // float4 per thread and panel
#define GPAD 0                                   // [k][row] panels are read and written along `row`: no padding needed (4 x 16 KB = 64 KB of LDS)

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;                           // nullable [Nc]
    const float* mask_src;                       // nullable, same shape / layout as A: A element counts only where mask_src > 0
    uint32_t Mc, Nc, Kc;                         // C is [Mc, Nc], contraction length Kc
    uint32_t lda, ldb, ldc;
    int a_km, b_kn;                              // A stored [Kc, Mc] (else [Mc, Kc]); B stored [Kc, Nc] (else [Nc, Kc])
    int relu;
    uint32_t k_per_split;                        // contraction range of blockIdx.z; C of split z at C + z * c_split_stride
    size_t c_split_stride;
    int tload;                                   // k_gemm_split: [k, rows] operands are read with per-k dword loads (k-contiguous in registers)
    float* colsum;                               // k_gemm_f32 with a_km: nullable [splits, Mc]: sums of A's (masked) rows over the split's k range
    int a_gradient = 0;                          // A holds gradients (any magnitude): a 16-bit split of it must keep the fp32 range (bf16 parts)
    size_t colsum_stride = 0;                    // floats between two splits' column sums (0 = Mc)
};

// one 128 x GBK panel of an operand into registers: GNJ float4 per thread.
//   row-major [rows, k] source (km == 0): thread -> row (t & 127), GBK/2 consecutive k starting at (t >> 7) * GBK/2
//   [k, rows] source        (km == 1): thread -> k = (t >> 5) + 8 j, 4 consecutive rows at (t & 31) * 4
__device__ __forceinline__ void panel_load(const float* __restrict__ P, const float* __restrict__ mask, uint32_t ld, int km,
                                           uint32_t row0, uint32_t rows, uint32_t k0, uint32_t k_end, float4 (&v)[GNJ]) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < GNJ; ++j) {
        uint32_t r, k;
        bool ok;
        size_t off;
        if (!km) { r = row0 + (t & 127); k = k0 + (t >> 7) * (GBK / 2) + 4 * j; ok = r < rows && k < k_end; off = (size_t)r * ld + k; }
        else { k = k0 + (t >> 5) + 8 * j; r = row0 + (t & 31) * 4; ok = r < rows && k < k_end; off = (size_t)k * ld + r; }
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            x = *reinterpret_cast<const float4*>(P + off);
            if (mask != nullptr) {
                const float4 m = *reinterpret_cast<const float4*>(mask + off);
                x.x = m.x > 0.f ? x.x : 0.f; x.y = m.y > 0.f ? x.y : 0.f; x.z = m.z > 0.f ? x.z : 0.f; x.w = m.w > 0.f ? x.w : 0.f;
            }
        }
        v[j] = x;
    }
}

// registers -> LDS panel S[k][row] (row stride GBM + GPAD)
__device__ __forceinline__ void panel_store(float* __restrict__ S, int km, const float4 (&v)[GNJ]) {
    const uint32_t t = threadIdx.x;
    constexpr int ST = GBM + GPAD;
#pragma unroll
    for (int j = 0; j < GNJ; ++j) {
        if (!km) {
            const uint32_t r = t & 127, k = (t >> 7) * (GBK / 2) + 4 * j;
            S[(k + 0) * ST + r] = v[j].x; S[(k + 1) * ST + r] = v[j].y; S[(k + 2) * ST + r] = v[j].z; S[(k + 3) * ST + r] = v[j].w;
        } else {
            const uint32_t k = (t >> 5) + 8 * j, r = (t & 31) * 4;
            *reinterpret_cast<float4*>(S + k * ST + r) = v[j];
        }
    }
}

__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g) {
    constexpr int ST = GBM + GPAD;
    __shared__ float sA[2][GBK * ST];
    __shared__ float sB[2][GBK * ST];
    // column tiles vary fastest: the workgroups that share an A row-panel run next to each other (L2 reuse of A)
    const uint32_t m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const uint32_t k_begin = blockIdx.z * g.k_per_split;
    const uint32_t k_end = k_begin + g.k_per_split < g.Kc ? k_begin + g.k_per_split : g.Kc;
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[GNJ], rb[GNJ];
    // the bias gradient rides on the weight-gradient product: A = (dy masked)^T is staged as [k = sample][row = output neuron] panels, a
    // thread's panel registers are always the same 4 neurons, so their running sum over the split's samples is 4 adds per float4 -- in
    // the first column tile's workgroups only; no second pass over dy and the relu mask (k_masked_colsum: 43 us per layer)
    const bool cs = g.colsum != nullptr && g.a_km && blockIdx.x == 0;                     // uniform
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto cs_add = [&]() {
#pragma unroll
        for (int j = 0; j < GNJ; ++j) { csum.x += ra[j].x; csum.y += ra[j].y; csum.z += ra[j].z; csum.w += ra[j].w; }
    };
    if (k_begin < k_end) {
        panel_load(g.A, g.mask_src, g.lda, g.a_km, m0, g.Mc, k_begin, k_end, ra);
        panel_load(g.B, nullptr, g.ldb, g.b_kn, n0, g.Nc, k_begin, k_end, rb);
        if (cs) cs_add();
    }
    int buf = 0;
    for (uint32_t k0 = k_begin; k0 < k_end; k0 += GBK) {
        panel_store(sA[buf], g.a_km, ra);
        panel_store(sB[buf], g.b_kn, rb);
        __syncthreads();                                   // panel `buf` complete; the other buffer was consumed one step ago
        if (k0 + GBK < k_end) {
            panel_load(g.A, g.mask_src, g.lda, g.a_km, m0, g.Mc, k0 + GBK, k_end, ra);
            panel_load(g.B, nullptr, g.ldb, g.b_kn, n0, g.Nc, k0 + GBK, k_end, rb);
            if (cs) cs_add();
        }
        const float* pa = sA[buf] + hi * ST + wm * 64 + col;
        const float* pb = sB[buf] + hi * ST + wn * 64 + col;
#pragma unroll
        for (int s = 0; s < GBK / 2; ++s) {
            const float a0 = pa[2 * s * ST], a1 = pa[2 * s * ST + 32];
            const float b0 = pb[2 * s * ST], b1 = pb[2 * s * ST + 32];
            acc[0][0] = GMFMA(a0, b0, acc[0][0]);
            acc[0][1] = GMFMA(a0, b1, acc[0][1]);
            acc[1][0] = GMFMA(a1, b0, acc[1][0]);
            acc[1][1] = GMFMA(a1, b1, acc[1][1]);
        }
        buf ^= 1;
    }
    // epilogue: lane l, register r of a 32x32 accumulator holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    float* C = g.C + (size_t)blockIdx.z * g.c_split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t n = n0 + wn * 64 + j * 32 + col;
            if (n >= g.Nc) continue;
            const float b = g.bias != nullptr ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < g.Mc) {
                    float v = acc[i][j][r] + b;
                    if (g.relu) v = fmaxf(v, 0.f);
                    C[(size_t)m * g.ldc + n] = v;
                }
            }
        }
    if (cs) {                                              // the 8 k-phases of a neuron's running sums, added in a fixed order
        __syncthreads();
        float* red = &sA[0][0];
        *reinterpret_cast<float4*>(red + (threadIdx.x >> 5) * GBM + (threadIdx.x & 31) * 4) = csum;
        __syncthreads();
        if (threadIdx.x < GBM) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += red[i * GBM + threadIdx.x];
            if (m0 + threadIdx.x < g.Mc) g.colsum[(size_t)blockIdx.z * (g.colsum_stride ? g.colsum_stride : (size_t)g.Mc) + m0 + threadIdx.x] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same three products on the bf16 matrix cores, to fp32 accuracy (csrc/xr_mlp.hip, "3-way operand splitting"): every
// fp32 operand element is split EXACTLY into three bf16 numbers while its panel is staged into LDS, and a 32 x 32 x 16 block
// of products is carried by the six bf16 x bf16 MFMA terms above 2^-23 of it (fp32 accumulate): 6 x 8 passes instead of the
// 8 x 16 passes of eight v_mfma_f32_32x32x2_f32.  Same tiling (128 x 128 per workgroup, 4 waves in 2 x 2, 2 x 2 accumulators
// each), same epilogue; panels are 32 k deep, stored [part][row][k] with 80-byte rows so that a lane's 8 consecutive k of
// one part are one conflict-free ds_read_b128 (60 KB per workgroup, two workgroups per CU; the next panel's global loads
// are in flight while the current one is multiplied).
typedef __bf16 gb8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb4 __attribute__((ext_vector_type(4)));
#define GMFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define G3K 32                                   // k per panel: two MFMA k-steps
#define G3RS (G3K + 8)                           // halves per LDS row (80 B)
#define G3PART (GBM * G3RS)                      // halves per part of one operand panel
#define G3NJ (G3K / 8)                           // float4 per thread and panel

// The split kernel's three arithmetics (one template, `GSplit<KIND>`):
//   G_B3  three bf16 parts, exact (8 + 8 + 8 significand bits), six product terms: fp32-rounding accuracy       (round 2)
//   G_H2  two fp16 parts (hi + lo ~ 22 bits), three terms: ~2^-21 of |a||b| per product, |operand| < 65504         (round 5: forward)
//   G_B2  two bf16 parts (16 bits), three terms: 2^-16 relative per product, fp32 range                           (round 5: dy . w)
enum { G_B3 = 0, G_H2 = 1, G_B2 = 2 };
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));
typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
template <int KIND> struct GSplit;
template <> struct GSplit<G_B3> {
    typedef __bf16 T; typedef gb4 V4; typedef gb8 V8;
    static constexpr int P = 3;
    static __device__ __forceinline__ void split(float x, T (&p)[3]) {
        p[0] = (__bf16)x;
        const float r1 = x - (float)p[0];        // exact
        p[1] = (__bf16)r1;
        p[2] = (__bf16)(r1 - (float)p[1]);       // exact remainder, rounded once
    }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct GSplit<G_H2> {
    typedef _Float16 T; typedef gh4 V4; typedef gh8 V8;
    static constexpr int P = 2;
    static __device__ __forceinline__ void split(float x, T (&p)[2]) { p[0] = (_Float16)x; p[1] = (_Float16)(x - (float)p[0]); }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct GSplit<G_B2> {
    typedef __bf16 T; typedef gb4 V4; typedef gb8 V8;
    static constexpr int P = 2;
    static __device__ __forceinline__ void split(float x, T (&p)[2]) { p[0] = (__bf16)x; p[1] = (__bf16)(x - (float)p[0]); }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
// panel_load for 32-k panels (same thread mapping as above with GBK = 32)
__device__ __forceinline__ void g3_panel_load(const float* __restrict__ P, const float* __restrict__ mask, uint32_t ld, int km,
                                              uint32_t row0, uint32_t rows, uint32_t k0, uint32_t k_end, float4 (&v)[G3NJ]) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < G3NJ; ++j) {
        uint32_t r, k;
        bool ok;
        size_t off;
        if (!km) { r = row0 + (t & 127); k = k0 + (t >> 7) * (G3K / 2) + 4 * j; ok = r < rows && k < k_end; off = (size_t)r * ld + k; }
        else { k = k0 + (t >> 5) + 8 * j; r = row0 + (t & 31) * 4; ok = r < rows && k < k_end; off = (size_t)k * ld + r; }
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            x = *reinterpret_cast<const float4*>(P + off);
            if (mask != nullptr) {
                const float4 m = *reinterpret_cast<const float4*>(mask + off);
                x.x = m.x > 0.f ? x.x : 0.f; x.y = m.y > 0.f ? x.y : 0.f; x.z = m.z > 0.f ? x.z : 0.f; x.w = m.w > 0.f ? x.w : 0.f;
            }
        }
        v[j] = x;
    }
}
// a [k, rows] source read the other way round: thread -> row (t & 127), 16 k's starting at (t >> 7) * 16, one dword per k
// (a wave's 64 rows are 256 contiguous bytes per k) -> the SAME register layout as the row-major case, so the panel goes
// to LDS with one 8-byte store per part and 4 k's instead of twelve 2-byte stores (measured with the float4-along-rows
// loads: dX 0.9x, dW 0.55x of the fp32-MFMA kernel, profiles/r02_gemm_bf16x3_vs_fp32_mfma_v1.txt)
__device__ __forceinline__ void g3_panel_load_t(const float* __restrict__ P, const float* __restrict__ mask, uint32_t ld,
                                                uint32_t row0, uint32_t rows, uint32_t k0, uint32_t k_end, float4 (&v)[G3NJ]) {
    const uint32_t t = threadIdx.x, r = row0 + (t & 127);
#pragma unroll
    for (int j = 0; j < G3NJ; ++j) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = k0 + (t >> 7) * (G3K / 2) + 4 * j + e;
            x[e] = 0.f;
            if (r < rows && k < k_end) {
                const size_t off = (size_t)k * ld + r;
                x[e] = P[off];
                if (mask != nullptr && !(mask[off] > 0.f)) x[e] = 0.f;
            }
        }
        v[j] = make_float4(x[0], x[1], x[2], x[3]);
    }
}
// registers -> LDS: S[part][row][k] (row stride G3RS halves)
template <int KIND>
__device__ __forceinline__ void g3_panel_store(typename GSplit<KIND>::T* __restrict__ S, int km, const float4 (&v)[G3NJ]) {
    using G = GSplit<KIND>;
    typedef typename G::T T;
    constexpr int P = G::P;
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < G3NJ; ++j) {
        const float x[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        if (!km) {                               // 4 consecutive k of one row: one 8-byte store per part
            const uint32_t r = t & 127, k = (t >> 7) * (G3K / 2) + 4 * j;
            typename G::V4 q[P];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                T parts[P];
                G::split(x[e], parts);
#pragma unroll
                for (int p = 0; p < P; ++p) q[p][e] = parts[p];
            }
            T* d = S + r * G3RS + k;
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<typename G::V4*>(d + p * G3PART) = q[p];
        } else {                                 // 4 consecutive rows at one k
            const uint32_t k = (t >> 5) + 8 * j, r = (t & 31) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                T parts[P];
                G::split(x[e], parts);
                T* d = S + (r + e) * G3RS + k;
#pragma unroll
                for (int p = 0; p < P; ++p) d[p * G3PART] = parts[p];
            }
        }
    }
}

// NB = 128-column blocks of the output tile: 1 = a 128 x 128 tile; 2 = 128 x 256 -- a layer's whole output width in one workgroup, so that
// the A panels (x; dy AND its relu mask in the input gradient) are read once instead of once per column tile (round 5)
template <int KIND, int NB = 1>
__global__ void __launch_bounds__(256, 2) k_gemm_split(GemmArgs g) {
    using G = GSplit<KIND>;
    typedef typename G::T T;
    typedef typename G::V8 V8;
    constexpr int P = G::P, NJ = 2 * NB;
    __shared__ __attribute__((aligned(16))) T sA[P * G3PART];
    __shared__ __attribute__((aligned(16))) T sB[NB * P * G3PART];
    const uint32_t m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN * NB;
    const uint32_t k_begin = blockIdx.z * g.k_per_split;
    const uint32_t k_end = k_begin + g.k_per_split < g.Kc ? k_begin + g.k_per_split : g.Kc;
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[G3NJ], rb[NB][G3NJ];
    const bool ta = g.a_km && g.tload, tb = g.b_kn && g.tload;             // uniform
    auto load = [&](uint32_t k0) {
        if (ta) g3_panel_load_t(g.A, g.mask_src, g.lda, m0, g.Mc, k0, k_end, ra);
        else g3_panel_load(g.A, g.mask_src, g.lda, g.a_km, m0, g.Mc, k0, k_end, ra);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (tb) g3_panel_load_t(g.B, nullptr, g.ldb, n0 + nb * GBN, g.Nc, k0, k_end, rb[nb]);
            else g3_panel_load(g.B, nullptr, g.ldb, g.b_kn, n0 + nb * GBN, g.Nc, k0, k_end, rb[nb]);
        }
    };
    if (k_begin < k_end) load(k_begin);
    for (uint32_t k0 = k_begin; k0 < k_end; k0 += G3K) {
        __syncthreads();                                   // the previous panel has been consumed
        g3_panel_store<KIND>(sA, ta ? 0 : g.a_km, ra);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) g3_panel_store<KIND>(sB + nb * P * G3PART, tb ? 0 : g.b_kn, rb[nb]);
        __syncthreads();
        if (k0 + G3K < k_end) load(k0 + G3K);
        const T* pa = sA + (wm * 64 + col) * G3RS + hi * 8;
        // this wave's 64 * NB columns: rows wn * 64 * NB + j * 32 + col of the B panel (block = row / 128)
        const T* pb = sB + (NB == 2 ? wn * P * G3PART + col * G3RS : (wn * 64 + col) * G3RS) + hi * 8;
#pragma unroll
        for (int s = 0; s < G3K / 16; ++s) {
            V8 a[2][P], b[NJ][P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const V8*>(pa + p * G3PART + i * 32 * G3RS + s * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j][p] = *reinterpret_cast<const V8*>(pb + p * G3PART + j * 32 * G3RS + s * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {             // smallest terms first
                    if constexpr (P == 3) {
                        acc[i][j] = G::mfma(a[i][2], b[j][0], acc[i][j]);
                        acc[i][j] = G::mfma(a[i][0], b[j][2], acc[i][j]);
                        acc[i][j] = G::mfma(a[i][1], b[j][1], acc[i][j]);
                    }
                    acc[i][j] = G::mfma(a[i][1], b[j][0], acc[i][j]);
                    acc[i][j] = G::mfma(a[i][0], b[j][1], acc[i][j]);
                    acc[i][j] = G::mfma(a[i][0], b[j][0], acc[i][j]);
                }
        }
    }
    float* C = g.C + (size_t)blockIdx.z * g.c_split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t n = n0 + wn * 64 * NB + j * 32 + col;
            if (n >= g.Nc) continue;
            const float b = g.bias != nullptr ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < g.Mc) {
                    float v = acc[i][j][r] + b;
                    if (g.relu) v = fmaxf(v, 0.f);
                    C[(size_t)m * g.ldc + n] = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The weight gradient (both operands [k = sample][row]) on the split kernel's arithmetic, round 5.  Its operands arrive k-major: a
// float4 of global memory is 4 consecutive ROWS at one k, while the MFMA wants 8 consecutive K of one row per lane.  Round 2 turned the
// panel around on its way INTO LDS (twelve 2-byte stores per float4) or read it with per-k dword loads; both lost to the fp32-MFMA
// kernel.  gfx950's transposing LDS load does the turn on the way OUT: the panel is stored as it arrives -- S[part][k][row], one 8-byte
// store per part and float4 -- and ds_read_b64_tr_b16 hands lane j of every 16-lane group element (j & 3) of the four 8-byte rows
// addressed by lanes (j >> 2) + 4 e (profiles/r02_ds_read_tr_b16_lane_mapping.txt): with lane a pointing at &S[k0 + (a >> 2)][i0 + 4 (a & 3)]
// lane j receives row i0 + j at k0 .. k0 + 3, i.e. two such loads are one 32 x 32 x 16 operand.  The bias gradient's column sums ride
// on the A panels exactly as in k_gemm_f32.
#define GKT_RS (GBM + 16)                        // halves per k-row (288 B): four consecutive k-rows start 8 banks apart
#define GKT_PART (G3K * GKT_RS)
typedef short gs4 __attribute__((ext_vector_type(4)));
typedef short gs8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void gkt_panel_store(typename GSplit<KIND>::T* __restrict__ S, const float4 (&v)[G3NJ]) {
    using G = GSplit<KIND>;
    typedef typename G::T T;
    constexpr int P = G::P;
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < G3NJ; ++j) {
        const float x[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        const uint32_t k = (t >> 5) + 8 * j, r = (t & 31) * 4;          // the [k, rows] mapping of g3_panel_load
        typename G::V4 q[P];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            T parts[P];
            G::split(x[e], parts);
#pragma unroll
            for (int p = 0; p < P; ++p) q[p][e] = parts[p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<typename G::V4*>(S + p * GKT_PART + k * GKT_RS + r) = q[p];
    }
}
// the MFMA operand of rows row0 .. row0 + 31 at k = kbase .. kbase + 15 out of one part of a k-major panel
template <int KIND>
__device__ __forceinline__ typename GSplit<KIND>::V8 gkt_operand(const typename GSplit<KIND>::T* __restrict__ S, int row0, int kbase) {
    const int lane = threadIdx.x & 63, grp = lane >> 4, j = lane & 15;
    const typename GSplit<KIND>::T* p = S + (kbase + (grp >> 1) * 8 + (j >> 2)) * GKT_RS + row0 + 16 * (grp & 1) + 4 * (j & 3);
    const gs4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(XR_LDS_PTR(gs4, p));
    const gs4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(XR_LDS_PTR(gs4, p + 4 * GKT_RS));
    const gs8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    typename GSplit<KIND>::V8 out;
    __builtin_memcpy(&out, &v, sizeof(out));
    return out;
}

template <int KIND, int NB = 1>
__global__ void __launch_bounds__(256, 2) k_gemm_split_kt(GemmArgs g) {
    using G = GSplit<KIND>;
    typedef typename G::T T;
    typedef typename G::V8 V8;
    constexpr int P = G::P, NJ = 2 * NB;
    __shared__ __attribute__((aligned(16))) T sA[P * GKT_PART];
    __shared__ __attribute__((aligned(16))) T sB[NB * P * GKT_PART];
    const uint32_t m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN * NB;
    const uint32_t k_begin = blockIdx.z * g.k_per_split;
    const uint32_t k_end = k_begin + g.k_per_split < g.Kc ? k_begin + g.k_per_split : g.Kc;
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[G3NJ], rb[NB][G3NJ];
    const bool cs = g.colsum != nullptr && blockIdx.x == 0;                // uniform: the bias gradient, first column tile only
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](uint32_t k0) {
        g3_panel_load(g.A, g.mask_src, g.lda, 1, m0, g.Mc, k0, k_end, ra);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) g3_panel_load(g.B, nullptr, g.ldb, 1, n0 + nb * GBN, g.Nc, k0, k_end, rb[nb]);
        if (cs) {
#pragma unroll
            for (int j = 0; j < G3NJ; ++j) { csum.x += ra[j].x; csum.y += ra[j].y; csum.z += ra[j].z; csum.w += ra[j].w; }
        }
    };
    if (k_begin < k_end) load(k_begin);
    for (uint32_t k0 = k_begin; k0 < k_end; k0 += G3K) {
        __syncthreads();                                   // the previous panel has been consumed
        gkt_panel_store<KIND>(sA, ra);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) gkt_panel_store<KIND>(sB + nb * P * GKT_PART, rb[nb]);
        __syncthreads();
        if (k0 + G3K < k_end) load(k0 + G3K);
        // this wave's 64 * NB columns of the B panel: block wn (NB == 2), or rows wn * 64 .. of the one block
        const T* pbw = sB + (NB == 2 ? wn * P * GKT_PART : 0);
        const int rowb = NB == 2 ? 0 : wn * 64;
#pragma unroll
        for (int s = 0; s < G3K / 16; ++s) {
            V8 a[2][P], b[NJ][P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][p] = gkt_operand<KIND>(sA + p * GKT_PART, wm * 64 + i * 32, s * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j][p] = gkt_operand<KIND>(pbw + p * GKT_PART, rowb + j * 32, s * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {             // smallest terms first
                    if constexpr (P == 3) {
                        acc[i][j] = G::mfma(a[i][2], b[j][0], acc[i][j]);
                        acc[i][j] = G::mfma(a[i][0], b[j][2], acc[i][j]);
                        acc[i][j] = G::mfma(a[i][1], b[j][1], acc[i][j]);
                    }
                    acc[i][j] = G::mfma(a[i][1], b[j][0], acc[i][j]);
                    acc[i][j] = G::mfma(a[i][0], b[j][1], acc[i][j]);
                    acc[i][j] = G::mfma(a[i][0], b[j][0], acc[i][j]);
                }
        }
    }
    float* C = g.C + (size_t)blockIdx.z * g.c_split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t n = n0 + wn * 64 * NB + j * 32 + col;
            if (n >= g.Nc) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < g.Mc) C[(size_t)m * g.ldc + n] = acc[i][j][r];
            }
        }
    if (cs) {                                              // the 8 k-phases of a neuron's running sums, added in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(sA);
        *reinterpret_cast<float4*>(red + (threadIdx.x >> 5) * GBM + (threadIdx.x & 31) * 4) = csum;
        __syncthreads();
        if (threadIdx.x < GBM) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += red[i * GBM + threadIdx.x];
            if (m0 + threadIdx.x < g.Mc) g.colsum[(size_t)blockIdx.z * (g.colsum_stride ? g.colsum_stride : (size_t)g.Mc) + m0 + threadIdx.x] = sum;
        }
    }
}

static int gemm_launch(GemmArgs g, uint32_t splits, void* stream) {
    XR_REQUIRE(g.A && g.B && g.C, "null pointer");
    XR_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "leading dimensions must be multiples of 4 floats (16-byte vector loads)");
    XR_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0 && (g.mask_src == nullptr || ((uintptr_t)g.mask_src & 15) == 0),
               "operands must be 16-byte aligned");
    XR_REQUIRE((g.a_km ? g.Mc : g.Kc) % 4 == 0 && (g.b_kn ? g.Nc : g.Kc) % 4 == 0, "the contiguous dimension of each operand must be a multiple of 4");
    if (g.Mc == 0 || g.Nc == 0) return XR_OK;
    XR_REQUIRE(splits >= 1 && splits <= 65535, "bad split count");
    // XR_GEMM_F32 (read per call: a test or a measurement can switch between two launches):
    //   unset / split2 : products whose operands are both row-major [rows, k] -- the forward, and the input gradient with the weight handed
    //                    over transposed -- on the 16-bit matrix cores with 2-way split operands, three MFMAs per product block: fp16 parts for
    //                    the forward (~2^-21 of |x||w| per product; |operand| < 65504), bf16 parts where A holds gradients (2^-16 relative per
    //                    product, fp32 range); the weight gradient -- both operands k-major -- on bf16 parts as well, its panels turned around by
    //                    the transposing LDS load (k_gemm_split_kt)
    //   bf16x3         : the same products on EXACT 3-way bf16 operands, six MFMAs per product block (fp32-rounding accuracy; round 2-4's default)
    //   bf16x3all      : the 3-way split kernel for all three products ([k, rows] operands read with per-k dword loads)
    //   mfma           : the fp32-MFMA kernel throughout
    // Anything else is an error.  NOTE the split kernels' operand range: an Inf (or a magnitude within 2^-8 of FLT_MAX, whose
    // bf16 head rounds to Inf; beyond 65504 for fp16 parts) turns into NaN inside the split (Inf - Inf); the fp32-MFMA kernel propagates it
    // like an fmaf chain.
    const char* env = getenv("XR_GEMM_F32");
    const bool dflt = !env || !env[0] || strcmp(env, "split2") == 0;
    const bool b3 = env && strcmp(env, "bf16x3") == 0;
    const bool all = env && strcmp(env, "bf16x3all") == 0;
    const bool mfma = env && strcmp(env, "mfma") == 0;
    XR_REQUIRE(dflt || b3 || all || mfma, "XR_GEMM_F32 must be split2, bf16x3, bf16x3all or mfma");
    const bool kt = dflt && g.a_km && g.b_kn;                 // the weight gradient: k-major operands through the transposing LDS load
    const bool split = !mfma && (all || kt || (!g.a_km && !g.b_kn));
    g.tload = all ? 1 : 0;
    const uint32_t kb = split ? (uint32_t)G3K : (uint32_t)GBK;
    g.k_per_split = (uint32_t)(((uint64_t)(g.Kc + splits - 1) / splits + kb - 1) / kb * kb);
    if (g.k_per_split == 0) g.k_per_split = kb;
    dim3 grid(xr_div_up(g.Nc, GBN), xr_div_up(g.Mc, GBM), splits);
    XR_REQUIRE(grid.y <= 65535, "more than 65535 row tiles (8.3 M rows) in one call");
    // 128 x 256 tiles: the A panels (x; dy and its mask) are read once per 256 output columns.  The weight gradient takes them only when the
    // last 256-column block is more than half full (N' x 352: 273 us wide against 242 us narrow; 256 x 256: 131 against 185)
    const bool wide = split && dflt && g.Nc > GBN && (!kt || xr_div_up(g.Nc, 2 * GBN) * 2 * GBN - g.Nc < GBN);
    if (wide) grid.x = xr_div_up(g.Nc, 2 * GBN);
    if (kt && wide) hipLaunchKernelGGL((k_gemm_split_kt<G_B2, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (kt) hipLaunchKernelGGL(k_gemm_split_kt<G_B2>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (wide && g.a_gradient) hipLaunchKernelGGL((k_gemm_split<G_B2, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (wide) hipLaunchKernelGGL((k_gemm_split<G_H2, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (split && dflt && g.a_gradient) hipLaunchKernelGGL(k_gemm_split<G_B2>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (split && dflt) hipLaunchKernelGGL(k_gemm_split<G_H2>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (split) hipLaunchKernelGGL(k_gemm_split<G_B3>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, (hipStream_t)stream, g);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// y [M,N] = act(x [M,K] . w [N,K]^T + bias [N])
// ldx / ldy: row strides of x and y in floats (0 = dense; multiples of 4): a layer can read its input from, and write its output into, a
// column range of a wider buffer -- the skip connection's [x | h] and the view layer's [feature | alpha | dir] are then never concatenated
extern "C" int xr_linear_forward(const float* x, uint32_t ldx, const float* w, const float* bias, uint32_t M, uint32_t N, uint32_t K,
                                 int relu, float* y, uint32_t ldy, void* stream) {
    if (ldx == 0) ldx = K;
    if (ldy == 0) ldy = N;
    XR_REQUIRE(ldx >= K && ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0, "bad row stride / output alignment");
    GemmArgs g{x, w, y, bias, nullptr, M, N, K, ldx, K, ldy, 0, 0, relu, 0, 0, 0, nullptr};
    return gemm_launch(g, 1, stream);
}

// dx [M,K] = (dy [M,N] masked by mask_src [M,N] > 0 when given) . w [N,K].  w_transposed != 0: the weight is handed over TRANSPOSED
// (w_t [K,N] row-major): both operands are then [rows, contraction] like the forward's, and the product runs on the split-operand kernel
// (163 us against 267 us on the fp32 MFMA at 131072 x 256 x 256)
// lddy: row stride of dy AND of mask_src (0 = dense): the gradient of a layer whose output sits in a column range of a wider buffer is the
// same column range of the next layer's input gradient
extern "C" int xr_linear_backward_input(const float* dy, uint32_t lddy, const float* mask_src, const float* w, int w_transposed, uint32_t M, uint32_t N,
                                        uint32_t K, float* dx, void* stream) {
    if (lddy == 0) lddy = N;
    XR_REQUIRE(lddy >= N, "bad row stride");
    if (w_transposed) {
        GemmArgs g{dy, w, dx, nullptr, mask_src, M, K, N, lddy, N, K, 0, 0, 0, 0, 0, 0, nullptr};
        g.a_gradient = 1;
        return gemm_launch(g, 1, stream);
    }
    GemmArgs g{dy, w, dx, nullptr, mask_src, M, K, N, lddy, K, K, 0, 1, 0, 0, 0, 0, nullptr};
    g.a_gradient = 1;
    return gemm_launch(g, 1, stream);
}

// dw_partials [splits, N, K]: partial sums of (dy masked)^T . x over `splits` ranges of the M rows; the caller adds
// the partials (fixed order => bit-reproducible).  splits = xr_linear_backward_splits(M, N, K); N == K == 0: the bias gradient's.
extern "C" uint32_t xr_linear_backward_splits(uint32_t M, uint32_t N, uint32_t K) {
    if (N == 0 && K == 0) { const uint32_t b = M / 128; return b < 1 ? 1 : (b > 512 ? 512 : b); }
    // the output is only ceil(N/128) x ceil(K/128) tiles (4 for a 256 x 256 layer): split the M rows until ~1024
    // workgroups exist, but keep >= 8 k-panels (256 rows) per split (measured: 16 splits of 2048 rows = 64 workgroups
    // left 3/4 of the chip idle, 110-290 us per 32768-row call)
    const uint32_t tiles = xr_div_up(N, GBM) * xr_div_up(K, GBN);
#ifndef XR_GEMM_SPLIT_WGS
#define XR_GEMM_SPLIT_WGS 1024
#endif
    uint32_t s = XR_GEMM_SPLIT_WGS / (tiles ? tiles : 1);
    const uint32_t by_rows = M / 256;
    if (s > by_rows) s = by_rows;
    return s < 1 ? 1 : (s > 4096 ? 4096 : s);
}
// db partials [splits, N]: column sums of (dy where mask_src > 0) over `splits` ranges of the M rows.
// thread = 4 consecutive columns, workgroup = 64 column groups x 4 row phases; 16-byte loads, coalesced along N.
__global__ void __launch_bounds__(256) k_masked_colsum(const float* __restrict__ dy, const float* __restrict__ mask,
                                                       uint32_t M, uint32_t N, uint32_t rows_per_split,
                                                       float* __restrict__ out, uint32_t ld = 0 /* row stride of dy / mask, 0 = N */,
                                                       size_t out_stride = 0 /* floats between two splits' sums, 0 = N */) {
    if (ld == 0) ld = N;
    if (out_stride == 0) out_stride = N;
    __shared__ float4 s_part[256];
    const uint32_t cg = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.y * rows_per_split;
    const uint32_t r1 = r0 + rows_per_split < M ? r0 + rows_per_split : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cg * 4 < N) {
        for (uint32_t r = r0 + ph; r < r1; r += 4) {
            const size_t off = (size_t)r * ld + cg * 4;
            float4 v = *reinterpret_cast<const float4*>(dy + off);
            if (mask != nullptr) {
                const float4 m = *reinterpret_cast<const float4*>(mask + off);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0 && cg * 4 < N) {
        float4 t = s_part[threadIdx.x];
        for (int p = 1; p < 4; ++p) { const float4 o = s_part[threadIdx.x + 64 * p]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * out_stride + cg * 4) = t;
    }
}

extern "C" int xr_linear_backward_bias(const float* dy, const float* mask_src, uint32_t M, uint32_t N, uint32_t splits,
                                       float* db_partials, void* stream) {
    XR_REQUIRE(dy && db_partials, "null pointer");
    XR_REQUIRE(N % 4 == 0 && splits >= 1 && splits <= 65535, "N must be a multiple of 4; bad split count");
    XR_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)db_partials & 15) == 0 && (mask_src == nullptr || ((uintptr_t)mask_src & 15) == 0),
               "operands must be 16-byte aligned");
    if (N == 0) return XR_OK;
    const uint32_t rows = (M + splits - 1) / splits;
    hipLaunchKernelGGL(k_masked_colsum, dim3(xr_div_up(N / 4, 64), splits), dim3(256), 0, (hipStream_t)stream, dy, mask_src, M,
                       N, rows ? rows : 1, db_partials);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// db_partials (nullable) [splits, N]: weight AND bias gradient of a layer in one launch -- the same M-range column sums of (dy masked) that
// xr_linear_backward_bias computes with a pass of its own, taken from the panels the weight-gradient product stages anyway
// lddy (dy and mask_src) / ldx: row strides, 0 = dense.  part_stride: floats between two splits' partials (0 = N * K); with db_partials ==
// dw_partials + N * K and part_stride = N * K + N both partial sets sit in ONE [splits, N * K + N] buffer and one reduction finishes both
extern "C" int xr_linear_backward_weight(const float* dy, uint32_t lddy, const float* mask_src, const float* x, uint32_t ldx, uint32_t M, uint32_t N,
                                         uint32_t K, uint32_t splits, float* dw_partials, float* db_partials, size_t part_stride, void* stream) {
    if (lddy == 0) lddy = N;
    if (ldx == 0) ldx = K;
    if (part_stride == 0) part_stride = (size_t)N * K;
    XR_REQUIRE(lddy >= N && ldx >= K && part_stride >= (size_t)N * K && part_stride % 4 == 0, "bad stride");
    const char* env = getenv("XR_GEMM_F32");
    if (db_partials && env && strcmp(env, "bf16x3all") == 0) {   // that measurement mode has no column sums in its kernel: two launches
        GemmArgs g0{dy, x, dw_partials, nullptr, mask_src, N, K, M, lddy, ldx, K, 1, 1, 0, 0, part_stride, 0, nullptr};
        const int rc = gemm_launch(g0, splits, stream);
        if (rc != XR_OK) return rc;
        // per-split column sums over the same M ranges: k_masked_colsum with `splits` row ranges of k_per_split rows
        const uint32_t rows = (uint32_t)(((uint64_t)(M + splits - 1) / splits + GBK - 1) / GBK * GBK);
        hipLaunchKernelGGL(k_masked_colsum, dim3(xr_div_up(N / 4, 64), splits), dim3(256), 0, (hipStream_t)stream, dy, mask_src, M, N, rows ? rows : 1,
                           db_partials, lddy, db_partials == dw_partials + (size_t)N * K ? part_stride : (size_t)N);
        XR_LAUNCH_CHECK();
        return XR_OK;
    }
    GemmArgs g{dy, x, dw_partials, nullptr, mask_src, N, K, M, lddy, ldx, K, 1, 1, 0, 0, part_stride, 0, db_partials};
    g.colsum_stride = db_partials ? (db_partials == dw_partials + (size_t)N * K ? part_stride : (size_t)N) : 0;
    return gemm_launch(g, splits, stream);
}
