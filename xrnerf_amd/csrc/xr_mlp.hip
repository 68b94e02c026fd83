// Fully fused tiny MLP of the Instant-NGP path (density_net + SH-4 + color_net), forward and
// backward, on the CDNA4 matrix cores -- the only GEMM-shaped work on the path.
//
// Replaces tcnn.Network(FullyFusedMLP) x2 + tcnn.Encoding(SphericalHarmonics) as called from
// /root/reference/xrnerf/models/mlps/hashnerf_mlp.py:39-45,55-79,107-111.
//
// Measured dead end (round 1): fetching the next tile's 23 inputs during the current tile -- in registers (spills,
// 0.213 -> 0.255 ms) or straight into LDS with global_load_lds (no gain: 0.199 vs 0.197 ms) -- does not pay.
//
// Design (gfx950, wave64):
//   * everything is computed TRANSPOSED: neurons x samples.  One wave owns a tile of 32 samples;
//     v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate == an fmaf chain, exact fp32) produces a
//     32(neurons) x 32(samples) tile whose C/D register layout is
//         lane l: column (sample) = l & 31,  row (neuron) = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
//     The B operand of the NEXT layer wants  B[k = l>>5][n = l&31]  for a pair of k: because a
//     sum over k may be taken in any order, K-step `reg` simply uses the pair of neurons that
//     register `reg` already holds in the two lane halves -- so hidden activations NEVER leave
//     registers, there is no LDS round trip and no shuffle between layers.  The matching
//     permutation is applied to the A operand (the weights) purely through LDS addressing.
//   * weights live in LDS once per workgroup, row-major [out][in] with an odd row stride (in+1):
//     conflict-free for the forward access (32 lanes = 32 rows) and for the transposed access of
//     the backward chain (32 lanes = 32 consecutive columns).
//   * backward recomputes the forward activations in registers (nothing is saved by the
//     forward: 0 B/sample of HBM instead of 768 B), runs the dX chain with the same trick on W^T,
//     and forms dW = G * H^T on the matrix cores as well; that contraction runs over SAMPLES, i.e.
//     it needs both operands transposed, which is the one place a per-wave LDS staging tile
//     ([neuron][33]) is used.  dW accumulates in registers across all tiles a wave processes,
//     then block-reduces through LDS (waves take turns, plain adds) into one partial per workgroup; a second tiny
//     kernel sums the partials (fixed order).
//   * the color net's input is cat(density_out[1:16], SH16) + one pad lane (tcnn pads the
//     Identity-encoded input with 1.0).  In "slot" space m = 0..31 we use column (m+31)%32 of the
//     first color layer, i.e. slot m>=1 is input m-1 and slot 0 is the pad: then slot m <-> density
//     output row m for m<16, so density_out registers feed the color net in place and its input
//     gradient lands back on the density-output registers with no data movement either.
#include "xr_common.h"
#include "xr_aux.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define MLP_WAVES 4
#define MLP_THREADS (MLP_WAVES * 64)
#define W_HID 64          // n_neurons of the reference config (nerf_blender_local01.py:106-124)
#define ENC_DIM 32        // 16 levels x 2 features
#define ST33 33
#define XR_MLP_MAX_LD 200000000u   // load_enc_tile / store_enc_tile: the per-lane byte offset (5 ld) * 4 stays below 2^32

__device__ __forceinline__ constexpr int drow(int r) { return (r & 3) + 8 * (r >> 2); }

// ---- topology bookkeeping --------------------------------------------------------------------
// layer list of one network: in(32) -> NH x 64 -> out(16, padded to a 32-row tile in LDS)
__host__ __device__ constexpr int net_in_dim(int l) { return l == 0 ? 32 : W_HID; }
__host__ __device__ constexpr int net_out_dim(int nh, int l) { return l == nh ? 16 : W_HID; }        // global rows
__host__ __device__ constexpr int net_out_rows_lds(int nh, int l) { return l == nh ? 32 : W_HID; }   // padded rows
__host__ __device__ constexpr int net_stride(int l) { return net_in_dim(l) + 1; }
__host__ __device__ constexpr int net_lds_off(int nh, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += net_out_rows_lds(nh, i) * net_stride(i);
    return o;
}
__host__ __device__ constexpr int net_glb_off(int nh, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += net_out_dim(nh, i) * net_in_dim(i);
    return o;
}
template <int NH>
struct NetShape {
    static constexpr int n_mats = NH + 1;
    __host__ __device__ static constexpr int in_dim(int l) { return net_in_dim(l); }
    __host__ __device__ static constexpr int out_dim(int l) { return net_out_dim(NH, l); }
    __host__ __device__ static constexpr int out_rows_lds(int l) { return net_out_rows_lds(NH, l); }
    __host__ __device__ static constexpr int stride(int l) { return net_stride(l); }
    __host__ __device__ static constexpr int lds_off(int l) { return net_lds_off(NH, l); }
    __host__ __device__ static constexpr int glb_off(int l) { return net_glb_off(NH, l); }
    static constexpr int lds_floats = net_lds_off(NH, NH + 1);
    static constexpr int glb_floats = net_glb_off(NH, NH + 1);
};

// copy a network's weights global -> LDS (padded stride, zero-padded output rows).
// first_layer_rot: apply the color-net slot permutation  LDS[o][m] = W[o][(m+31)&31].
// Two phases: EVERY global load of the network is issued before the first LDS store (up to 32 registers per thread).
// A plain load -> store loop compiles to one exposed L2 round trip per iteration, 40 of them in a row per workgroup:
// measured as ~25 us of fixed time in a backward launch whose tile work is 80-160 us.
template <int NH, int L>
__device__ __forceinline__ void fetch_layer(float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                            const float* __restrict__ w, bool first_layer_rot) {
    using S = NetShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L);
    static_assert(prow * K % MLP_THREADS == 0, "layer size must be a multiple of the workgroup size");
    const float* src = w + S::glb_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int e = threadIdx.x + i * MLP_THREADS, o = e / K, m = e % K;
        const int col = (L == 0 && first_layer_rot) ? ((m + 31) & 31) : m;
        v[i] = src[(o < rows ? o : rows - 1) * K + col];                  // padded rows: any valid address, zeroed on store
    }
}
template <int NH, int L>
__device__ __forceinline__ void store_layer(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                            float* __restrict__ lds) {
    using S = NetShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L), st = S::stride(L);
    float* dst = lds + S::lds_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int e = threadIdx.x + i * MLP_THREADS, o = e / K, m = e % K;
        dst[o * st + m] = o < rows ? v[i] : 0.f;
    }
}
template <int NH>
__device__ inline void load_weights(float* __restrict__ lds, const float* __restrict__ w, bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH >= 1 && NH <= 3, "1..3 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / MLP_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / MLP_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / MLP_THREADS : 1];
    float v3[NH >= 3 ? S::out_rows_lds(NH >= 3 ? 3 : 0) * S::in_dim(NH >= 3 ? 3 : 0) / MLP_THREADS : 1];
    fetch_layer<NH, 0>(v0, w, first_layer_rot);
    fetch_layer<NH, 1>(v1, w, false);
    if constexpr (NH >= 2) fetch_layer<NH, 2>(v2, w, false);
    if constexpr (NH >= 3) fetch_layer<NH, 3>(v3, w, false);
    store_layer<NH, 0>(v0, lds);
    store_layer<NH, 1>(v1, lds);
    if constexpr (NH >= 2) store_layer<NH, 2>(v2, lds);
    if constexpr (NH >= 3) store_layer<NH, 3>(v3, lds);
}

// ---- building blocks (all tiles in the C/D register layout described above) ------------------
// out[TO] = W[TO*32][TI*32] . in[TI]
template <int TI, int TO, bool SB = true>
__device__ __forceinline__ void layer_fwd(const float* __restrict__ W, const f32x16 (&in)[TI], f32x16 (&out)[TO], int col, int hi) {
    constexpr int ST = TI * 32 + 1;
    constexpr int NS = TI * 16;                 // K-steps
    constexpr int PF = 3;                       // LDS operand prefetch distance (steps); one ds_read ~ 64-128 cycles
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[to][r] = 0.f;
    const float* wl = W + col * ST + 4 * hi;
    float a[PF + 1][TO];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int to = 0; to < TO; ++to) a[p][to] = wl[to * 32 * ST + (p / 16) * 32 + drow(p % 16)];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        // keep the scheduler from hoisting a whole layer's LDS operand reads (64-128 VGPRs) to the top
        if (SB && (st & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        if (st + PF < NS) {
#pragma unroll
            for (int to = 0; to < TO; ++to) a[(st + PF) % (PF + 1)][to] = wl[to * 32 * ST + ((st + PF) / 16) * 32 + drow((st + PF) % 16)];
        }
        const float b = in[st / 16][st % 16];
#pragma unroll
        for (int to = 0; to < TO; ++to) out[to] = MFMA32(a[st % (PF + 1)][to], b, out[to]);
    }
}
// gin[TI] = W^T . g[TO]     (W is [TO*32][TI*32], stride TI*32+1).  RSTEPS < 16: only the first RSTEPS
// register rows of g can be non-zero (output layers: 16 real neurons -> 8, colour gradient -> 3), the
// K-steps that would multiply zeros are not issued.
template <int TO, int TI, bool SB = true, int RSTEPS = 16>
__device__ __forceinline__ void layer_bwd(const float* __restrict__ W, const f32x16 (&g)[TO], f32x16 (&gin)[TI], int col, int hi) {
    constexpr int ST = TI * 32 + 1;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) gin[ti][r] = 0.f;
    const float* wl = W + (4 * hi) * ST + col;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < RSTEPS; ++r) {
            if (SB && (r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const float b = g[to][r];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const float a = wl[(to * 32 + drow(r)) * ST + ti * 32];
                gin[ti] = MFMA32(a, b, gin[ti]);
            }
        }
}
__device__ __forceinline__ void relu_tile(f32x16& t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = t[r] > 0.f ? t[r] : 0.f;
}
__device__ __forceinline__ void relu_mask(f32x16& g, const f32x16& h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = h[r] > 0.f ? g[r] : 0.f;
}
// acc[to][ti] += g[to] (rows = out neurons) x h[ti]^T (rows = in neurons), contraction over the
// 32 samples of the tile; `stage` is this wave's private LDS tile [(TO+TI)*32][33].
// Split in two so that the caller can put the layer's dX chain (weights from LDS, independent of the staging
// tile) between the transposing writes and the reads: PMC showed 23 % of the backward's wave cycles waiting
// on LDS instructions with write -> read back to back.
template <int TO, int TI>
__device__ __forceinline__ void dw_stage(const f32x16 (&g)[TO], const f32x16 (&h)[TI], float* __restrict__ stage, int col, int hi) {
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(to * 32 + drow(r) + 4 * hi) * ST33 + col] = g[to][r];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((TO + ti) * 32 + drow(r) + 4 * hi) * ST33 + col] = h[ti][r];
    __builtin_amdgcn_wave_barrier();   // same-wave LDS ops complete in order; keep the compiler from reordering
}
template <int TO, int TI>
__device__ __forceinline__ void dw_mfma(f32x16 (&acc)[TO][TI], const float* __restrict__ stage, int col, int hi) {
    constexpr int PF = 2;              // operand prefetch distance in K-steps
    const float* sg = stage + col * ST33 + hi;
    float a[PF + 1][TO], b[PF + 1][TI];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int to = 0; to < TO; ++to) a[p][to] = sg[(to * 32) * ST33 + 2 * p];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) b[p][ti] = sg[((TO + ti) * 32) * ST33 + 2 * p];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        if (t + PF < 16) {
#pragma unroll
            for (int to = 0; to < TO; ++to) a[(t + PF) % (PF + 1)][to] = sg[(to * 32) * ST33 + 2 * (t + PF)];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) b[(t + PF) % (PF + 1)][ti] = sg[((TO + ti) * 32) * ST33 + 2 * (t + PF)];
        }
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) acc[to][ti] = MFMA32(a[t % (PF + 1)][to], b[t % (PF + 1)][ti], acc[to][ti]);
    }
    __builtin_amdgcn_wave_barrier();
}
template <int TO, int TI, bool SB = true>
__device__ __forceinline__ void dw_accumulate(f32x16 (&acc)[TO][TI], const f32x16 (&g)[TO], const f32x16 (&h)[TI],
                                              float* __restrict__ stage, int col, int hi) {
    dw_stage<TO, TI>(g, h, stage, col, hi);
    dw_mfma<TO, TI>(acc, stage, col, hi);
}
// ---- 16-bit operand arrangements in LDS (shared by the fp16 kernels and the bf16-split ones below)
__host__ __device__ constexpr int hrow(int t, int hi, int e) { return 32 * (t >> 1) + 16 * (t & 1) + 8 * (e >> 2) + (e & 3) + 4 * hi; }
__host__ __device__ constexpr int h_rs(int k_dim) { return 2 * (k_dim / 16) * 8 + 8; }       // halves per row, 16-B pad

// layer l of a (1 | 2)-hidden network in the two LDS arrangements (halves)
template <int NH>
struct HShape {
    using S = NetShape<NH>;
    __host__ __device__ static constexpr int f_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += S::out_rows_lds(i) * h_rs(S::in_dim(i)); return o; }
    __host__ __device__ static constexpr int b_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += S::in_dim(i) * h_rs(S::out_rows_lds(i)); return o; }
    static constexpr int f_halves = f_off(NH + 1);
    static constexpr int b_halves = b_off(NH + 1);
};

// global fp32 [out][in] -> LDS fp16, forward arrangement (and the transposed one when `wb` is given).  The loop runs over the
// SOURCE elements (coalesced global reads); the k-slot of neuron m is the inverse of hrow: e = m[1:0] | m[3] << 2,
// hi = m[2], t = m >> 4.
__device__ __forceinline__ int hslot(int m, int ns) {           // offset of neuron m inside a [hi][t][8] row of ns steps
    const int e = (m & 3) | (((m >> 3) & 1) << 2), hi = (m >> 2) & 1, t = m >> 4;
    return (hi * ns + t) * 8 + e;
}
// dW on the bf16 matrix cores (XR_MLP_BWD_DW=b2, review item 3 of round 2): the same staging tile, each operand read as 8
// consecutive samples of one neuron and split in registers into two bf16 parts x = xh + xl (xh = bf16(x) round-to-nearest,
// xl = bf16(x - xh), the difference exact in fp32).  Three v_mfma_f32_32x32x16_bf16 per 16 samples keep
// gh*hh + gh*hl + gl*hh (every bf16 x bf16 product exact, fp32 accumulate, small terms first); the dropped gl*hl term and the
// rounding of the low parts are below 2^-16 of |g||h| per term -- inside the 1e-3 * max bar the gradients are tested to
// (tests/test_gpu_tcnn.py::test_nerf_mlp_bwd*).  6 matrix instructions of 8 passes instead of 16 of 16 per output tile.
typedef __bf16 bw8 __attribute__((ext_vector_type(8)));
// 2-vectors for the forward's 3-way split (split3_pair below).  The 2-way splits of this kernel stay element by element: written
// on pairs they are 181 instructions fewer per tile and the kernel is 3 us SLOWER (66 vs 63 us in the loop, one measurement).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2x8(const float* __restrict__ p, bw8& h, bw8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = p[e];
        const __bf16 hh = (__bf16)v;
        h[e] = hh;
        l[e] = (__bf16)(v - (float)hh);
    }
}
template <int TO, int TI>
__device__ __forceinline__ void dw_mfma_b2(f32x16 (&acc)[TO][TI], const float* __restrict__ stage, int col, int hi) {
    const float* sg = stage + col * ST33 + 8 * hi;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bw8 ah[TO], al[TO], bh[TI], bl[TI];
#pragma unroll
        for (int to = 0; to < TO; ++to) split2x8(sg + (to * 32) * ST33 + 16 * t, ah[to], al[to]);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) split2x8(sg + ((TO + ti) * 32) * ST33 + 16 * t, bh[ti], bl[ti]);
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[to], bh[ti], acc[to][ti], 0, 0, 0);
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bl[ti], acc[to][ti], 0, 0, 0);
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bh[ti], acc[to][ti], 0, 0, 0);
            }
    }
    __builtin_amdgcn_wave_barrier();
}
template <bool B2, int TO, int TI>
__device__ __forceinline__ void dw_product(f32x16 (&acc)[TO][TI], const float* __restrict__ stage, int col, int hi) {
    if constexpr (B2) dw_mfma_b2<TO, TI>(acc, stage, col, hi);
    else dw_mfma<TO, TI>(acc, stage, col, hi);
}
// one H tile at a time (the c1 layer under the 3-tile staging area of the bf16 dX mode): acc[.][TIX] += G (tiles 0..TO-1 of the
// staging area) x H (tile TO)
template <int TO, int TIA, int TIX>
__device__ __forceinline__ void dw_mfma_b2_col(f32x16 (&acc)[TO][TIA], const float* __restrict__ stage, int col, int hi) {
    const float* sg = stage + col * ST33 + 8 * hi;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bw8 ah[TO], al[TO], bh, bl;
#pragma unroll
        for (int to = 0; to < TO; ++to) split2x8(sg + (to * 32) * ST33 + 16 * t, ah[to], al[to]);
        split2x8(sg + (TO * 32) * ST33 + 16 * t, bh, bl);
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            acc[to][TIX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[to], bh, acc[to][TIX], 0, 0, 0);
            acc[to][TIX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bl, acc[to][TIX], 0, 0, 0);
            acc[to][TIX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bh, acc[to][TIX], 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// staging-area tile `slot` <- one tile (transposing write, as dw_stage)
__device__ __forceinline__ void dw_stage_one(const f32x16& h, int slot, float* __restrict__ stage, int col, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[(slot * 32 + drow(r) + 4 * hi) * ST33 + col] = h[r];
    __builtin_amdgcn_wave_barrier();
}

// dX chain on the bf16 matrix cores (XR_MLP_BWD_DW=b2x): W^T sits in LDS pre-split into two bf16 parts, each in the
// [in neuron][hi][K-step][8] arrangement of the fp16 kernels; a gradient tile goes from the accumulator layout to the B
// operands of its two K-steps by a register-local 2-way split (the k-slot permutation is the accumulator layout's own row
// order, so nothing is transposed).  Per K-step and output tile: two 16-B operand reads and three MFMAs (gl*wh, gh*wl, gh*wh).
#ifndef B2X_PF
#define B2X_PF 1            // LDS operand double buffer of the split layers (one K-step ahead); 0 = none: backward 64 vs 63 us in the loop
#endif
struct B2Tile { bw8 p[2][2]; };                                  // [part][K-step]
__device__ __forceinline__ B2Tile to_b2(const f32x16& t) {
    B2Tile r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h0 = (__bf16)t[e], h1 = (__bf16)t[8 + e];
        r.p[0][0][e] = h0; r.p[1][0][e] = (__bf16)(t[e] - (float)h0);
        r.p[0][1][e] = h1; r.p[1][1][e] = (__bf16)(t[8 + e] - (float)h1);
    }
    return r;
}
template <int NH, int L>
__device__ __forceinline__ void store_layer_bt2(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                                __bf16* __restrict__ wb, int ps, bool first_layer_rot) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L), nso = prow / 16, rsb = h_rs(prow);
    __bf16* dstb = wb + H::b_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int x = threadIdx.x + i * MLP_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;             // LDS slot of global column c
        const float w = o < rows ? v[i] : 0.f;
        const __bf16 h = (__bf16)w;
        __bf16* d = dstb + m * rsb + hslot(o, nso);
        d[0] = h; d[ps] = (__bf16)(w - (float)h);
    }
}
template <int NH>
__device__ inline void load_weights_bt2(__bf16* __restrict__ wb, int ps, const float* __restrict__ w, bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / MLP_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / MLP_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / MLP_THREADS : 1];
    fetch_layer<NH, 0>(v0, w, false);              // source order; the slot rotation is applied on store
    fetch_layer<NH, 1>(v1, w, false);
    if constexpr (NH >= 2) fetch_layer<NH, 2>(v2, w, false);
    store_layer_bt2<NH, 0>(v0, wb, ps, first_layer_rot);
    store_layer_bt2<NH, 1>(v1, wb, ps, false);
    if constexpr (NH >= 2) store_layer_bt2<NH, 2>(v2, wb, ps, false);
}
// gin[TI] = W^T . g[TO]; only the first NSTEPS K-steps of g can be non-zero
template <int TO, int TI, int NSTEPS = 2 * TO>
__device__ __forceinline__ void layer_bwd_b2(const __bf16* __restrict__ wb, int ps, const B2Tile (&g)[TO], f32x16 (&gin)[TI], int col, int hi) {
    constexpr int NSO = 2 * TO, RSB = 2 * NSO * 8 + 8;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) gin[ti][r] = 0.f;
    const __bf16* wl = wb + col * RSB + hi * NSO * 8;
    bw8 a[B2X_PF + 1][TI][2];
    auto load = [&](int t, bw8 (&d)[TI][2]) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const __bf16* p = wl + ti * 32 * RSB + t * 8;
            d[ti][0] = *reinterpret_cast<const bw8*>(p);
            d[ti][1] = *reinterpret_cast<const bw8*>(p + ps);
        }
    };
    if (B2X_PF) load(0, a[0]);
#pragma unroll
    for (int t = 0; t < NSTEPS; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (B2X_PF) { if (t + 1 < NSTEPS) load(t + 1, a[(t + 1) & 1]); }
        else load(t, a[0]);
        const B2Tile& x = g[t >> 1];
        const int k = t & 1;
        const int cur = B2X_PF ? (t & 1) : 0;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            gin[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ti][1], x.p[0][k], gin[ti], 0, 0, 0);
            gin[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ti][0], x.p[1][k], gin[ti], 0, 0, 0);
            gin[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ti][0], x.p[0][k], gin[ti], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
// Forward recompute of the backward on the bf16 matrix cores (XR_MLP_BWD_DW=b2f): W in the forward arrangement
// [out neuron][hi][K-step][8], two bf16 parts.  Only what the backward's recompute reads is kept: the colour net's output
// layer is never recomputed, and the density net's output layer keeps its 16 real rows (lanes 16..31 read rows 0..15 again;
// their accumulator rows are not used) -- with W^T beside it the weights take 98.6 KiB and no fp32 copy is needed.
template <int NH>
struct F2Shape {
    __host__ __device__ static constexpr int rows(int l) { return l == NH ? (NH == 1 ? 16 : 0) : W_HID; }
    __host__ __device__ static constexpr int off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += rows(i) * h_rs(net_in_dim(i)); return o; }
    static constexpr int halves = off(NH + 1);
};
template <int NH, int L>
__device__ __forceinline__ void store_layer_fb2(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                                __bf16* __restrict__ wf, int psf, __bf16* __restrict__ wb, int psb, bool first_layer_rot) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    using F = F2Shape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L);
    constexpr int ns = K / 16, rs = h_rs(K), nso = prow / 16, rsb = h_rs(prow), frows = F::rows(L);
    __bf16* dstf = wf + F::off(L);
    __bf16* dstb = wb + H::b_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int x = threadIdx.x + i * MLP_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;             // LDS slot of global column c
        const float w = o < rows ? v[i] : 0.f;
        const __bf16 h = (__bf16)w, l = (__bf16)(w - (float)h);
        if (frows > 0 && o < frows) { __bf16* d = dstf + o * rs + hslot(m, ns); d[0] = h; d[psf] = l; }
        __bf16* d = dstb + m * rsb + hslot(o, nso);
        d[0] = h; d[psb] = l;
    }
}
template <int NH>
__device__ inline void load_weights_fb2(__bf16* __restrict__ wf, int psf, __bf16* __restrict__ wb, int psb, const float* __restrict__ w,
                                        bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / MLP_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / MLP_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / MLP_THREADS : 1];
    fetch_layer<NH, 0>(v0, w, false);
    fetch_layer<NH, 1>(v1, w, false);
    if constexpr (NH >= 2) fetch_layer<NH, 2>(v2, w, false);
    store_layer_fb2<NH, 0>(v0, wf, psf, wb, psb, first_layer_rot);
    store_layer_fb2<NH, 1>(v1, wf, psf, wb, psb, false);
    if constexpr (NH >= 2) store_layer_fb2<NH, 2>(v2, wf, psf, wb, psb, false);
}
// out[TO] = W . in[TI]; ROW16: the layer has 16 rows in LDS (lanes 16..31 repeat them)
template <int TI, int TO, bool ROW16 = false>
__device__ __forceinline__ void layer_fwd_b2(const __bf16* __restrict__ wf, int ps, const B2Tile (&in)[TI], f32x16 (&out)[TO], int col, int hi) {
    constexpr int NS = 2 * TI, RS = 2 * NS * 8 + 8;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[to][r] = 0.f;
    const __bf16* wl = wf + (ROW16 ? (col & 15) : col) * RS + hi * NS * 8;
    bw8 a[B2X_PF + 1][TO][2];
    auto load = [&](int t, bw8 (&d)[TO][2]) {
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const __bf16* p = wl + to * 32 * RS + t * 8;
            d[to][0] = *reinterpret_cast<const bw8*>(p);
            d[to][1] = *reinterpret_cast<const bw8*>(p + ps);
        }
    };
    if (B2X_PF) load(0, a[0]);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (B2X_PF) { if (t + 1 < NS) load(t + 1, a[(t + 1) & 1]); }
        else load(t, a[0]);
        const B2Tile& x = in[t >> 1];
        const int k = t & 1;
        const int cur = B2X_PF ? (t & 1) : 0;
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][to][1], x.p[0][k], out[to], 0, 0, 0);
            out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][to][0], x.p[1][k], out[to], 0, 0, 0);
            out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][to][0], x.p[0][k], out[to], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
// ---- fp32 operands as TWO fp16 parts (used by the forward kernels further down and by the backward's recompute, MODE 4) ----------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define H2_IN_SCALE 16.0f       // hash-grid features into the first layer: O(1e-4) at initialisation -> low parts of 2^-11 of that stay out of the
                                // deepest fp16 subnormals; features up to 4094 in magnitude are carried exactly, larger ones saturate (below)
// Range of the split.  fp16 ends at 65504: an operand above it would convert to inf and poison the row with NaN.  Every operand of
// this arithmetic is therefore SATURATED to +-65504 before it is split (the first layer's input after its scale, the colour net's
// input, every hidden activation -- there in the ReLU's own instruction, v_med3_f32(h, 0, 65504) -- and the weights), so the error of an
// out-of-range value is bounded, and the forward kernels COUNT the waves that saw one in the caller's range word
// (xr_set_mlp_range_word): the trainer reads it at the grid refresh's host read-back and the bench line prints it
// (`mlp_range_events`).  The backward's recompute saturates the same way (same ReLU decisions, same operands) and does not count.
#define H2_MAX 65504.0f
struct H2Tile { h8 p[2][2]; };               // [hi | lo part][K-step]
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2_sat(float x) { return __builtin_amdgcn_fmed3f(x, -H2_MAX, H2_MAX); }
// an INPUT tile (any sign): x * scale, saturated.  TRACK: mx = max(mx, |x * scale|) before the saturation
template <bool SAT = true, bool TRACK = false>
__device__ __forceinline__ H2Tile to_h2(const f32x16& t, float scale, float& mx) {
    H2Tile r;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 x = f32x2{t[8 * k + e], t[8 * k + e + 1]} * scale;
            if (TRACK) mx = fmaxf(fmaxf(mx, fabsf(x[0])), fabsf(x[1]));
            if (SAT) { x[0] = h2_sat(x[0]); x[1] = h2_sat(x[1]); }
            const f16x2 h = __builtin_convertvector(x, f16x2);
            const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
            r.p[0][k][e] = h[0]; r.p[0][k][e + 1] = h[1];
            r.p[1][k][e] = l[0]; r.p[1][k][e + 1] = l[1];
        }
    return r;
}
__device__ __forceinline__ H2Tile to_h2(const f32x16& t, float scale = 1.0f) { float d = 0.f; return to_h2<true, false>(t, scale, d); }
// a hidden activation tile already through h2_relu_sat: in range by construction
__device__ __forceinline__ H2Tile to_h2_act(const f32x16& t) { float d = 0.f; return to_h2<false, false>(t, 1.0f, d); }
// ReLU + the saturation in one instruction per value: t = med3(t * s, 0, 65504)
template <bool TRACK = false>
__device__ __forceinline__ void h2_relu_sat(f32x16& t, float s, float& mx) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 x = f32x2{t[r], t[r + 1]} * s;
        const float a = x[0], b = x[1];
        if (TRACK) mx = fmaxf(fmaxf(mx, a), b);
        t[r] = __builtin_amdgcn_fmed3f(a, 0.f, H2_MAX); t[r + 1] = __builtin_amdgcn_fmed3f(b, 0.f, H2_MAX);
    }
}
__device__ __forceinline__ void h2_relu_sat(f32x16& t, float s = 1.0f) { float d = 0.f; h2_relu_sat<false>(t, s, d); }
// one count per wave that saw an operand out of range (word: the caller's, nullable)
__device__ __forceinline__ void h2_range_report(float mx, uint32_t* __restrict__ word) {
    if (word != nullptr && __any(mx > H2_MAX) && (threadIdx.x & 63) == 0) atomicAdd(word, 1u);
}
// ROW16: the layer has 16 rows in LDS (lanes 16..31 repeat them; their accumulator rows are not used)
template <int TI, int TO, bool ROW16 = false>
__device__ __forceinline__ void layer_fwd_h2(const _Float16* __restrict__ wf, int ps, const H2Tile (&in)[TI], f32x16 (&out)[TO], int col, int hi) {
    constexpr int NS = 2 * TI, RS = 2 * NS * 8 + 8;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[to][r] = 0.f;
    const _Float16* wl = wf + (ROW16 ? (col & 15) : col) * RS + hi * NS * 8;
    h8 a[2][TO][2];
    auto load = [&](int t, h8 (&d)[TO][2]) {
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const _Float16* p = wl + to * 32 * RS + t * 8;
            d[to][0] = *reinterpret_cast<const h8*>(p);
            d[to][1] = *reinterpret_cast<const h8*>(p + ps);
        }
    };
    load(0, a[0]);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < NS) load(t + 1, a[(t + 1) & 1]);
        const H2Tile& x = in[t >> 1];
        const int k = t & 1, cur = t & 1;
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            out[to] = MFMA16(a[cur][to][1], x.p[0][k], out[to]);
            out[to] = MFMA16(a[cur][to][0], x.p[1][k], out[to]);
            out[to] = MFMA16(a[cur][to][0], x.p[0][k], out[to]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int NH, int L>
__device__ __forceinline__ void store_layer_fh2(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                                _Float16* __restrict__ wf, int psf, __bf16* __restrict__ wb, int psb, bool first_layer_rot) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    using F = F2Shape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L);
    constexpr int ns = K / 16, rs = h_rs(K), nso = prow / 16, rsb = h_rs(prow), frows = F::rows(L);
    _Float16* dstf = wf + F::off(L);
    __bf16* dstb = wb + H::b_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int x = threadIdx.x + i * MLP_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;             // LDS slot of global column c
        const float w = o < rows ? v[i] : 0.f;
        if (frows > 0 && o < frows) {
            const float ws = h2_sat(w);                                                 // (as store_layer_h2)
            const _Float16 h = (_Float16)ws;
            _Float16* d = dstf + o * rs + hslot(m, ns);
            d[0] = h; d[psf] = (_Float16)(ws - (float)h);
        }
        const __bf16 hb = (__bf16)w;
        __bf16* d = dstb + m * rsb + hslot(o, nso);
        d[0] = hb; d[psb] = (__bf16)(w - (float)hb);
    }
}
template <int NH>
__device__ inline void load_weights_fh2(_Float16* __restrict__ wf, int psf, __bf16* __restrict__ wb, int psb, const float* __restrict__ w,
                                        bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / MLP_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / MLP_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / MLP_THREADS : 1];
    fetch_layer<NH, 0>(v0, w, false);
    fetch_layer<NH, 1>(v1, w, false);
    if constexpr (NH >= 2) fetch_layer<NH, 2>(v2, w, false);
    store_layer_fh2<NH, 0>(v0, wf, psf, wb, psb, first_layer_rot);
    store_layer_fh2<NH, 1>(v1, wf, psf, wb, psb, false);
    if constexpr (NH >= 2) store_layer_fh2<NH, 2>(v2, wf, psf, wb, psb, false);
}
// block-level reduction target: LDS buffer in the GLOBAL (compact, [out][in]) parameter layout.
// The four waves of a workgroup take turns (caller: `for w: if (wave == w) dw_flush(.., first = (w == 0)); barrier`):
// the first one stores, the others read-add-write with plain LDS accesses.  NOT ds_add_f32: measured on MI355X
// (tools/lds_probe.hip) an LDS fp32 atomic add costs 81 ns per wave instruction -- ~3 cycles per lane, 27x an
// integer LDS atomic -- which made this flush (640 of them per workgroup) ~50 us of the 233-us kernel.
template <int TO, int TI>
__device__ __forceinline__ void dw_flush(const f32x16 (&acc)[TO][TI], float* __restrict__ dst, int rows, int K, bool rot,
                                         int col, int hi, bool first) {
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = to * 32 + drow(r) + 4 * hi;
                int m = ti * 32 + col;
                if (rot) m = (m + 31) & 31;
                if (o < rows) dst[o * K + m] = first ? acc[to][ti][r] : dst[o * K + m] + acc[to][ti][r];
            }
}

// SH-4 on d' = 2x-1 (tcnn SphericalHarmonics degree 4)
__device__ __forceinline__ void sh4_eval(float x, float y, float z, float* o) {
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

// ---- per-tile pieces ---------------------------------------------------------------------------
// Row r of the tile sits at enc_t + (drow(r) + 4 hi) * ld + s.  Written as a wave-uniform row base (scalar registers) plus ONE
// 32-bit per-lane byte offset, so that the 16 accesses are `global_load_dword v, v_off, s[base]`: as sixteen per-lane 64-bit
// addresses the compiler hoists 32 VGPRs of them out of the tile loop, and in the one-wave-per-SIMD backward those are the
// registers that end up in scratch (11 serialized reloads per tile, ~+7 us per launch).  Hosts check (5 ld) * 4 < 2^32.
__device__ __forceinline__ void load_enc_tile(const float* __restrict__ enc_t, uint32_t ld, uint32_t s, f32x16& xe, int hi) {
    const uint32_t voff = (4u * (uint32_t)hi * ld + s) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        xe[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(enc_t + (size_t)drow(r) * ld) + voff);
}
__device__ __forceinline__ void store_enc_tile(float* __restrict__ denc_t, uint32_t ld, uint32_t s, const f32x16& g, int hi, float scale = 1.0f) {
    const uint32_t voff = (4u * (uint32_t)hi * ld + s) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(denc_t + (size_t)drow(r) * ld) + voff) = scale == 1.0f ? g[r] : g[r] * scale;
}
// color-net input tile in slot space from the density output tile + SH of the view direction
__device__ __forceinline__ void build_color_in(const f32x16& dout, const float* __restrict__ dirs, uint32_t dir_stride,
                                               uint32_t s, float pad_value, f32x16& cin, int hi) {
    const float* d = dirs + (size_t)s * dir_stride;
    float sh[16];
    sh4_eval(d[0] * 2.f - 1.f, d[1] * 2.f - 1.f, d[2] * 2.f - 1.f, sh);
#pragma unroll
    for (int r = 0; r < 8; ++r) cin[r] = dout[r];
    if (hi == 0) cin[0] = pad_value;                       // slot 0 = pad (density row 0 is sigma, not an input)
#pragma unroll
    for (int r = 8; r < 16; ++r) {                         // slots 16..31 = SH[0..15]
        const int m0 = drow(r) - 16;                       // hi = 0 -> m0, hi = 1 -> m0 + 4
        cin[r] = hi ? sh[m0 + 4] : sh[m0];
    }
}

// ------------------------------------------------------------------ forward kernel
template <int NHD, int NHC, bool WITH_COLOR>
__global__ __launch_bounds__(MLP_THREADS, 3) void k_nerf_mlp_fwd(const float* __restrict__ enc_t, uint32_t ld,
                                                               const float* __restrict__ dirs, uint32_t dir_stride,
                                                               uint32_t n, const uint32_t* __restrict__ n_dev,
                                                               const uint32_t* __restrict__ rows,
                                                               const float* __restrict__ w_density,
                                                               const float* __restrict__ w_color, float pad_value,
                                                               float4* __restrict__ raw, const int32_t* __restrict__ splat_idx,
                                                               float* __restrict__ splat_grid) {
    if (n_dev) n = min(n, *n_dev);
    if (n == 0) return;
    using SD = NetShape<NHD>;
    using SC = NetShape<NHC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wd = lds;
    float* wc = lds + SD::lds_floats;
    load_weights<NHD>(wd, w_density, false);
    if (WITH_COLOR) load_weights<NHC>(wc, w_color, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const uint32_t n_tiles = (n + 31) / 32;
    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint32_t s = tile * 32 + col;
        const uint32_t sc = s < n ? s : n - 1;             // clamp loads of the ragged last tile
        f32x16 x[1], h[2], h2[2], dout[1];
        load_enc_tile(enc_t, ld, sc, x[0], hi);
        layer_fwd<1, 2>(wd + SD::lds_off(0), x, h, col, hi);
        relu_tile(h[0]); relu_tile(h[1]);
#pragma unroll
        for (int l = 1; l < NHD; ++l) {
            layer_fwd<2, 2>(wd + SD::lds_off(l), h, h2, col, hi);
            relu_tile(h2[0]); relu_tile(h2[1]);
            h[0] = h2[0]; h[1] = h2[1];
        }
        layer_fwd<2, 1>(wd + SD::lds_off(NHD), h, dout, col, hi);
        float4 o = make_float4(0.f, 0.f, 0.f, dout[0][0]);   // hi==0, reg 0 <-> row 0 = sigma
        if (WITH_COLOR) {
            f32x16 cin[1], cout[1];
            build_color_in(dout[0], dirs, dir_stride, rows ? rows[sc] : sc, pad_value, cin[0], hi);
            layer_fwd<1, 2>(wc + SC::lds_off(0), cin, h, col, hi);
            relu_tile(h[0]); relu_tile(h[1]);
#pragma unroll
            for (int l = 1; l < NHC; ++l) {
                layer_fwd<2, 2>(wc + SC::lds_off(l), h, h2, col, hi);
                relu_tile(h2[0]); relu_tile(h2[1]);
                h[0] = h2[0]; h[1] = h2[1];
            }
            layer_fwd<2, 1>(wc + SC::lds_off(NHC), h, cout, col, hi);
            o.x = cout[0][0]; o.y = cout[0][1]; o.z = cout[0][2];   // rows 0,1,2 in the hi==0 half
        }
        if (hi == 0 && s < n) {
            // density-only launches of the grid refresh: K8 (splat_grid_samples_nerf_max_nearest_neighbor.cu:7-28) in the epilogue -- the
            // sample's optical thickness at the smallest step goes straight into its cell's maximum, `raw` is not written
            if (!WITH_COLOR && splat_idx != nullptr) atomicMax((uint32_t*)&splat_grid[(uint32_t)splat_idx[s]], __float_as_uint(expf(o.w) * xr_min_step()));
            else raw[s] = o;
        }
    }
}

// ------------------------------------------------------------------ live rows of a backward pass
// A sample whose dL/d(raw) row is exactly (0,0,0,0) contributes exactly nothing to dW and gets an exactly-zero
// dL/d(encoding): behind an opaque surface the compositor's transmittance is exactly 0 in fp32, and in steady-state
// training that is MORE THAN HALF of the marched samples (tools/zero_grad_fraction.py: 0.53-0.55 at the bench
// workload).  The backward therefore runs on the stable compaction of the live rows: k_live_count counts them per
// 1024-row segment, k_live_fill writes the ordered row list (and the zero rows of denc_t), the MLP backward kernels
// take tile t, column c from live_rows[32 t + c].  Stable order + fixed partition: bit-reproducible run to run.
#define LIVE_SEG XR_LIVE_SEG
#define LIVE_THREADS 256
__device__ __forceinline__ bool row_live(const float4 d) { return d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f; }

__global__ __launch_bounds__(LIVE_THREADS) void k_live_count(const float4* __restrict__ draw, uint32_t n, const uint32_t* __restrict__ n_dev,
                                                            uint32_t* __restrict__ seg_count) {
    if (n_dev) n = min(n, *n_dev);
    __shared__ uint32_t ws[LIVE_THREADS / 64];
    const uint32_t r0 = blockIdx.x * LIVE_SEG + threadIdx.x * 4;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u)
        if (r0 + u < n) c += row_live(draw[r0 + u]) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_count[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(LIVE_THREADS) void k_live_fill(const float4* __restrict__ draw, uint32_t n, const uint32_t* __restrict__ n_dev,
                                                           const uint32_t* __restrict__ seg_count, uint32_t n_seg,
                                                           uint32_t* __restrict__ live_rows, uint32_t* __restrict__ n_live,
                                                           float* __restrict__ denc_t, uint32_t ld) {
    if (n_dev) n = min(n, *n_dev);
    __shared__ uint32_t ws[LIVE_THREADS / 64], wbase[LIVE_THREADS / 64 + 1];
    // rows before this segment: sum of the earlier segments' counts
    uint32_t before = 0;
    for (uint32_t s = threadIdx.x; s < blockIdx.x; s += LIVE_THREADS) before += seg_count[s];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = before;
    __syncthreads();
    before = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    __syncthreads();
    const uint32_t r0 = blockIdx.x * LIVE_SEG + threadIdx.x * 4;
    bool lv[4]; uint32_t c = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) { lv[u] = r0 + u < n && row_live(draw[r0 + u]); c += lv[u] ? 1u : 0u; }
    // exclusive scan of c over the block (wave scan, then the four wave totals)
    uint32_t inc = c;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int w = 0; w < LIVE_THREADS / 64; ++w) { wbase[w] = a; a += ws[w]; }
        wbase[LIVE_THREADS / 64] = a;
    }
    __syncthreads();
    uint32_t pos = before + wbase[threadIdx.x >> 6] + inc - c;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) if (lv[u]) live_rows[pos++] = r0 + u;
    if (blockIdx.x == n_seg - 1 && threadIdx.x == 0) {
        const uint32_t total = before + wbase[LIVE_THREADS / 64];
        n_live[0] = total;
        n_live[1] += total; n_live[2] += n;        // running totals (live rows, valid rows) since the caller last cleared them
    }
    // dead rows inside [0, n): their dL/d(encoding) is exactly zero (what the full backward would have written)
    if (!denc_t) return;
    if (c == 0 && r0 + 3 < n && (ld & 3) == 0 && (((uintptr_t)denc_t) & 15) == 0) {
#pragma unroll
        for (int f = 0; f < ENC_DIM; ++f) *reinterpret_cast<float4*>(denc_t + (size_t)f * ld + r0) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u)
            if (!lv[u] && r0 + u < n)
                for (int f = 0; f < ENC_DIM; ++f) denc_t[(size_t)f * ld + r0 + u] = 0.f;
    }
}

// ------------------------------------------------------------------ backward kernel
// Specialised for the reference topology family NHD = 1, NHC = 2 (density 32->64->16,
// color 32->64->64->16): every activation and all 12 dW accumulator tiles stay in registers.
// MODE 0: fp32 MFMA throughout; 1: dW products on the bf16 matrix cores (2-way split); 2 (default): the dX chain as well; 3: and the
// forward recompute (no fp32 copy of the weights).  LDS per workgroup, bytes: fp32 weights 50 176 (modes 0-2) | W in two bf16 parts
// 43 520 (mode 3); W^T in two bf16 parts 57 344 (modes 2-3); per-wave transposing stage 4 x 16 896 (modes 0-1) | 4 x 12 672 (modes 2-3):
// 117 760 / 158 208 / 151 552 of the 163 840.
template <bool LIVE, int MODE>
__global__ __launch_bounds__(MLP_THREADS, 1) void k_nerf_mlp_bwd_1_2(
    const float* __restrict__ enc_t, uint32_t ld, const float* __restrict__ dirs, uint32_t dir_stride, uint32_t n,
    const uint32_t* __restrict__ n_dev, const float* __restrict__ w_density, const float* __restrict__ w_color,
    float pad_value, const float4* __restrict__ draw, float* __restrict__ denc_t, float* __restrict__ partial /*[grid][GW]*/,
    const uint32_t* __restrict__ live_rows, const uint32_t* __restrict__ n_live) {
    if (n_dev) n = min(n, *n_dev);
    if (LIVE) n = *n_live;                                             // rows of the compacted space
    using SD = NetShape<1>;
    using SC = NetShape<2>;
    constexpr int GW = SD::glb_floats + SC::glb_floats;                // 3072 + 7168
    constexpr bool DWB = MODE >= 1, DXB = MODE >= 2, FWB = MODE >= 3, FWH = MODE == 4;      // FWH: the recompute on two fp16 parts instead of two bf16 parts
    using HD = HShape<1>;
    using HC = HShape<2>;
    using FD = F2Shape<1>;
    using FC = F2Shape<2>;
    constexpr int PD = HD::b_halves, PC = HC::b_halves;                // halves per part of W^T
    constexpr int PFD = FD::halves, PFC = FC::halves;                  // halves per part of W (FWB)
    constexpr int STAGE = (DXB ? 3 : 4) * 32 * ST33;                   // floats per wave
    constexpr int W32 = FWB ? 0 : SD::lds_floats + SC::lds_floats;     // fp32 weights (none when the recompute is on bf16 too)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wd = lds;
    float* wc = wd + SD::lds_floats;
    __bf16* wfd = reinterpret_cast<__bf16*>(lds + W32);                // FWB: W of both networks, two bf16 parts each
    __bf16* wfc = wfd + (FWB ? 2 * PFD : 0);
    __bf16* wbd = wfc + (FWB ? 2 * PFC : 0);                           // DXB: W^T of both networks, two bf16 parts each
    __bf16* wbc = wbd + 2 * PD;
    float* stage_all = DXB ? reinterpret_cast<float*>(wbc + 2 * PC) : lds + W32;             // MLP_WAVES * STAGE floats
    constexpr int LDS_FLOATS = W32 + (FWB ? PFD + PFC : 0) + (DXB ? PD + PC : 0) + MLP_WAVES * STAGE;
    static_assert(PD % 8 == 0 && PC % 8 == 0 && PFD % 8 == 0 && PFC % 8 == 0 && (SD::lds_floats + SC::lds_floats) % 4 == 0,
                  "16-byte alignment of the operand reads");
    if constexpr (FWH) {
        load_weights_fh2<1>(reinterpret_cast<_Float16*>(wfd), PFD, wbd, PD, w_density, false);
        load_weights_fh2<2>(reinterpret_cast<_Float16*>(wfc), PFC, wbc, PC, w_color, true);
    } else if constexpr (FWB) {
        load_weights_fb2<1>(wfd, PFD, wbd, PD, w_density, false);
        load_weights_fb2<2>(wfc, PFC, wbc, PC, w_color, true);
    } else {
        load_weights<1>(wd, w_density, false);
        load_weights<2>(wc, w_color, true);
        if constexpr (DXB) {
            load_weights_bt2<1>(wbd, PD, w_density, false);
            load_weights_bt2<2>(wbc, PC, w_color, true);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    float* stage = stage_all + wave * STAGE;

    f32x16 a_d0[2][1], a_d1[1][2], a_c0[2][1], a_c1[2][2], a_c2[1][2];   // dW accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a_d0[0][0][r] = a_d0[1][0][r] = 0.f; a_d1[0][0][r] = a_d1[0][1][r] = 0.f;
        a_c0[0][0][r] = a_c0[1][0][r] = 0.f;
        a_c1[0][0][r] = a_c1[0][1][r] = a_c1[1][0][r] = a_c1[1][1][r] = 0.f;
        a_c2[0][0][r] = a_c2[0][1][r] = 0.f;
    }

    const uint32_t n_tiles = (n + 31) / 32;
    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint32_t s0 = tile * 32 + col;
        const bool live = s0 < n;
        const uint32_t s = LIVE ? live_rows[live ? s0 : n - 1] : s0;
        const uint32_t sc = LIVE ? s : (live ? s : n - 1);
        // ---- recompute forward, keep activations
        f32x16 xe[1], hd[2], dout[1], cin[1], hc1[2], hc2[2];
        load_enc_tile(enc_t, ld, sc, xe[0], hi);
        if constexpr (FWH) {
            // the forward's own arithmetic (k_nerf_mlp_fwd_h2: same split, same products in the same order): its ReLU decisions bit for bit
            const _Float16* hfd = reinterpret_cast<const _Float16*>(wfd);
            const _Float16* hfc = reinterpret_cast<const _Float16*>(wfc);
            // (and its saturation: h2_relu_sat / to_h2)
            { const H2Tile xb[1] = {to_h2(xe[0], H2_IN_SCALE)}; layer_fwd_h2<1, 2>(hfd + FD::off(0), PFD, xb, hd, col, hi); }
            h2_relu_sat(hd[0], 1.0f / H2_IN_SCALE); h2_relu_sat(hd[1], 1.0f / H2_IN_SCALE);     // relu + the input scale taken out again (exact)
            { const H2Tile hb[2] = {to_h2_act(hd[0]), to_h2_act(hd[1])}; layer_fwd_h2<2, 1, true>(hfd + FD::off(1), PFD, hb, dout, col, hi); }
            build_color_in(dout[0], dirs, dir_stride, sc, pad_value, cin[0], hi);
            { const H2Tile cb[1] = {to_h2(cin[0])}; layer_fwd_h2<1, 2>(hfc + FC::off(0), PFC, cb, hc1, col, hi); }
            h2_relu_sat(hc1[0]); h2_relu_sat(hc1[1]);
            { const H2Tile hb[2] = {to_h2_act(hc1[0]), to_h2_act(hc1[1])}; layer_fwd_h2<2, 2>(hfc + FC::off(1), PFC, hb, hc2, col, hi); }
            h2_relu_sat(hc2[0]); h2_relu_sat(hc2[1]);
        } else if constexpr (FWB) {
            { const B2Tile xb[1] = {to_b2(xe[0])}; layer_fwd_b2<1, 2>(wfd + FD::off(0), PFD, xb, hd, col, hi); }
            relu_tile(hd[0]); relu_tile(hd[1]);
            { const B2Tile hb[2] = {to_b2(hd[0]), to_b2(hd[1])}; layer_fwd_b2<2, 1, true>(wfd + FD::off(1), PFD, hb, dout, col, hi); }
            build_color_in(dout[0], dirs, dir_stride, sc, pad_value, cin[0], hi);
            { const B2Tile cb[1] = {to_b2(cin[0])}; layer_fwd_b2<1, 2>(wfc + FC::off(0), PFC, cb, hc1, col, hi); }
            relu_tile(hc1[0]); relu_tile(hc1[1]);
            { const B2Tile hb[2] = {to_b2(hc1[0]), to_b2(hc1[1])}; layer_fwd_b2<2, 2>(wfc + FC::off(1), PFC, hb, hc2, col, hi); }
            relu_tile(hc2[0]); relu_tile(hc2[1]);
        } else {
            layer_fwd<1, 2, false>(wd + SD::lds_off(0), xe, hd, col, hi);
            relu_tile(hd[0]); relu_tile(hd[1]);
            layer_fwd<2, 1, false>(wd + SD::lds_off(1), hd, dout, col, hi);
            build_color_in(dout[0], dirs, dir_stride, sc, pad_value, cin[0], hi);
            layer_fwd<1, 2, false>(wc + SC::lds_off(0), cin, hc1, col, hi);
            relu_tile(hc1[0]); relu_tile(hc1[1]);
            layer_fwd<2, 2, false>(wc + SC::lds_off(1), hc1, hc2, col, hi);
            relu_tile(hc2[0]); relu_tile(hc2[1]);
        }
        // ---- output gradients: rows 0..2 of the color output tile = dL/d(rgb raw), all else 0.
        // dead (ragged) samples get a zero gradient so they contribute nothing to dW.
        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && hi == 0) dr = draw[s];
        f32x16 g1[1], g2[2], g2b[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[0][r] = 0.f;
        g1[0][0] = dr.x; g1[0][1] = dr.y; g1[0][2] = dr.z;             // hi==1 lanes hold zeros
        // color output layer
        dw_stage<1, 2>(g1, hc2, stage, col, hi);
        if constexpr (DXB) { const B2Tile gb[1] = {to_b2(g1[0])}; layer_bwd_b2<1, 2, 1>(wbc + HC::b_off(2), PC, gb, g2, col, hi); }
        else layer_bwd<1, 2, false, 3>(wc + SC::lds_off(2), g1, g2, col, hi);   // rows 0..2 (rgb) only
        dw_product<DWB, 1, 2>(a_c2, stage, col, hi);
        relu_mask(g2[0], hc2[0]); relu_mask(g2[1], hc2[1]);
        // color hidden layer 2
        if constexpr (DXB) {
            // 3-tile staging area: both G tiles and one H tile at a time
            const f32x16 h0[1] = {hc1[0]};
            dw_stage<2, 1>(g2, h0, stage, col, hi);
            const B2Tile gb[2] = {to_b2(g2[0]), to_b2(g2[1])};
            layer_bwd_b2<2, 2>(wbc + HC::b_off(1), PC, gb, g2b, col, hi);
            dw_mfma_b2_col<2, 2, 0>(a_c1, stage, col, hi);
            dw_stage_one(hc1[1], 2, stage, col, hi);
            dw_mfma_b2_col<2, 2, 1>(a_c1, stage, col, hi);
        } else {
            dw_stage<2, 2>(g2, hc1, stage, col, hi);
            layer_bwd<2, 2, false>(wc + SC::lds_off(1), g2, g2b, col, hi);
            dw_product<DWB, 2, 2>(a_c1, stage, col, hi);
        }
        relu_mask(g2b[0], hc1[0]); relu_mask(g2b[1], hc1[1]);
        // color input layer
        dw_stage<2, 1>(g2b, cin, stage, col, hi);
        if constexpr (DXB) { const B2Tile gb[2] = {to_b2(g2b[0]), to_b2(g2b[1])}; layer_bwd_b2<2, 1>(wbc + HC::b_off(0), PC, gb, g1, col, hi); }
        else layer_bwd<2, 1, false>(wc + SC::lds_off(0), g2b, g1, col, hi);       // g1 = dL/d(color input slots)
        dw_product<DWB, 2, 1>(a_c0, stage, col, hi);
        // slots 1..15 are density-output rows 1..15; row 0 takes dL/d(sigma raw); rows >= 16 are padding
#pragma unroll
        for (int r = 8; r < 16; ++r) g1[0][r] = 0.f;
        if (hi == 0) g1[0][0] = dr.w;
        // density output layer
        if constexpr (DXB) {
            // the encoded features are not kept across the tile (16 registers this kernel does not have: they went to scratch,
            // nine serialized reloads per tile); they are fetched again here, two layers before their use -- L2 hits
            uint32_t sc2 = sc;
            asm volatile("" : "+v"(sc2));                                  // a second load, not the first one kept alive
            load_enc_tile(enc_t, ld, sc2, xe[0], hi);
        }
        dw_stage<1, 2>(g1, hd, stage, col, hi);
        if constexpr (DXB) { const B2Tile gb[1] = {to_b2(g1[0])}; layer_bwd_b2<1, 2, 1>(wbd + HD::b_off(1), PD, gb, g2, col, hi); }
        else layer_bwd<1, 2, false, 8>(wd + SD::lds_off(1), g1, g2, col, hi);   // 16 real output neurons
        dw_product<DWB, 1, 2>(a_d1, stage, col, hi);
        relu_mask(g2[0], hd[0]); relu_mask(g2[1], hd[1]);
        // density input layer
        dw_stage<2, 1>(g2, xe, stage, col, hi);
        if constexpr (DXB) { const B2Tile gb[2] = {to_b2(g2[0]), to_b2(g2[1])}; layer_bwd_b2<2, 1>(wbd + HD::b_off(0), PD, gb, g1, col, hi); }
        else layer_bwd<2, 1, false>(wd + SD::lds_off(0), g2, g1, col, hi);        // g1 = dL/d(encoded features)
        dw_product<DWB, 2, 1>(a_d0, stage, col, hi);
        if (live) {
            store_enc_tile(denc_t, ld, s, g1[0], hi);
        }
    }
    // ---- block reduction of dW through LDS (compact global layout), then one partial per block
    // Every wave is past its last tile: weights and staging are dead, the whole dynamic LDS holds TWO copies of the gradient.
    // Waves 0 / 1 store into copy A / B side by side, waves 2 / 3 add, the write-out sums the copies: two turns instead of
    // four (fixed order (w0 + w2) + (w1 + w3)).
    __syncthreads();
    static_assert(2 * GW <= LDS_FLOATS, "two reduction copies must fit the launch's LDS");
    float* redA = lds;
    for (int ph = 0; ph < 2; ++ph) {
        if ((wave >> 1) == ph) {
            float* red = redA + (wave & 1) * GW;
            float* redc = red + SD::glb_floats;
            dw_flush<2, 1>(a_d0, red + SD::glb_off(0), 64, 32, false, col, hi, ph == 0);
            dw_flush<1, 2>(a_d1, red + SD::glb_off(1), 16, 64, false, col, hi, ph == 0);
            dw_flush<2, 1>(a_c0, redc + SC::glb_off(0), 64, 32, true, col, hi, ph == 0);
            dw_flush<2, 2>(a_c1, redc + SC::glb_off(1), 64, 64, false, col, hi, ph == 0);
            dw_flush<1, 2>(a_c2, redc + SC::glb_off(2), 16, 64, false, col, hi, ph == 0);
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * GW;
    for (int e = threadIdx.x; e < GW; e += MLP_THREADS) out[e] = redA[e] + redA[GW + e];
}

// grad[j] += sum_b partial[b][j], fixed order: 64 columns x 4 row groups per block, each thread sums
// every 4th partial, then the 4 group sums are added in a fixed order through LDS
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partial, uint32_t nb, uint32_t gw,
                                                          uint32_t split, float* __restrict__ g_density,
                                                          float* __restrict__ g_color, int overwrite = 0) {
    __shared__ float red[4][64];
    XrAuxWork w;                                   // (xr_aux.h: the one definition of this sum; the training step runs it inside
    w.partial = partial; w.nb = nb; w.gw = gw; w.split = split; w.g0 = g_density; w.g1 = g_color;     //  the table scatter's binning launch)
    w.overwrite = overwrite; w.adam = 0;
    xr_aux_reduce_block<256>(w, blockIdx.x, red);
}

// the same fixed-order sum for any [n_partials][stride] buffer (xr_sum_partials): columns j < n
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partial, uint32_t nb, uint32_t stride, uint32_t n, float* __restrict__ out) {
    __shared__ float red[4][64];
    const uint32_t c = threadIdx.x & 63u, rg = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < n) {
        uint32_t b = rg;
        for (; b + 4 * 15 < nb; b += 4 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(b + 4 * u) * stride + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; b < nb; b += 4) s += partial[(size_t)b * stride + j];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && j < n) out[j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------ generic single network
// tcnn.Network(FullyFusedMLP) used on its own: x [n, n_in] (arbitrary row / column strides, n_in <= 32,
// missing input columns are filled with pad_value like tcnn's Identity-encoded input) -> y [n, 16] row-major.
// Same building blocks as the fused NeRF kernels; this is the compatibility surface, not the hot path.
__device__ __forceinline__ void load_x_tile(const float* __restrict__ x, long rs, long cs, uint32_t s, int n_in, float pad,
                                            f32x16& t, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = drow(r) + 4 * hi;
        t[r] = c < n_in ? x[(long)s * rs + (long)c * cs] : pad;
    }
}
template <int NH>
__global__ __launch_bounds__(MLP_THREADS, 2) void k_mlp_fwd(const float* __restrict__ x, long rs, long cs, int n_in, float pad,
                                                             uint32_t n, const float* __restrict__ w, float* __restrict__ y) {
    using S = NetShape<NH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    load_weights<NH>(lds, w, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const uint32_t n_tiles = (n + 31) / 32;
    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint32_t s = tile * 32 + col, sc = s < n ? s : n - 1;
        f32x16 xin[1], h[2], h2[2], out[1];
        load_x_tile(x, rs, cs, sc, n_in, pad, xin[0], hi);
        layer_fwd<1, 2>(lds + S::lds_off(0), xin, h, col, hi);
        relu_tile(h[0]); relu_tile(h[1]);
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            layer_fwd<2, 2>(lds + S::lds_off(l), h, h2, col, hi);
            relu_tile(h2[0]); relu_tile(h2[1]);
            h[0] = h2[0]; h[1] = h2[1];
        }
        layer_fwd<2, 1>(lds + S::lds_off(NH), h, out, col, hi);
        if (s < n) {
#pragma unroll
            for (int r = 0; r < 8; ++r) y[(size_t)s * 16 + drow(r) + 4 * hi] = out[0][r];
        }
    }
}
// backward: recompute, dX chain, dW (block partials like the fused kernel).  NH in {1, 2}.
template <int NH>
__global__ __launch_bounds__(MLP_THREADS, 1) void k_mlp_bwd(const float* __restrict__ x, long rs, long cs, int n_in, float pad,
                                                             uint32_t n, const float* __restrict__ w, const float* __restrict__ dy,
                                                             float* __restrict__ dx /*[n, n_in] row-major or null*/,
                                                             float* __restrict__ partial) {
    using S = NetShape<NH>;
    constexpr int GW = S::glb_floats;
    constexpr int STAGE = 4 * 32 * ST33;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage_all = lds + S::lds_floats;
    load_weights<NH>(lds, w, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    float* stage = stage_all + wave * STAGE;
    f32x16 a_in[2][1], a_hid[2][2], a_out[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a_in[0][0][r] = a_in[1][0][r] = 0.f; a_out[0][0][r] = a_out[0][1][r] = 0.f;
        a_hid[0][0][r] = a_hid[0][1][r] = a_hid[1][0][r] = a_hid[1][1][r] = 0.f;
    }
    const uint32_t n_tiles = (n + 31) / 32;
    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += gridDim.x * MLP_WAVES) {
        const uint32_t s = tile * 32 + col;
        const bool live = s < n;
        const uint32_t sc = live ? s : n - 1;
        f32x16 xin[1], h1[2], h2[2], g1[1], ga[2], gb[2];
        load_x_tile(x, rs, cs, sc, n_in, pad, xin[0], hi);
        layer_fwd<1, 2, false>(lds + S::lds_off(0), xin, h1, col, hi);
        relu_tile(h1[0]); relu_tile(h1[1]);
        if (NH == 2) {
            layer_fwd<2, 2, false>(lds + S::lds_off(1), h1, h2, col, hi);
            relu_tile(h2[0]); relu_tile(h2[1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[0][r] = 0.f;
        if (live) {
#pragma unroll
            for (int r = 0; r < 8; ++r) g1[0][r] = dy[(size_t)s * 16 + drow(r) + 4 * hi];
        }
        if (NH == 2) {
            dw_accumulate<1, 2, false>(a_out, g1, h2, stage, col, hi);
            layer_bwd<1, 2, false, 8>(lds + S::lds_off(2), g1, ga, col, hi);
            relu_mask(ga[0], h2[0]); relu_mask(ga[1], h2[1]);
            dw_accumulate<2, 2, false>(a_hid, ga, h1, stage, col, hi);
            layer_bwd<2, 2, false>(lds + S::lds_off(1), ga, gb, col, hi);
            relu_mask(gb[0], h1[0]); relu_mask(gb[1], h1[1]);
        } else {
            dw_accumulate<1, 2, false>(a_out, g1, h1, stage, col, hi);
            layer_bwd<1, 2, false, 8>(lds + S::lds_off(1), g1, gb, col, hi);
            relu_mask(gb[0], h1[0]); relu_mask(gb[1], h1[1]);
        }
        dw_accumulate<2, 1, false>(a_in, gb, xin, stage, col, hi);
        if (dx) {
            layer_bwd<2, 1, false>(lds + S::lds_off(0), gb, g1, col, hi);
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = drow(r) + 4 * hi;
                    if (c < n_in) dx[(size_t)s * n_in + c] = g1[0][r];
                }
            }
        }
    }
    __syncthreads();
    float* red = stage_all;
    for (int e = threadIdx.x; e < GW; e += MLP_THREADS) red[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < MLP_WAVES; ++w) {                              // wave-serial plain adds into the zero-filled buffer
        if (wave == w) {
            dw_flush<2, 1>(a_in, red + S::glb_off(0), 64, 32, false, col, hi, false);
            if (NH == 2) dw_flush<2, 2>(a_hid, red + S::glb_off(1), 64, 64, false, col, hi, false);
            dw_flush<1, 2>(a_out, red + S::glb_off(NH), 16, 64, false, col, hi, false);
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * GW;
    for (int e = threadIdx.x; e < GW; e += MLP_THREADS) out[e] = red[e];
}

// ================================================================== reference-precision mode (fp16 MFMA)
// tiny-cuda-nn, which the reference calls, runs FullyFusedMLP in fp16 with fp32 accumulation and hands fp16 outputs back
// (hashnerf_mlp.py:76-77 casts them to fp32).  This is the same arithmetic on v_mfma_f32_32x32x16_f16 (16x the fp32
// MFMA rate): weights and activations rounded to fp16, products summed in fp32, outputs taken from the fp32 accumulators.
// Selected by xrnerf_amd.ops.set_precision('f16') / XRNERF_MLP_PRECISION=f16; the fp32 kernels above stay the parity
// mode and the default.  Topology (1, 2) only (configs/instant_ngp).
//
// Same transposed scheme as above: a wave owns 32 samples, tiles are 32 neurons x 32 samples in the C/D layout
//     lane l: column (sample) = l & 31, row (neuron) = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
// One 32x32x16 step contracts over 16 k's: lane half hi supplies 8 of them to A (8 halves of weight row l & 31) and 8 to
// B (8 halves of sample column l & 31).  A sum over k is order-free, so step t takes as its k's exactly the neurons the
// lane ALREADY holds in registers 8 (t & 1) .. 8 (t & 1) + 7 of tile t >> 1:
//     k-slot (t, hi, e)  <->  neuron hrow(t, hi, e) = 32 (t >> 1) + 16 (t & 1) + 8 (e >> 2) + (e & 3) + 4 hi
// -- activations go from accumulators to the next layer's B operand by a register-local fp32 -> fp16 conversion, and the
// weights are stored in LDS pre-permuted ([row][hi][t][8 halves], row stride + 16 B: conflict-free ds_read_b128).  The
// backward keeps a second, transposed arrangement for the dX chain.  dW contracts over the tile's 32 samples through a
// per-wave fp16 staging tile [row][32 samples (+8)].  Gradients are scaled by 128 before the fp16 conversion (tcnn's
// loss scale) and unscaled in fp32.
#define H16_LOSS_SCALE 128.0f

template <int NH, int L>
__device__ __forceinline__ void store_layer_h(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / MLP_THREADS],
                                              _Float16* __restrict__ wf, _Float16* __restrict__ wb, bool first_layer_rot) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L);
    constexpr int ns = K / 16, rs = h_rs(K), nso = prow / 16, rsb = h_rs(prow);
    _Float16* dst = wf + H::f_off(L);
    _Float16* dstb = wb ? wb + H::b_off(L) : nullptr;
#pragma unroll
    for (int i = 0; i < prow * K / MLP_THREADS; ++i) {
        const int x = threadIdx.x + i * MLP_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;             // LDS slot of global column c
        const _Float16 h = (_Float16)(o < rows ? v[i] : 0.f);
        dst[o * rs + hslot(m, ns)] = h;
        if (dstb) dstb[m * rsb + hslot(o, nso)] = h;
    }
}
// same two-phase scheme as load_weights (all global loads in flight, then the converting LDS stores); the fetch reads
// the SOURCE elements in order (coalesced), i.e. fetch_layer without the slot rotation
template <int NH>
__device__ inline void load_weights_h(_Float16* __restrict__ wf, _Float16* __restrict__ wb, const float* __restrict__ w, bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "the fp16 mode is built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / MLP_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / MLP_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / MLP_THREADS : 1];
    fetch_layer<NH, 0>(v0, w, false);
    fetch_layer<NH, 1>(v1, w, false);
    if constexpr (NH >= 2) fetch_layer<NH, 2>(v2, w, false);
    store_layer_h<NH, 0>(v0, wf, wb, first_layer_rot);
    store_layer_h<NH, 1>(v1, wf, wb, false);
    if constexpr (NH >= 2) store_layer_h<NH, 2>(v2, wf, wb, false);
}

struct HTile { h8 p[2]; };                       // a 32 x 32 tile as the two fp16 B operands of its two K-steps
__device__ __forceinline__ HTile to_h(const f32x16& t) {
    HTile r;
#pragma unroll
    for (int e = 0; e < 8; ++e) { r.p[0][e] = (_Float16)t[e]; r.p[1][e] = (_Float16)t[8 + e]; }
    return r;
}
// out[TO] = W . in[TI].  SB: explicit operand prefetch (distance HPF K-steps) fenced by scheduling barriers -- the
// backward kernel runs one wave per SIMD at the register limit, and a scheduler free to hoist every ds_read_b128 of a
// layer (4 VGPRs each) to the top pushes its accumulators into scratch.
#ifndef HPF
#define HPF 2
#endif
#ifndef HSB
#define HSB false
#endif
template <int TI, int TO, bool SB = false>
__device__ __forceinline__ void layer_fwd_h(const _Float16* __restrict__ wf, const HTile (&in)[TI], f32x16 (&out)[TO], int col, int hi) {
    constexpr int NS = 2 * TI, RS = 2 * NS * 8 + 8;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[to][r] = 0.f;
    if (!SB) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const h8 a = *reinterpret_cast<const h8*>(wf + (to * 32 + col) * RS + (hi * NS + t) * 8);
                out[to] = MFMA16(a, in[t >> 1].p[t & 1], out[to]);
            }
        }
    } else {
        const _Float16* wl = wf + col * RS + hi * NS * 8;
        h8 a[HPF + 1][TO];
#pragma unroll
        for (int p = 0; p < HPF && p < NS; ++p)
#pragma unroll
            for (int to = 0; to < TO; ++to) a[p][to] = *reinterpret_cast<const h8*>(wl + to * 32 * RS + p * 8);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + HPF < NS) {
#pragma unroll
                for (int to = 0; to < TO; ++to) a[(t + HPF) % (HPF + 1)][to] = *reinterpret_cast<const h8*>(wl + to * 32 * RS + (t + HPF) * 8);
            }
#pragma unroll
            for (int to = 0; to < TO; ++to) out[to] = MFMA16(a[t % (HPF + 1)][to], in[t >> 1].p[t & 1], out[to]);
        }
    }
}
// gin[TI] = W^T . g[TO]; only the first NSTEPS K-steps of g can be non-zero
template <int TO, int TI, int NSTEPS = 2 * TO, bool SB = false>
__device__ __forceinline__ void layer_bwd_h(const _Float16* __restrict__ wb, const HTile (&g)[TO], f32x16 (&gin)[TI], int col, int hi) {
    constexpr int NSO = 2 * TO, RSB = 2 * NSO * 8 + 8;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) gin[ti][r] = 0.f;
    if (!SB) {
#pragma unroll
        for (int t = 0; t < NSTEPS; ++t) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                const h8 a = *reinterpret_cast<const h8*>(wb + (ti * 32 + col) * RSB + (hi * NSO + t) * 8);
                gin[ti] = MFMA16(a, g[t >> 1].p[t & 1], gin[ti]);
            }
        }
    } else {
        const _Float16* wl = wb + col * RSB + hi * NSO * 8;
        h8 a[HPF + 1][TI];
#pragma unroll
        for (int p = 0; p < HPF && p < NSTEPS; ++p)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) a[p][ti] = *reinterpret_cast<const h8*>(wl + ti * 32 * RSB + p * 8);
#pragma unroll
        for (int t = 0; t < NSTEPS; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + HPF < NSTEPS) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) a[(t + HPF) % (HPF + 1)][ti] = *reinterpret_cast<const h8*>(wl + ti * 32 * RSB + (t + HPF) * 8);
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) gin[ti] = MFMA16(a[t % (HPF + 1)][ti], g[t >> 1].p[t & 1], gin[ti]);
        }
    }
}
#define HST 40                                   // staging row: 32 samples + 8 halves of padding (80 B)
template <int TO, int TI>
__device__ __forceinline__ void dw_stage_h(const HTile (&g)[TO], const HTile (&h)[TI], _Float16* __restrict__ stage, int col, int hi) {
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(to * 32 + drow(r) + 4 * hi) * HST + col] = g[to].p[r >> 3][r & 7];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((TO + ti) * 32 + drow(r) + 4 * hi) * HST + col] = h[ti].p[r >> 3][r & 7];
    __builtin_amdgcn_wave_barrier();
}
template <int TO, int TI>
__device__ __forceinline__ void dw_mfma_h(f32x16 (&acc)[TO][TI], const _Float16* __restrict__ stage, int col, int hi) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        h8 a[TO], b[TI];
#pragma unroll
        for (int to = 0; to < TO; ++to) a[to] = *reinterpret_cast<const h8*>(stage + (to * 32 + col) * HST + 16 * t + 8 * hi);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) b[ti] = *reinterpret_cast<const h8*>(stage + ((TO + ti) * 32 + col) * HST + 16 * t + 8 * hi);
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) acc[to][ti] = MFMA16(a[to], b[ti], acc[to][ti]);
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void scale_tile(f32x16& t, float s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] *= s;
}

template <bool WITH_COLOR>
__global__ __launch_bounds__(MLP_THREADS, 2) void k_nerf_mlp_fwd_h(const float* __restrict__ enc_t, uint32_t ld,
                                                                    const float* __restrict__ dirs, uint32_t dir_stride,
                                                                    uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                    const uint32_t* __restrict__ rows,
                                                                    const float* __restrict__ w_density,
                                                                    const float* __restrict__ w_color, float pad_value,
                                                                    float4* __restrict__ raw, const int32_t* __restrict__ splat_idx,
                                                                    float* __restrict__ splat_grid) {
    if (n_dev) n = min(n, *n_dev);
    if (n == 0) return;
    using HD = HShape<1>;
    using HC = HShape<2>;
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    _Float16* wd = ldsh;
    _Float16* wc = ldsh + HD::f_halves;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const uint32_t n_tiles = (n + 31) / 32, stride = gridDim.x * MLP_WAVES;
    uint32_t tile = blockIdx.x * MLP_WAVES + wave;
    // the first tile's inputs are in flight while the workgroup converts the weights
    f32x16 x;
    float d3[3] = {0.f, 0.f, 0.f};
    auto fetch = [&](uint32_t tl, f32x16& xe, float (&dd)[3]) {
        const uint32_t s = tl * 32 + col, sc = s < n ? s : n - 1;
        load_enc_tile(enc_t, ld, sc, xe, hi);
        if (WITH_COLOR) {
            const float* d = dirs + (size_t)(rows ? rows[sc] : sc) * dir_stride;
            dd[0] = d[0]; dd[1] = d[1]; dd[2] = d[2];
        }
    };
    if (tile < n_tiles) fetch(tile, x, d3);
    load_weights_h<1>(wd, nullptr, w_density, false);
    if (WITH_COLOR) load_weights_h<2>(wc, nullptr, w_color, true);
    __syncthreads();
    for (; tile < n_tiles; tile += stride) {
        const uint32_t s = tile * 32 + col;
        HTile xin[1] = {to_h(x)};
        const float dx = d3[0], dy = d3[1], dz = d3[2];
        if (tile + stride < n_tiles) fetch(tile + stride, x, d3);           // next tile's loads under this tile's MFMAs
        f32x16 h[2], dout[1];
        layer_fwd_h<1, 2>(wd + HD::f_off(0), xin, h, col, hi);
        relu_tile(h[0]); relu_tile(h[1]);
        HTile hh[2] = {to_h(h[0]), to_h(h[1])};
        layer_fwd_h<2, 1>(wd + HD::f_off(1), hh, dout, col, hi);
        float4 o = make_float4(0.f, 0.f, 0.f, dout[0][0]);
        if (WITH_COLOR) {
            f32x16 cin, cout[1];
            const float dd[3] = {dx, dy, dz};
            build_color_in(dout[0], dd, 3, 0, pad_value, cin, hi);
            HTile ci[1] = {to_h(cin)};
            layer_fwd_h<1, 2>(wc + HC::f_off(0), ci, h, col, hi);
            relu_tile(h[0]); relu_tile(h[1]);
            hh[0] = to_h(h[0]); hh[1] = to_h(h[1]);
            layer_fwd_h<2, 2>(wc + HC::f_off(1), hh, h, col, hi);
            relu_tile(h[0]); relu_tile(h[1]);
            hh[0] = to_h(h[0]); hh[1] = to_h(h[1]);
            layer_fwd_h<2, 1>(wc + HC::f_off(2), hh, cout, col, hi);
            o.x = cout[0][0]; o.y = cout[0][1]; o.z = cout[0][2];
        }
        if (hi == 0 && s < n) {
            // density-only launches of the grid refresh: K8 (splat_grid_samples_nerf_max_nearest_neighbor.cu:7-28) in the epilogue -- the
            // sample's optical thickness at the smallest step goes straight into its cell's maximum, `raw` is not written
            if (!WITH_COLOR && splat_idx != nullptr) atomicMax((uint32_t*)&splat_grid[(uint32_t)splat_idx[s]], __float_as_uint(expf(o.w) * xr_min_step()));
            else raw[s] = o;
        }
    }
}

template <bool LIVE>
__global__ __launch_bounds__(MLP_THREADS, 1) void k_nerf_mlp_bwd_h(
    const float* __restrict__ enc_t, uint32_t ld, const float* __restrict__ dirs, uint32_t dir_stride, uint32_t n,
    const uint32_t* __restrict__ n_dev, const float* __restrict__ w_density, const float* __restrict__ w_color,
    float pad_value, const float4* __restrict__ draw, float* __restrict__ denc_t, float* __restrict__ partial /*[grid][GW]*/,
    const uint32_t* __restrict__ live_rows, const uint32_t* __restrict__ n_live) {
    if (n_dev) n = min(n, *n_dev);
    if (LIVE) n = *n_live;
    using SD = NetShape<1>;
    using SC = NetShape<2>;
    using HD = HShape<1>;
    using HC = HShape<2>;
    constexpr int GW = SD::glb_floats + SC::glb_floats;
    constexpr int STAGE = 4 * 32 * HST;                                  // halves per wave
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    _Float16* wdf = ldsh;
    _Float16* wcf = wdf + HD::f_halves;
    _Float16* wdb = wcf + HC::f_halves;
    _Float16* wcb = wdb + HD::b_halves;
    _Float16* stage_all = wcb + HC::b_halves;                           // MLP_WAVES * STAGE halves; doubles as the fp32 reduction buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    _Float16* stage = stage_all + wave * STAGE;
    const uint32_t n_tiles = (n + 31) / 32, tstride = gridDim.x * MLP_WAVES;
    // (measured: fetching the NEXT tile's 23 inputs under the current tile costs 40 more spilled registers in this
    // one-wave-per-SIMD kernel: 86 -> 136 us at 2^18 samples; the forward kernel, with registers to spare, gains 28 -> 20 us)
    load_weights_h<1>(wdf, wdb, w_density, false);
    load_weights_h<2>(wcf, wcb, w_color, true);
    __syncthreads();

    f32x16 a_d0[2][1], a_d1[1][2], a_c0[2][1], a_c1[2][2], a_c2[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a_d0[0][0][r] = a_d0[1][0][r] = 0.f; a_d1[0][0][r] = a_d1[0][1][r] = 0.f;
        a_c0[0][0][r] = a_c0[1][0][r] = 0.f;
        a_c1[0][0][r] = a_c1[0][1][r] = a_c1[1][0][r] = a_c1[1][1][r] = 0.f;
        a_c2[0][0][r] = a_c2[0][1][r] = 0.f;
    }
    for (uint32_t tile = blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += tstride) {
        const uint32_t s0 = tile * 32 + col;
        const bool live = s0 < n;
        const uint32_t s = LIVE ? live_rows[live ? s0 : n - 1] : s0;
        const uint32_t sc = LIVE ? s : (live ? s : n - 1);
        // ---- recompute the forward; every activation is kept in the fp16 form the forward used
        f32x16 t0, t2[2], dout[1];
        load_enc_tile(enc_t, ld, sc, t0, hi);
        HTile xe[1] = {to_h(t0)};
        layer_fwd_h<1, 2, HSB>(wdf + HD::f_off(0), xe, t2, col, hi);
        relu_tile(t2[0]); relu_tile(t2[1]);
        HTile hd[2] = {to_h(t2[0]), to_h(t2[1])};
        layer_fwd_h<2, 1, HSB>(wdf + HD::f_off(1), hd, dout, col, hi);
        build_color_in(dout[0], dirs, dir_stride, sc, pad_value, t0, hi);
        HTile cin[1] = {to_h(t0)};
        layer_fwd_h<1, 2, HSB>(wcf + HC::f_off(0), cin, t2, col, hi);
        relu_tile(t2[0]); relu_tile(t2[1]);
        HTile hc1[2] = {to_h(t2[0]), to_h(t2[1])};
        layer_fwd_h<2, 2, HSB>(wcf + HC::f_off(1), hc1, t2, col, hi);
        relu_tile(t2[0]); relu_tile(t2[1]);
        HTile hc2[2] = {to_h(t2[0]), to_h(t2[1])};
        // ---- output gradients (x loss scale)
        float4 drc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && hi == 0) drc = draw[s];
        f32x16 g1, g2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[r] = 0.f;
        g1[0] = drc.x * H16_LOSS_SCALE; g1[1] = drc.y * H16_LOSS_SCALE; g1[2] = drc.z * H16_LOSS_SCALE;
        HTile gh1[1] = {to_h(g1)};
        // color output layer
        dw_stage_h<1, 2>(gh1, hc2, stage, col, hi);
        layer_bwd_h<1, 2, 1, HSB>(wcb + HC::b_off(2), gh1, g2, col, hi);          // rows 0..2: the first K-step only
        dw_mfma_h<1, 2>(a_c2, stage, col, hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) {                                        // relu mask by the fp16 activation's sign
            g2[0][r] = (float)hc2[0].p[r >> 3][r & 7] > 0.f ? g2[0][r] : 0.f;
            g2[1][r] = (float)hc2[1].p[r >> 3][r & 7] > 0.f ? g2[1][r] : 0.f;
        }
        HTile gh2[2] = {to_h(g2[0]), to_h(g2[1])};
        // color hidden layer 2
        dw_stage_h<2, 2>(gh2, hc1, stage, col, hi);
        layer_bwd_h<2, 2, 4, HSB>(wcb + HC::b_off(1), gh2, g2, col, hi);
        dw_mfma_h<2, 2>(a_c1, stage, col, hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g2[0][r] = (float)hc1[0].p[r >> 3][r & 7] > 0.f ? g2[0][r] : 0.f;
            g2[1][r] = (float)hc1[1].p[r >> 3][r & 7] > 0.f ? g2[1][r] : 0.f;
        }
        gh2[0] = to_h(g2[0]); gh2[1] = to_h(g2[1]);
        // color input layer
        dw_stage_h<2, 1>(gh2, cin, stage, col, hi);
        f32x16 gi[1];
        layer_bwd_h<2, 1, 4, HSB>(wcb + HC::b_off(0), gh2, gi, col, hi);             // dL/d(color input slots)
        dw_mfma_h<2, 1>(a_c0, stage, col, hi);
#pragma unroll
        for (int r = 8; r < 16; ++r) gi[0][r] = 0.f;                          // slots >= 16 are SH / padding, not density outputs
        if (hi == 0) gi[0][0] = drc.w * H16_LOSS_SCALE;                       // row 0 = sigma
        gh1[0] = to_h(gi[0]);
        // density output layer
        dw_stage_h<1, 2>(gh1, hd, stage, col, hi);
        layer_bwd_h<1, 2, 1, HSB>(wdb + HD::b_off(1), gh1, g2, col, hi);          // 16 real output neurons: the first K-step
        dw_mfma_h<1, 2>(a_d1, stage, col, hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g2[0][r] = (float)hd[0].p[r >> 3][r & 7] > 0.f ? g2[0][r] : 0.f;
            g2[1][r] = (float)hd[1].p[r >> 3][r & 7] > 0.f ? g2[1][r] : 0.f;
        }
        gh2[0] = to_h(g2[0]); gh2[1] = to_h(g2[1]);
        // density input layer
        dw_stage_h<2, 1>(gh2, xe, stage, col, hi);
        layer_bwd_h<2, 1, 4, HSB>(wdb + HD::b_off(0), gh2, gi, col, hi);             // dL/d(encoded features) x loss scale
        dw_mfma_h<2, 1>(a_d0, stage, col, hi);
        if (live) {
            store_enc_tile(denc_t, ld, s, gi[0], hi, 1.0f / H16_LOSS_SCALE);
        }
    }
    constexpr float inv = 1.0f / H16_LOSS_SCALE;
    scale_tile(a_d0[0][0], inv); scale_tile(a_d0[1][0], inv); scale_tile(a_d1[0][0], inv); scale_tile(a_d1[0][1], inv);
    scale_tile(a_c0[0][0], inv); scale_tile(a_c0[1][0], inv);
    scale_tile(a_c1[0][0], inv); scale_tile(a_c1[0][1], inv); scale_tile(a_c1[1][0], inv); scale_tile(a_c1[1][1], inv);
    scale_tile(a_c2[0][0], inv); scale_tile(a_c2[0][1], inv);
    // ---- block reduction of dW through LDS (fp32, compact global layout), then one partial per block
    // two copies over the whole (now dead) dynamic LDS, two turns: see k_nerf_mlp_bwd_1_2
    __syncthreads();
    static_assert(2 * GW * sizeof(float) <= (size_t)(HD::f_halves + HC::f_halves + HD::b_halves + HC::b_halves + MLP_WAVES * STAGE) * 2,
                  "two reduction copies must fit the launch's LDS");
    float* redA = reinterpret_cast<float*>(ldsh);
    for (int ph = 0; ph < 2; ++ph) {
        if ((wave >> 1) == ph) {
            float* red = redA + (wave & 1) * GW;
            float* redc = red + SD::glb_floats;
            dw_flush<2, 1>(a_d0, red + SD::glb_off(0), 64, 32, false, col, hi, ph == 0);
            dw_flush<1, 2>(a_d1, red + SD::glb_off(1), 16, 64, false, col, hi, ph == 0);
            dw_flush<2, 1>(a_c0, redc + SC::glb_off(0), 64, 32, true, col, hi, ph == 0);
            dw_flush<2, 2>(a_c1, redc + SC::glb_off(1), 64, 64, false, col, hi, ph == 0);
            dw_flush<1, 2>(a_c2, redc + SC::glb_off(2), 16, 64, false, col, hi, ph == 0);
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * GW;
    for (int e = threadIdx.x; e < GW; e += MLP_THREADS) out[e] = redA[e] + redA[GW + e];
}

// ==================================================================================================================
// fp32 results on the bf16 matrix cores: 3-way operand splitting
// ==================================================================================================================
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).  A fp32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits, same exponent range): x = xh + xm + xl with xh = bf16(x), xm = bf16(x - xh),
// xl = bf16(x - xh - xm), both differences exact in fp32.  A product of two fp32 operands is then
//     w.x = wh.xh + (wh.xm + wm.xh) + (wh.xl + wl.xh + wm.xm) + [wm.xl + wl.xm + wl.xl],
// each bf16 x bf16 product exact in fp32; the bracket is below 2^-23 of |w||x| and is dropped, i.e. the six kept products
// carry the product to fp32 rounding accuracy.  Six v_mfma_f32_32x32x16_bf16 (fp32 accumulate, smallest terms first)
// replace eight v_mfma_f32_32x32x2_f32: 6 x 8 passes instead of 8 x 16 -- 0.375 of the matrix-core time -- and the operand
// traffic from LDS drops 4x (three 16-B reads per 16 k's and 32 rows instead of eight 4-B reads per 2 k's).  This is the
// forward of the fp32 parity mode (same 1e-4 bar against the oracle, tests/test_gpu_tcnn.py); the result differs from the
// fp32-MFMA kernel at the level two fp32 summation orders differ.  Same tile / k-slot scheme as the fp16 kernels above:
// activations go from the accumulators to the next layer's B operands by a register-local split, the weights sit in LDS
// pre-split and pre-permuted (three copies of the fp16 kernels' [row][hi][t][8] arrangement: 84 KiB for both networks,
// hence ONE workgroup of 8 waves per CU sharing them).  Topology (1, 2) only.
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#ifndef BX_WAVES
#define BX_WAVES 8
#endif
#ifndef BX_PF_OP
#define BX_PF_OP 1        // LDS operand double buffer (1 step ahead); 0: the same 50 us in the loop at 8 waves
#endif
#ifndef BX_PF_TILE
#define BX_PF_TILE 1      // next tile's inputs fetched under the current tile
#endif
#define BX_THREADS (BX_WAVES * 64)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;          // exact
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;         // exact
    l = (__bf16)r2;
}
struct BTile { b8 p[3][2]; };                // a 32 x 32 tile as B operands: [hi | mid | lo part][K-step]
// the same split on two values at a time: written on 2-vectors so that every conversion is ONE v_cvt_pk_bf16_f32 for the pair
// and the differences are packed subtractions (element by element the compiler paired only some of them: 6.1 vector
// instructions per element against 4.5 -- in a kernel bound by how many instructions a SIMD can issue)
__device__ __forceinline__ void split3_pair(f32x2 x, bf16x2& h, bf16x2& m, bf16x2& l) {
    h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);          // exact
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);         // exact
    l = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ BTile to_b3(const f32x16& t) {
    BTile r;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            bf16x2 h, m, l;
            split3_pair(f32x2{t[8 * k + e], t[8 * k + e + 1]}, h, m, l);
            r.p[0][k][e] = h[0]; r.p[0][k][e + 1] = h[1];
            r.p[1][k][e] = m[0]; r.p[1][k][e + 1] = m[1];
            r.p[2][k][e] = l[0]; r.p[2][k][e + 1] = l[1];
        }
    return r;
}
// global fp32 [out][in] -> LDS, three bf16 parts `ps` halves apart, each in the forward arrangement of store_layer_h
template <int NH, int L>
__device__ __forceinline__ void fetch_layer_b3(float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / BX_THREADS],
                                               const float* __restrict__ w) {
    using S = NetShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L);
    static_assert(prow * K % BX_THREADS == 0, "layer size must be a multiple of the workgroup size");
    const float* src = w + S::glb_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / BX_THREADS; ++i) {
        const int x = threadIdx.x + i * BX_THREADS, o = x / K, c = x % K;
        v[i] = src[(o < rows ? o : rows - 1) * K + c];                    // padded rows: any valid address, zeroed on store
    }
}
template <int NH, int L>
__device__ __forceinline__ void store_layer_b3(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / BX_THREADS],
                                               __bf16* __restrict__ wf, int ps, bool first_layer_rot) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L), ns = K / 16, rs = h_rs(K);
    __bf16* dst = wf + H::f_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / BX_THREADS; ++i) {
        const int x = threadIdx.x + i * BX_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;           // LDS slot of global column c
        __bf16 h, mi, lo;
        split3(o < rows ? v[i] : 0.f, h, mi, lo);
        __bf16* d = dst + o * rs + hslot(m, ns);
        d[0] = h; d[ps] = mi; d[2 * ps] = lo;
    }
}
template <int NH>
__device__ inline void load_weights_b3(__bf16* __restrict__ wf, int ps, const float* __restrict__ w, bool first_layer_rot) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "the split forward is built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / BX_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / BX_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / BX_THREADS : 1];
    fetch_layer_b3<NH, 0>(v0, w);                  // every global load in flight before the first LDS store
    fetch_layer_b3<NH, 1>(v1, w);
    if constexpr (NH >= 2) fetch_layer_b3<NH, 2>(v2, w);
    store_layer_b3<NH, 0>(v0, wf, ps, first_layer_rot);
    store_layer_b3<NH, 1>(v1, wf, ps, false);
    if constexpr (NH >= 2) store_layer_b3<NH, 2>(v2, wf, ps, false);
}
// out[TO] = W . in[TI] to fp32 accuracy: per K-step and output tile three 16-B operand reads and six MFMAs.  The operand
// reads of step t + 1 are issued before the MFMAs of step t, fenced by scheduling barriers: a scheduler free to hoist every
// ds_read_b128 of a layer (12 VGPRs per step and output tile) to the top spills the accumulators (165 VGPRs of scratch).
template <int TI, int TO>
__device__ __forceinline__ void layer_fwd_b3(const __bf16* __restrict__ wf, int ps, const BTile (&in)[TI], f32x16 (&out)[TO], int col, int hi) {
    constexpr int NS = 2 * TI, RS = 2 * NS * 8 + 8;
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[to][r] = 0.f;
    const __bf16* wl = wf + col * RS + hi * NS * 8;
    b8 a[BX_PF_OP + 1][TO][3];
    auto load = [&](int t, b8 (&d)[TO][3]) {
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const __bf16* p = wl + to * 32 * RS + t * 8;
            d[to][0] = *reinterpret_cast<const b8*>(p);
            d[to][1] = *reinterpret_cast<const b8*>(p + ps);
            d[to][2] = *reinterpret_cast<const b8*>(p + 2 * ps);
        }
    };
    if (BX_PF_OP) load(0, a[0]);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (BX_PF_OP) { if (t + 1 < NS) load(t + 1, a[(t + 1) & 1]); }
        else load(t, a[0]);
        const BTile& x = in[t >> 1];
        const int k = t & 1;
        const int cur = BX_PF_OP ? (t & 1) : 0;
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const b8 ah = a[cur][to][0], am = a[cur][to][1], al = a[cur][to][2];
            out[to] = MFMA16B(al, x.p[0][k], out[to]);
            out[to] = MFMA16B(ah, x.p[2][k], out[to]);
            out[to] = MFMA16B(am, x.p[1][k], out[to]);
            out[to] = MFMA16B(am, x.p[0][k], out[to]);
            out[to] = MFMA16B(ah, x.p[1][k], out[to]);
            out[to] = MFMA16B(ah, x.p[0][k], out[to]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the same on the FP16 matrix cores with a 2-way split (round 5) -------------------------------------------------------------
// fp16 carries 11 significand bits, so x = xh + xl with xh = fp16(x), xl = fp16(x - xh) (the difference exact in fp32) keeps 22 bits,
// and the three products wh.xh + wh.xl + wl.xh -- each exact in the fp32 accumulator -- carry w.x to ~2^-21 of |w||x|: within a few
// ulps of fp32, at HALF the matrix instructions and ~60 % of the conversion instructions of the 3-way bf16 split (3 MFMAs and two
// conversions + one subtraction per value instead of 6 and 3 + 2).  What fp16 costs is exponent range: a low part below 2^-14 is a
// subnormal (absolute precision 2^-25 ~ 3e-8 -- the level at which the reference's own fp16 tcnn quantises EVERY value) and
// a value above 65504 would overflow (it is saturated and counted instead: H2_MAX above); the hash-grid features, O(1e-4) at
// initialisation, are therefore scaled by H2_IN_SCALE = 2^4 into the first layer and its accumulators scaled back (both exact).  The weights sit in LDS in two fp16 parts (56 KiB for both networks).
template <int NH, int L>
__device__ __forceinline__ void store_layer_h2(const float (&v)[NetShape<NH>::out_rows_lds(L) * NetShape<NH>::in_dim(L) / BX_THREADS],
                                               _Float16* __restrict__ wf, int ps, bool first_layer_rot, float& mx) {
    using S = NetShape<NH>;
    using H = HShape<NH>;
    constexpr int K = S::in_dim(L), rows = S::out_dim(L), prow = S::out_rows_lds(L), ns = K / 16, rs = h_rs(K);
    _Float16* dst = wf + H::f_off(L);
#pragma unroll
    for (int i = 0; i < prow * K / BX_THREADS; ++i) {
        const int x = threadIdx.x + i * BX_THREADS, o = x / K, c = x % K;        // global [o][c]
        const int m = (L == 0 && first_layer_rot) ? ((c + 1) & 31) : c;           // LDS slot of global column c
        mx = fmaxf(mx, fabsf(v[i]));
        const float w = o < rows ? h2_sat(v[i]) : 0.f;
        const _Float16 h = (_Float16)w;
        _Float16* d = dst + o * rs + hslot(m, ns);
        d[0] = h; d[ps] = (_Float16)(w - (float)h);
    }
}
template <int NH>
__device__ inline void load_weights_h2(_Float16* __restrict__ wf, int ps, const float* __restrict__ w, bool first_layer_rot, float& mx) {
    using S = NetShape<NH>;
    static_assert(NH == 1 || NH == 2, "built for 1 or 2 hidden layers");
    float v0[S::out_rows_lds(0) * S::in_dim(0) / BX_THREADS], v1[S::out_rows_lds(1) * S::in_dim(1) / BX_THREADS];
    float v2[NH >= 2 ? S::out_rows_lds(NH >= 2 ? 2 : 0) * S::in_dim(NH >= 2 ? 2 : 0) / BX_THREADS : 1];
    fetch_layer_b3<NH, 0>(v0, w);
    fetch_layer_b3<NH, 1>(v1, w);
    if constexpr (NH >= 2) fetch_layer_b3<NH, 2>(v2, w);
    store_layer_h2<NH, 0>(v0, wf, ps, first_layer_rot, mx);
    store_layer_h2<NH, 1>(v1, wf, ps, false, mx);
    if constexpr (NH >= 2) store_layer_h2<NH, 2>(v2, wf, ps, false, mx);
}
// TRACK: with the range count (a launch that was handed a range word); without it the saturation alone (one v_med3 per value, in the
// ReLU's place: free) -- the max chain behind the count is 0.5 VALU instruction per operand on a kernel that is bound by exactly those
// (27.5 -> 31.9 us at 2^18 rows), so the training loop's normal iterations run untracked and the refresh iterations tracked.
template <bool WITH_COLOR, bool TRACK>
__global__ __launch_bounds__(BX_THREADS, 2) void k_nerf_mlp_fwd_h2(const float* __restrict__ enc_t, uint32_t ld,
                                                                    const float* __restrict__ dirs, uint32_t dir_stride,
                                                                    uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                    const uint32_t* __restrict__ rows,
                                                                    const float* __restrict__ w_density,
                                                                    const float* __restrict__ w_color, float pad_value,
                                                                    float4* __restrict__ raw, const int32_t* __restrict__ splat_idx,
                                                                    float* __restrict__ splat_grid, uint32_t* __restrict__ range_word) {
    if (n_dev) n = min(n, *n_dev);
    if (n == 0) return;
    using HD = HShape<1>;
    using HC = HShape<2>;
    constexpr int PD = HD::f_halves, PC = HC::f_halves;             // halves per part
    float mx = 0.f;                                                 // largest operand magnitude this thread split (h2_range_report)
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh2[];
    _Float16* wd = ldsh2;
    _Float16* wc = ldsh2 + 2 * PD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const uint32_t n_tiles = (n + 31) / 32, stride = gridDim.x * BX_WAVES;
    uint32_t tile = blockIdx.x * BX_WAVES + wave;
    f32x16 x;
    float d3[3] = {0.f, 0.f, 0.f};
    auto fetch = [&](uint32_t tl, f32x16& xe, float (&dd)[3]) {
        const uint32_t s = tl * 32 + col, sc = s < n ? s : n - 1;
        load_enc_tile(enc_t, ld, sc, xe, hi);
        if (WITH_COLOR) {
            const float* d = dirs + (size_t)(rows ? rows[sc] : sc) * dir_stride;
            dd[0] = d[0]; dd[1] = d[1]; dd[2] = d[2];
        }
    };
    if (tile < n_tiles) fetch(tile, x, d3);
    load_weights_h2<1>(wd, PD, w_density, false, mx);
    if (WITH_COLOR) load_weights_h2<2>(wc, PC, w_color, true, mx);
    __syncthreads();
    for (; tile < n_tiles; tile += stride) {
        const uint32_t s = tile * 32 + col;
        H2Tile xin[1] = {to_h2<true, TRACK>(x, H2_IN_SCALE, mx)};
        const float dx = d3[0], dy = d3[1], dz = d3[2];
        if (tile + stride < n_tiles) fetch(tile + stride, x, d3);   // next tile's loads under this tile's MFMAs
        f32x16 h[2], dout[1];
        layer_fwd_h2<1, 2>(wd + HD::f_off(0), PD, xin, h, col, hi);
        h2_relu_sat<TRACK>(h[0], 1.0f / H2_IN_SCALE, mx);            // relu + the input scale taken out again (exact) + the saturation
        h2_relu_sat<TRACK>(h[1], 1.0f / H2_IN_SCALE, mx);
        H2Tile hh[2] = {to_h2_act(h[0]), to_h2_act(h[1])};
        layer_fwd_h2<2, 1>(wd + HD::f_off(1), PD, hh, dout, col, hi);
        float4 o = make_float4(0.f, 0.f, 0.f, dout[0][0]);
        if (WITH_COLOR) {
            f32x16 cin, cout[1];
            const float dd[3] = {dx, dy, dz};
            build_color_in(dout[0], dd, 3, 0, pad_value, cin, hi);
            H2Tile ci[1] = {to_h2<true, TRACK>(cin, 1.0f, mx)};
            layer_fwd_h2<1, 2>(wc + HC::f_off(0), PC, ci, h, col, hi);
            h2_relu_sat<TRACK>(h[0], 1.0f, mx); h2_relu_sat<TRACK>(h[1], 1.0f, mx);
            hh[0] = to_h2_act(h[0]); hh[1] = to_h2_act(h[1]);
            layer_fwd_h2<2, 2>(wc + HC::f_off(1), PC, hh, h, col, hi);
            h2_relu_sat<TRACK>(h[0], 1.0f, mx); h2_relu_sat<TRACK>(h[1], 1.0f, mx);
            hh[0] = to_h2_act(h[0]); hh[1] = to_h2_act(h[1]);
            layer_fwd_h2<2, 1>(wc + HC::f_off(2), PC, hh, cout, col, hi);
            o.x = cout[0][0]; o.y = cout[0][1]; o.z = cout[0][2];
        }
        if (hi == 0 && s < n) {
            if (!WITH_COLOR && splat_idx != nullptr) atomicMax((uint32_t*)&splat_grid[(uint32_t)splat_idx[s]], __float_as_uint(expf(o.w) * xr_min_step()));
            else raw[s] = o;
        }
    }
    if (TRACK) h2_range_report(mx, range_word);
}

// Measured and dropped: handing the tiles out dynamically (a ticket counter, because the next batch's ray march co-runs on
// ~50 CUs in the training loop and a static partition lasts as long as its slowest SIMD: 44 us alone, 60 us in the loop).
// One returning atomic per 32-sample tile is 8100 same-address atomics per launch, and those retire one per ~13.5 ns: the
// launch took 130 us (profiles/r02_mlp_fwd_bf16x3_dynamic_tickets_negative.txt).
template <bool WITH_COLOR>
__global__ __launch_bounds__(BX_THREADS, 1) void k_nerf_mlp_fwd_b3(const float* __restrict__ enc_t, uint32_t ld,
                                                                    const float* __restrict__ dirs, uint32_t dir_stride,
                                                                    uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                    const uint32_t* __restrict__ rows,
                                                                    const float* __restrict__ w_density,
                                                                    const float* __restrict__ w_color, float pad_value,
                                                                    float4* __restrict__ raw, const int32_t* __restrict__ splat_idx,
                                                                    float* __restrict__ splat_grid) {
    if (n_dev) n = min(n, *n_dev);
    if (n == 0) return;
    using HD = HShape<1>;
    using HC = HShape<2>;
    constexpr int PD = HD::f_halves, PC = HC::f_halves;             // halves per part
    static_assert(PD % 8 == 0 && PC % 8 == 0, "parts must keep the 16-byte alignment of the operand reads");
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
    __bf16* wd = ldsb;
    __bf16* wc = ldsb + 3 * PD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const uint32_t n_tiles = (n + 31) / 32, stride = gridDim.x * BX_WAVES;
    uint32_t tile = blockIdx.x * BX_WAVES + wave;
    // the first tile's inputs are in flight while the workgroup splits the weights
    f32x16 x;
    float d3[3] = {0.f, 0.f, 0.f};
    auto fetch = [&](uint32_t tl, f32x16& xe, float (&dd)[3]) {
        const uint32_t s = tl * 32 + col, sc = s < n ? s : n - 1;
        load_enc_tile(enc_t, ld, sc, xe, hi);
        if (WITH_COLOR) {
            const float* d = dirs + (size_t)(rows ? rows[sc] : sc) * dir_stride;
            dd[0] = d[0]; dd[1] = d[1]; dd[2] = d[2];
        }
    };
    if (tile < n_tiles) fetch(tile, x, d3);
    load_weights_b3<1>(wd, PD, w_density, false);
    if (WITH_COLOR) load_weights_b3<2>(wc, PC, w_color, true);
    __syncthreads();
    for (; tile < n_tiles; tile += stride) {
        const uint32_t s = tile * 32 + col;
        if (!BX_PF_TILE && tile >= (blockIdx.x + gridDim.x) * BX_WAVES) fetch(tile, x, d3);      // (the first tile came before the weight split)
        BTile xin[1] = {to_b3(x)};
        const float dx = d3[0], dy = d3[1], dz = d3[2];
        if (BX_PF_TILE && tile + stride < n_tiles) fetch(tile + stride, x, d3);   // next tile's loads under this tile's MFMAs
        f32x16 h[2], dout[1];
        layer_fwd_b3<1, 2>(wd + HD::f_off(0), PD, xin, h, col, hi);
        relu_tile(h[0]); relu_tile(h[1]);
        BTile hh[2] = {to_b3(h[0]), to_b3(h[1])};
        layer_fwd_b3<2, 1>(wd + HD::f_off(1), PD, hh, dout, col, hi);
        float4 o = make_float4(0.f, 0.f, 0.f, dout[0][0]);
        if (WITH_COLOR) {
            f32x16 cin, cout[1];
            const float dd[3] = {dx, dy, dz};
            build_color_in(dout[0], dd, 3, 0, pad_value, cin, hi);
            BTile ci[1] = {to_b3(cin)};
            layer_fwd_b3<1, 2>(wc + HC::f_off(0), PC, ci, h, col, hi);
            relu_tile(h[0]); relu_tile(h[1]);
            hh[0] = to_b3(h[0]); hh[1] = to_b3(h[1]);
            layer_fwd_b3<2, 2>(wc + HC::f_off(1), PC, hh, h, col, hi);
            relu_tile(h[0]); relu_tile(h[1]);
            hh[0] = to_b3(h[0]); hh[1] = to_b3(h[1]);
            layer_fwd_b3<2, 1>(wc + HC::f_off(2), PC, hh, cout, col, hi);
            o.x = cout[0][0]; o.y = cout[0][1]; o.z = cout[0][2];
        }
        if (hi == 0 && s < n) {
            // density-only launches of the grid refresh: K8 (splat_grid_samples_nerf_max_nearest_neighbor.cu:7-28) in the epilogue -- the
            // sample's optical thickness at the smallest step goes straight into its cell's maximum, `raw` is not written
            if (!WITH_COLOR && splat_idx != nullptr) atomicMax((uint32_t*)&splat_grid[(uint32_t)splat_idx[s]], __float_as_uint(expf(o.w) * xr_min_step()));
            else raw[s] = o;
        }
    }
}

// ==================================================================================================================
// Any depth: the layers' weights STREAMED through LDS
// ==================================================================================================================
// tiny-cuda-nn reads `n_hidden_layers` (default 5) and ignores keys it does not know; the reference's config writes `num_layers`
// (configs/instant_ngp/nerf_blender_local01.py:106-124), so the networks hashnerf_mlp.py:39-45 builds are very probably 5 + 5 hidden
// layers of 64 -- 77 824 flop per sample instead of 20 480.  Twelve 64-wide layers do not fit the 160 KB of LDS in any operand format
// the kernels above use (fp32: 164 KB, three bf16 parts: 280 KB), and their activations and weight-gradient accumulators do not fit
// a wave's registers.  So the loop nest is turned inside out: a workgroup takes a batch of sample tiles through ONE layer at a time.
//   * the layer's weights are fetched from global memory (L2-resident: 80 KB per network) into registers while the previous layer
//     computes, split and permuted into LDS behind it (two alternating buffers, one barrier per layer);
//   * every wave keeps the activations (forward) / gradients (backward) of its tile of 32 samples in registers from layer to layer
//     -- the transposed neurons x samples scheme of the kernels above, nothing goes through LDS between layers;
//   * forward arithmetic = the 2-way fp16 operand split of k_nerf_mlp_fwd_h2 (within a few ulps of fp32, layer_fwd_h2);
//   * the backward recomputes the forward with the SAME arithmetic (its ReLU decisions are the forward's bit for bit), leaves each
//     layer's input tile in a per-workgroup global scratch area as [neuron][32 samples] -- the layout the weight gradient's
//     contraction over samples reads its H operand from directly, 32 B per lane, no transposing LDS tile -- and the ReLU decisions as
//     one bit per value in LDS; then walks the layers backwards with W^T streamed in the 2-way split arrangement of layer_bwd_b2,
//     one layer's dW accumulators (<= 4 tiles) alive at a time, reduced across the waves through LDS into the workgroup's partial.
// All hidden layers are 64 x 64, so the depth is a RUNTIME loop count (any 1 <= n_hidden <= XR_MLP_MAX_HIDDEN per network).
#define DP_PS 4608                     // halves per bf16 part of a streamed layer: 64 rows x h_rs(64)
#define DP_FW 8                        // waves per workgroup, forward
#ifndef DEEP_CUT
#define DEEP_CUT 0          // measurement builds only (wrong results): 1 no dW flush, 2 no dW products, 4 no scratch stores, 8 no dX chain
#endif
#define DP_BW 8                        // waves per workgroup, backward
#define XR_MLP_MAX_HIDDEN 8

struct DeepLayer { int K, kshift, rows, prow, goff; };
__host__ __device__ inline int deep_glb_floats(int nh) { return 32 * W_HID + (nh - 1) * W_HID * W_HID + W_HID * 16; }
__device__ __forceinline__ DeepLayer deep_layer(int nh, int l) {
    DeepLayer d;
    d.K = l == 0 ? 32 : W_HID; d.kshift = l == 0 ? 5 : 6;
    d.rows = l == nh ? 16 : W_HID; d.prow = l == nh ? 32 : W_HID;
    d.goff = l == 0 ? 0 : 32 * W_HID + (l - 1) * W_HID * W_HID;
    return d;
}
// a layer's weights, global -> registers in SOURCE order (coalesced); padded rows read as zero
template <int THREADS>
__device__ __forceinline__ void deep_fetch(float (&v)[W_HID * W_HID / THREADS], const float* __restrict__ w, const DeepLayer L) {
#pragma unroll
    for (int i = 0; i < W_HID * W_HID / THREADS; ++i) {
        const int e = threadIdx.x + i * THREADS, o = e >> L.kshift, c = e & (L.K - 1);
        v[i] = (e < L.prow * L.K && o < L.rows) ? w[L.goff + o * L.K + c] : 0.f;
    }
}
// registers -> LDS, forward arrangement [out row][hi][K-step][8] in two fp16 parts (store_layer_h2)
template <int THREADS>
__device__ __forceinline__ void deep_commit_f2(const float (&v)[W_HID * W_HID / THREADS], _Float16* __restrict__ buf, const DeepLayer L, bool rot) {
    const int ns = L.K / 16, rs = 2 * ns * 8 + 8;
#pragma unroll
    for (int i = 0; i < W_HID * W_HID / THREADS; ++i) {
        const int e = threadIdx.x + i * THREADS, o = e >> L.kshift, c = e & (L.K - 1);
        if (e < L.prow * L.K) {
            const int m = rot ? ((c + 1) & 31) : c;
            const float ws = h2_sat(v[i]);
            const _Float16 h = (_Float16)ws;
            _Float16* d = buf + o * rs + hslot(m, ns);
            d[0] = h; d[DP_PS] = (_Float16)(ws - (float)h);
        }
    }
}
// registers -> LDS, transposed arrangement [in slot][hi][K-step][8] in two bf16 parts (store_layer_bt2)
template <int THREADS>
__device__ __forceinline__ void deep_commit_b2(const float (&v)[W_HID * W_HID / THREADS], __bf16* __restrict__ buf, const DeepLayer L, bool rot) {
    const int nso = L.prow / 16, rsb = 2 * nso * 8 + 8;
#pragma unroll
    for (int i = 0; i < W_HID * W_HID / THREADS; ++i) {
        const int e = threadIdx.x + i * THREADS, o = e >> L.kshift, c = e & (L.K - 1);
        if (e < L.prow * L.K) {
            const int m = rot ? ((c + 1) & 31) : c;
            const __bf16 h = (__bf16)v[i];
            __bf16* d = buf + m * rsb + hslot(o, nso);
            d[0] = h; d[DP_PS] = (__bf16)(v[i] - (float)h);
        }
    }
}
__device__ __forceinline__ uint32_t relu_tile_bits(f32x16& t) {           // relu in place -> one bit per value that stayed
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const bool on = t[r] > 0.f; m |= on ? (1u << r) : 0u; t[r] = on ? fminf(t[r], H2_MAX) : 0.f; }   // (saturated as h2_relu_sat)
    return m;
}
__device__ __forceinline__ void mask_tile_bits(f32x16& g, uint32_t m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = ((m >> r) & 1u) ? g[r] : 0.f;
}

template <bool WITH_COLOR, bool TRACK>
__global__ __launch_bounds__(DP_FW * 64, 1) void k_nerf_mlp_fwd_deep(const float* __restrict__ enc_t, uint32_t ld,
                                                                     const float* __restrict__ dirs, uint32_t dir_stride,
                                                                     uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                     const uint32_t* __restrict__ rows,
                                                                     const float* __restrict__ w_density, const float* __restrict__ w_color,
                                                                     int nhd, int nhc, float pad_value, float4* __restrict__ raw,
                                                                     const int32_t* __restrict__ splat_idx, float* __restrict__ splat_grid,
                                                                     uint32_t* __restrict__ range_word) {
    if (n_dev) n = min(n, *n_dev);
    if (n == 0) return;
    float mx = 0.f;                                                     // (h2_range_report; the weights' range is the (1, 2) kernel's and the trainer's check)
    constexpr int THREADS = DP_FW * 64;
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];        // two layer buffers of two 16-bit parts each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    const int n_stage = (nhd + 1) + (WITH_COLOR ? nhc + 1 : 0);
    // stage g of a pass: layer (net, l)
    float v[W_HID * W_HID / THREADS];
    int k = 0;                                                          // stages done so far (buffer parity)
    auto fetch_stage = [&](int g) {
        const int net = g > nhd ? 1 : 0, l = net ? g - (nhd + 1) : g;
        deep_fetch<THREADS>(v, net ? w_color : w_density, deep_layer(net ? nhc : nhd, l));
    };
    auto commit_stage = [&](int g, int slot) {
        const int net = g > nhd ? 1 : 0, l = net ? g - (nhd + 1) : g;
        deep_commit_f2<THREADS>(v, reinterpret_cast<_Float16*>(ldsb) + slot * 2 * DP_PS, deep_layer(net ? nhc : nhd, l), net == 1 && l == 0);
    };
    fetch_stage(0);
    commit_stage(0, 0);
    const uint32_t per_pass = DP_FW * 32;
    for (uint32_t base = blockIdx.x * per_pass; base < n; base += gridDim.x * per_pass) {
        const bool last_pass = base + gridDim.x * per_pass >= n;
        const uint32_t s = base + wave * 32 + col, sc = s < n ? s : n - 1;
        f32x16 x, h[2];
        float d3[3] = {0.f, 0.f, 0.f};
        load_enc_tile(enc_t, ld, sc, x, hi);
        if (WITH_COLOR) {
            const float* d = dirs + (size_t)(rows ? rows[sc] : sc) * dir_stride;
            d3[0] = d[0]; d3[1] = d[1]; d3[2] = d[2];
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < n_stage; ++g) {
            const bool has_next = !(last_pass && g + 1 == n_stage);
            const int gn = g + 1 == n_stage ? 0 : g + 1;
            __syncthreads();                       // stage g's weights are in place; everyone is done with the other buffer
            if (has_next) fetch_stage(gn);
            const _Float16* wf = reinterpret_cast<const _Float16*>(ldsb) + (k & 1) * 2 * DP_PS;
            const int net = g > nhd ? 1 : 0, l = net ? g - (nhd + 1) : g, nh = net ? nhc : nhd;
            if (l == 0) {
                // (the hash-grid features enter scaled by 2^4, the accumulators are scaled back: see k_nerf_mlp_fwd_h2)
                const float sc_in = net == 0 ? H2_IN_SCALE : 1.0f, sc_out = net == 0 ? 1.0f / H2_IN_SCALE : 1.0f;
                const H2Tile xin[1] = {to_h2<true, TRACK>(x, sc_in, mx)};
                layer_fwd_h2<1, 2>(wf, DP_PS, xin, h, col, hi);
                h2_relu_sat<TRACK>(h[0], sc_out, mx); h2_relu_sat<TRACK>(h[1], sc_out, mx);
            } else if (l < nh) {
                const H2Tile hh[2] = {to_h2_act(h[0]), to_h2_act(h[1])};
                layer_fwd_h2<2, 2>(wf, DP_PS, hh, h, col, hi);
                h2_relu_sat<TRACK>(h[0], 1.0f, mx); h2_relu_sat<TRACK>(h[1], 1.0f, mx);
            } else {
                const H2Tile hh[2] = {to_h2_act(h[0]), to_h2_act(h[1])};
                f32x16 dout[1];
                layer_fwd_h2<2, 1>(wf, DP_PS, hh, dout, col, hi);
                if (net == 0) {
                    o.w = dout[0][0];                                        // hi == 0, register 0 <-> row 0 = sigma
                    if (WITH_COLOR) build_color_in(dout[0], d3, 3, 0, pad_value, x, hi);     // x: the colour net's input tile
                } else { o.x = dout[0][0]; o.y = dout[0][1]; o.z = dout[0][2]; }
            }
            if (has_next) commit_stage(gn, (k + 1) & 1);
            ++k;
        }
        if (hi == 0 && s < n) {
            if (!WITH_COLOR && splat_idx != nullptr) atomicMax((uint32_t*)&splat_grid[(uint32_t)splat_idx[s]], __float_as_uint(expf(o.w) * xr_min_step()));
            else raw[s] = o;
        }
    }
    if (TRACK) h2_range_report(mx, range_word);
}

template <int TO>
__device__ __forceinline__ void dw_stage_g(const f32x16 (&g)[TO], float* __restrict__ stage, int col, int hi) {
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(to * 32 + drow(r) + 4 * hi) * ST33 + col] = g[to][r];
    __builtin_amdgcn_wave_barrier();
}
// a tile in the accumulator layout -> the scratch area's [neuron][32 samples]
__device__ __forceinline__ void scr_store_tile(float* __restrict__ hs, int tile, const f32x16& h, int col, int hi) {
    if (DEEP_CUT & 4) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) hs[(tile * 32 + drow(r) + 4 * hi) * 32 + col] = h[r];
}
template <int N>
__device__ __forceinline__ void zero_tiles(f32x16 (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[i][r] = 0.f;
}
// the dW of the layer just walked: the eight waves' accumulators -> four LDS copies (two turns) -> summed into the workgroup's partial.
// `old`: the partial's values of the passes before, fetched at the top of the stage (the read-modify-write's load latency would
// otherwise sit exposed at the end of every stage).
template <int TO, int TI>
__device__ __forceinline__ void deep_flush(const f32x16 (&acc)[TO][TI], float* __restrict__ red, float* __restrict__ dst, int rows, int K, bool rot,
                                           const float (&old)[W_HID * W_HID / (DP_BW * 64)], int wave, int col, int hi) {
    if (DEEP_CUT & 1) return;
    __syncthreads();                               // every wave is done with its staging tile (the copies alias the staging area)
    if (wave < 4) dw_flush<TO, TI>(acc, red + wave * W_HID * W_HID, rows, K, rot, col, hi, true);
    __syncthreads();
    if (wave >= 4) dw_flush<TO, TI>(acc, red + (wave & 3) * W_HID * W_HID, rows, K, rot, col, hi, false);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < W_HID * W_HID / (DP_BW * 64); ++i) {
        const int e = threadIdx.x + i * DP_BW * 64;
        if (e < rows * K) dst[e] = old[i] + ((red[e] + red[W_HID * W_HID + e]) + (red[2 * W_HID * W_HID + e] + red[3 * W_HID * W_HID + e]));
    }
}
// a layer's input tile(s) from the scratch area as the H operand of the weight gradient: fetched at the top of the stage, used behind the
// dX chain (two waves per SIMD do not hide a memory round trip issued right in front of its use)
template <int TI>
struct HOperand { float4 q[TI][2][2]; };
template <int TI>
__device__ __forceinline__ void fetch_h(HOperand<TI>& H, const float* __restrict__ hs, int col, int hi) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            H.q[ti][t][0] = *reinterpret_cast<const float4*>(hs + (ti * 32 + col) * 32 + 16 * t + 8 * hi);
            H.q[ti][t][1] = *reinterpret_cast<const float4*>(hs + (ti * 32 + col) * 32 + 16 * t + 8 * hi + 4);
        }
}
template <int TO, int TI>
__device__ __forceinline__ void dw_mfma_b2_h(f32x16 (&acc)[TO][TI], const float* __restrict__ stage, const HOperand<TI>& H, int col, int hi) {
    if (DEEP_CUT & 2) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bw8 ah[TO], al[TO], bh[TI], bl[TI];
#pragma unroll
        for (int to = 0; to < TO; ++to) split2x8(stage + (to * 32 + col) * ST33 + 16 * t + 8 * hi, ah[to], al[to]);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const float4 a = H.q[ti][t][0], b = H.q[ti][t][1];
            const float q[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            split2x8(q, bh[ti], bl[ti]);
        }
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[to], bh[ti], acc[to][ti], 0, 0, 0);
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bl[ti], acc[to][ti], 0, 0, 0);
                acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[to], bh[ti], acc[to][ti], 0, 0, 0);
            }
    }
    __builtin_amdgcn_wave_barrier();
}

template <bool LIVE>
__global__ __launch_bounds__(DP_BW * 64, 1) void k_nerf_mlp_bwd_deep(
    const float* __restrict__ enc_t, uint32_t ld, const float* __restrict__ dirs, uint32_t dir_stride, uint32_t n,
    const uint32_t* __restrict__ n_dev, const float* __restrict__ w_density, const float* __restrict__ w_color, int nhd, int nhc,
    float pad_value, const float4* __restrict__ draw, float* __restrict__ denc_t, float* __restrict__ partial /*[grid][GW]*/,
    float* __restrict__ scratch /*[grid][slots][waves][64][32]*/, const uint32_t* __restrict__ live_rows,
    const uint32_t* __restrict__ n_live) {
    if (n_dev) n = min(n, *n_dev);
    if (LIVE) n = *n_live;                                             // rows of the compacted space
    constexpr int THREADS = DP_BW * 64;
    const int gwd = deep_glb_floats(nhd), GW = gwd + deep_glb_floats(nhc);
    const int n_slot = nhd + nhc + 2;                                   // density: features, hidden 1..nhd; colour: input slots, hidden 1..nhc
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
    __bf16* wbuf = ldsb;                                                // two layer buffers of two 16-bit parts (recompute: fp16, backward: bf16)
    float* stage_all = reinterpret_cast<float*>(ldsb + 2 * 2 * DP_PS);  // per wave [64][33]; between layers: four copies of a layer's dW
    uint32_t* bits = reinterpret_cast<uint32_t*>(stage_all + DP_BW * 64 * ST33);   // [slot][wave][64 lanes]: the ReLU decisions of a lane's 32 values
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, hi = lane >> 5;
    float* stage = stage_all + wave * 64 * ST33;
    float* part = partial + (size_t)blockIdx.x * GW;
    float* scr0 = scratch + ((size_t)blockIdx.x * n_slot * DP_BW + wave) * 2048;
    auto scr = [&](int slot) { return scr0 + (size_t)slot * DP_BW * 2048; };
    auto bit = [&](int slot) -> uint32_t& { return bits[(slot * DP_BW + wave) * 64 + lane]; };
    // stages of a pass.  Forward recompute: density l = 0..nhd, colour l = 0..nhc-1 (its output is not needed).  Backward: colour
    // l = nhc..0, density l = nhd..0.
    const int nF = (nhd + 1) + nhc, n_stage = nF + (nhc + 1) + (nhd + 1);
    auto net_of = [&](int g) { return g < nF ? (g > nhd ? 1 : 0) : (g - nF <= nhc ? 1 : 0); };
    auto lay_of = [&](int g) { return g < nF ? (g > nhd ? g - (nhd + 1) : g) : (g - nF <= nhc ? nhc - (g - nF) : nhd - (g - nF - (nhc + 1))); };
    float v[W_HID * W_HID / THREADS];
    int k = 0;
    auto fetch_stage = [&](int g) {
        const int net = net_of(g);
        deep_fetch<THREADS>(v, net ? w_color : w_density, deep_layer(net ? nhc : nhd, lay_of(g)));
    };
    auto commit_stage = [&](int g, int slot) {
        const int net = net_of(g), l = lay_of(g);
        const DeepLayer L = deep_layer(net ? nhc : nhd, l);
        if (g < nF) deep_commit_f2<THREADS>(v, reinterpret_cast<_Float16*>(wbuf) + slot * 2 * DP_PS, L, net == 1 && l == 0);
        else deep_commit_b2<THREADS>(v, wbuf + slot * 2 * DP_PS, L, net == 1 && l == 0);
    };
    bool first_pass = true;
    fetch_stage(0);
    commit_stage(0, 0);
    const uint32_t per_pass = DP_BW * 32;
    for (uint32_t base = blockIdx.x * per_pass; base < n || first_pass; base += gridDim.x * per_pass) {
        const bool last_pass = base + gridDim.x * per_pass >= n;
        const uint32_t s0 = base + wave * 32 + col;
        const bool live = s0 < n;
        const uint32_t si = n ? (live ? s0 : n - 1) : 0;
        const uint32_t s = LIVE ? (n ? live_rows[si] : 0) : si;
        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && hi == 0) dr = draw[s];
        {
            // ---- forward recompute (its own scope: the activations are dead behind it, the gradients not alive in it)
            f32x16 x, h[2];
            load_enc_tile(enc_t, ld, s, x, hi);
            scr_store_tile(scr(0), 0, x, col, hi);
            for (int g = 0; g < nF; ++g) {
                __syncthreads();
                fetch_stage(g + 1);
                const _Float16* wl = reinterpret_cast<const _Float16*>(wbuf) + (k & 1) * 2 * DP_PS;
                const int net = net_of(g), l = lay_of(g), nh = net ? nhc : nhd, slot0 = net ? nhd + 1 : 0;
                if (l == 0) {
                    const H2Tile xin[1] = {to_h2(x, net == 0 ? H2_IN_SCALE : 1.0f)};
                    layer_fwd_h2<1, 2>(wl, DP_PS, xin, h, col, hi);
                    if (net == 0) { scale_tile(h[0], 1.0f / H2_IN_SCALE); scale_tile(h[1], 1.0f / H2_IN_SCALE); }
                } else if (l < nh) {
                    const H2Tile hh[2] = {to_h2_act(h[0]), to_h2_act(h[1])};
                    layer_fwd_h2<2, 2>(wl, DP_PS, hh, h, col, hi);
                } else {                                              // density output layer -> the colour net's input slots
                    const H2Tile hh[2] = {to_h2_act(h[0]), to_h2_act(h[1])};
                    f32x16 dout[1];
                    layer_fwd_h2<2, 1>(wl, DP_PS, hh, dout, col, hi);
                    build_color_in(dout[0], dirs, dir_stride, s, pad_value, x, hi);
                    scr_store_tile(scr(nhd + 1), 0, x, col, hi);
                }
                if (l < nh) {
                    const uint32_t b0 = relu_tile_bits(h[0]), b1 = relu_tile_bits(h[1]);
                    bit(slot0 + l + 1) = b0 | (b1 << 16);
                    float* hs = scr(slot0 + l + 1);
                    scr_store_tile(hs, 0, h[0], col, hi); scr_store_tile(hs, 1, h[1], col, hi);
                }
                commit_stage(g + 1, (k + 1) & 1);
                ++k;
            }
        }
        // ---- backward through the layers
        f32x16 g2[2], g1[1];
        for (int g = nF; g < n_stage; ++g) {
            const bool has_next = !(last_pass && g + 1 == n_stage);
            const int gn = g + 1 == n_stage ? 0 : g + 1;
            __syncthreads();
            if (has_next) fetch_stage(gn);
            const __bf16* wl = wbuf + (k & 1) * 2 * DP_PS;
            const int net = net_of(g), l = lay_of(g), nh = net ? nhc : nhd, slot0 = net ? nhd + 1 : 0;
            const DeepLayer LL = deep_layer(nh, l);
            float* dst = part + (net ? gwd : 0) + LL.goff;
            float old[W_HID * W_HID / THREADS];
#pragma unroll
            for (int i = 0; i < W_HID * W_HID / THREADS; ++i) {
                const int e = threadIdx.x + i * THREADS;
                old[i] = (!first_pass && e < LL.rows * LL.K) ? dst[e] : 0.f;
            }
            if (l == nh) {
                // ---------------------------------------------------------------- output layer of a network
                if (net == 1) {
                    zero_tiles(g1);
                    g1[0][0] = dr.x; g1[0][1] = dr.y; g1[0][2] = dr.z;          // hi == 1 lanes hold zeros
                }
                HOperand<2> H;
                fetch_h<2>(H, scr(slot0 + nh), col, hi);
                dw_stage_g<1>(g1, stage, col, hi);
                { const B2Tile gb[1] = {to_b2(g1[0])}; layer_bwd_b2<1, 2, 1>(wl, DP_PS, gb, g2, col, hi); }
                f32x16 acc[1][2];
                zero_tiles(acc[0]);
                dw_mfma_b2_h<1, 2>(acc, stage, H, col, hi);
                const uint32_t b = bit(slot0 + nh);
                mask_tile_bits(g2[0], b); mask_tile_bits(g2[1], b >> 16);
                deep_flush<1, 2>(acc, stage_all, dst, 16, 64, false, old, wave, col, hi);
            } else if (l > 0) {
                // ---------------------------------------------------------------- hidden layer
                HOperand<2> H;
                fetch_h<2>(H, scr(slot0 + l), col, hi);
                dw_stage_g<2>(g2, stage, col, hi);
                { const B2Tile gb[2] = {to_b2(g2[0]), to_b2(g2[1])}; layer_bwd_b2<2, 2>(wl, DP_PS, gb, g2, col, hi); }
                const uint32_t b = bit(slot0 + l);
                mask_tile_bits(g2[0], b); mask_tile_bits(g2[1], b >> 16);
                __builtin_amdgcn_sched_barrier(0);               // (the accumulators of the weight gradient come alive behind the dX chain)
                f32x16 acc[2][2];
                zero_tiles(acc[0]); zero_tiles(acc[1]);
                dw_mfma_b2_h<2, 2>(acc, stage, H, col, hi);
                deep_flush<2, 2>(acc, stage_all, dst, 64, 64, false, old, wave, col, hi);
            } else {
                // ---------------------------------------------------------------- input layer of a network
                HOperand<1> H;
                fetch_h<1>(H, scr(slot0), col, hi);
                dw_stage_g<2>(g2, stage, col, hi);
                { const B2Tile gb[2] = {to_b2(g2[0]), to_b2(g2[1])}; layer_bwd_b2<2, 1>(wl, DP_PS, gb, g1, col, hi); }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 acc[2][1];
                zero_tiles(acc[0]); zero_tiles(acc[1]);
                dw_mfma_b2_h<2, 1>(acc, stage, H, col, hi);
                if (net == 1) {
                    // slots 1..15 are density-output rows 1..15; row 0 takes dL/d(sigma raw); rows >= 16 (SH, pad) end here
#pragma unroll
                    for (int r = 8; r < 16; ++r) g1[0][r] = 0.f;
                    if (hi == 0) g1[0][0] = dr.w;
                } else if (live) store_enc_tile(denc_t, ld, s, g1[0], hi);
                deep_flush<2, 1>(acc, stage_all, dst, 64, 32, net == 1, old, wave, col, hi);
            }
            if (has_next) commit_stage(gn, (k + 1) & 1);
            ++k;
        }
        first_pass = false;
    }
}

// ------------------------------------------------------------------ host side
// set around a density-only forward by xr_nerf_density_splat: the launch splats instead of writing `raw`
static thread_local const int32_t* g_fwd_splat_idx = nullptr;
static thread_local float* g_fwd_splat_grid = nullptr;
// the caller's range word (xr_set_mlp_range_word, per host thread like the helper stream): the XR_MLP_F16X2 forwards count into it
static thread_local uint32_t* g_range_word = nullptr;
extern "C" int xr_set_mlp_range_word(uint32_t* word) { g_range_word = word; return XR_OK; }
static int g_cus = 0;
extern "C" int xr_device_cus(void) {       // (internal, hidden: xr_common.h)
    if (g_cus == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        g_cus = p.multiProcessorCount;
    }
    return g_cus;
}

template <int NHD, int NHC>
static int launch_fwd(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n, const uint32_t* n_dev,
                      const uint32_t* rows, const float* wd, const float* wc, float pad, float* raw, hipStream_t stream) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t n_tiles = (n + 31) / 32;
#ifndef XR_MLP_FWD_WGS_PER_CU
#define XR_MLP_FWD_WGS_PER_CU 3       // 1..3 (tools/build_variant.sh)
#endif
    const int per_cu = XR_MLP_FWD_WGS_PER_CU;
    const uint32_t grid = min(xr_div_up(n_tiles, MLP_WAVES), (uint32_t)cus * (uint32_t)per_cu);
    if (dirs) {
        const size_t lds = (NetShape<NHD>::lds_floats + NetShape<NHC>::lds_floats) * sizeof(float);
        auto k = k_nerf_mlp_fwd<NHD, NHC, true>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XR_EHIP;
        hipLaunchKernelGGL(k, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev, rows, wd, wc, pad, (float4*)raw, (const int32_t*)nullptr, (float*)nullptr);
    } else {
        const size_t lds = NetShape<NHD>::lds_floats * sizeof(float);
        auto k = k_nerf_mlp_fwd<NHD, NHC, false>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XR_EHIP;
        hipLaunchKernelGGL(k, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev, rows, wd, wc, pad, (float4*)raw, g_fwd_splat_idx, g_fwd_splat_grid);
    }
    return XR_OK;
}

// hipFuncSetAttribute is a driver call (~3-5 us of host time): once per kernel and size, not once per launch
static int mlp_set_lds(const void* kernel, size_t lds) {
    static const void* seen_k[32];
    static size_t seen_b[32];
    static int n_seen = 0;
    for (int i = 0; i < n_seen; ++i) if (seen_k[i] == kernel && seen_b[i] >= lds) return XR_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XR_EHIP;
    for (int i = 0; i < n_seen; ++i) if (seen_k[i] == kernel) { seen_b[i] = lds; return XR_OK; }
    if (n_seen < 32) { seen_k[n_seen] = kernel; seen_b[n_seen] = lds; ++n_seen; }
    return XR_OK;
}
// any depth: the streamed kernel (fp32 results on the bf16 matrix cores, 3-way operand split)
static int launch_fwd_deep(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n, const uint32_t* n_dev,
                           const uint32_t* rows, const float* wd, const float* wc, int nhd, int nhc, float pad, float* raw, hipStream_t stream) {
    XR_REQUIRE(nhd >= 1 && nhd <= XR_MLP_MAX_HIDDEN && nhc >= 1 && nhc <= XR_MLP_MAX_HIDDEN, "1..8 hidden layers per network");
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t grid = min(xr_div_up(n, DP_FW * 32), (uint32_t)cus);
    const size_t lds = (size_t)2 * 2 * DP_PS * sizeof(__bf16);
    auto k = dirs ? (g_range_word ? k_nerf_mlp_fwd_deep<true, true> : k_nerf_mlp_fwd_deep<true, false>)
                  : (g_range_word ? k_nerf_mlp_fwd_deep<false, true> : k_nerf_mlp_fwd_deep<false, false>);
    if (mlp_set_lds((const void*)k, lds) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(DP_FW * 64), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev, rows, wd, wc, nhd, nhc, pad, (float4*)raw,
                       dirs ? (const int32_t*)nullptr : g_fwd_splat_idx, dirs ? (float*)nullptr : g_fwd_splat_grid, g_range_word);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

static int mlp_fwd_f32(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                               const uint32_t* n_dev, const uint32_t* rows, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                               float pad_value, float* raw, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(enc_t && w_density && raw, "null pointer");
    XR_REQUIRE(!dirs || (w_color && dir_stride >= 3), "color path needs w_color and dir_stride >= 3");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && ((uintptr_t)raw & 15) == 0, "bad ld / raw alignment");
    int rc;
    if (n_hidden_density == 1 && n_hidden_color == 2) rc = launch_fwd<1, 2>(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, pad_value, raw, (hipStream_t)stream_);
    else if (n_hidden_density == 1 && n_hidden_color == 1) rc = launch_fwd<1, 1>(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, pad_value, raw, (hipStream_t)stream_);
    else if (n_hidden_density == 2 && n_hidden_color == 2) rc = launch_fwd<2, 2>(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, pad_value, raw, (hipStream_t)stream_);
    else if (n_hidden_density == 2 && n_hidden_color == 3) rc = launch_fwd<2, 3>(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, pad_value, raw, (hipStream_t)stream_);
    // any other depth (tcnn's default 5 + 5): the layers' weights streamed through LDS
    else return launch_fwd_deep(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, raw, (hipStream_t)stream_);
    if (rc != XR_OK) { xr_set_error("xr_nerf_mlp_fwd: cannot configure dynamic LDS"); return rc; }
    XR_LAUNCH_CHECK();
    return XR_OK;
}

static uint32_t bwd_grid(uint32_t n) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t n_tiles = (n + 31) / 32;
    return min(xr_div_up(n_tiles, MLP_WAVES), (uint32_t)cus);
}
static uint32_t bwd_grid_deep(uint32_t n) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    return max(1u, min(xr_div_up(n, DP_BW * 32), (uint32_t)cus));
}
static bool bwd_is_deep(int nhd, int nhc) { return !(nhd == 1 && nhc == 2); }
// workspace of a backward over n rows: live_rows[n] | seg_count[ceil(n / LIVE_SEG)] | n_live (4 words) | -> 256 B | dW partials
// [cus][GW] | streamed kernels: the activation scratch area [cus][nhd + nhc + 2][waves][64][32]
static size_t bwd_list_bytes(uint32_t n) { return ((((size_t)n + xr_div_up(n, LIVE_SEG) + 4) * sizeof(uint32_t)) + 255) & ~(size_t)255; }
static size_t bwd_partial_bytes(int nhd, int nhc) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const size_t gw = bwd_is_deep(nhd, nhc) ? (size_t)deep_glb_floats(nhd) + deep_glb_floats(nhc) : (size_t)NetShape<1>::glb_floats + NetShape<2>::glb_floats;
    return ((size_t)cus * gw * sizeof(float) + 255) & ~(size_t)255;
}
static size_t bwd_scratch_bytes(int nhd, int nhc) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    return bwd_is_deep(nhd, nhc) ? (size_t)cus * (nhd + nhc + 2) * DP_BW * 2048 * sizeof(float) : 0;
}
static float* bwd_partials(const void* workspace, uint32_t n) { return reinterpret_cast<float*>((char*)workspace + bwd_list_bytes(n)); }
extern "C" size_t xr_nerf_mlp_bwd_workspace_bytes(uint32_t n, int n_hidden_density, int n_hidden_color) {
    return bwd_list_bytes(n) + bwd_partial_bytes(n_hidden_density, n_hidden_color) + bwd_scratch_bytes(n_hidden_density, n_hidden_color);
}
static bool live_rows_enabled() {          // XR_MLP_LIVE=0: run the backward over every row (measurement)
    static int on = -1;
    if (on < 0) { const char* e = getenv("XR_MLP_LIVE"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}
static_assert(LIVE_SEG == XR_LIVE_SEGMENT_ROWS, "the header names the segment size");
// seg_counts_ready != 0: seg_count already holds the live rows per segment (xr_composite_train2 counted them while writing
// the rows) -- only the ranking / list pass runs
extern "C" int xr_live_rows(const float* dloss_doutput, uint32_t n, const uint32_t* n_dev, uint32_t* seg_count, uint32_t* live_rows,
                             uint32_t* n_live, float* zero_denc_t, uint32_t ld, int seg_counts_ready, void* stream_) {
    XR_REQUIRE(dloss_doutput && seg_count && live_rows && n_live, "null pointer");
    XR_REQUIRE(((uintptr_t)dloss_doutput & 15) == 0, "dloss_doutput must be 16-byte aligned");
    XR_REQUIRE(!zero_denc_t || ld >= n, "bad ld");
    hipStream_t stream = (hipStream_t)stream_;
    if (n == 0) { XR_HIP(hipMemsetAsync(n_live, 0, sizeof(uint32_t), stream)); return XR_OK; }
    const uint32_t n_seg = xr_div_up(n, LIVE_SEG);
    if (!seg_counts_ready)
        hipLaunchKernelGGL(k_live_count, dim3(n_seg), dim3(LIVE_THREADS), 0, stream, (const float4*)dloss_doutput, n, n_dev, seg_count);
    hipLaunchKernelGGL(k_live_fill, dim3(n_seg), dim3(LIVE_THREADS), 0, stream, (const float4*)dloss_doutput, n, n_dev,
                       (const uint32_t*)seg_count, n_seg, live_rows, n_live, zero_denc_t, ld);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
// where a list of n rows sits in an xr_nerf_mlp_bwd workspace (at its start, whatever the topology): callers that build the list
// themselves to share it with xr_hashgrid_bwd use these slots instead of allocating
extern "C" int xr_nerf_mlp_bwd_list_slots(void* workspace, size_t workspace_bytes, uint32_t n, uint32_t** live_rows,
                                          uint32_t** seg_count, uint32_t** n_live) {
    XR_REQUIRE(workspace && live_rows && seg_count && n_live, "null pointer");
    XR_REQUIRE(workspace_bytes >= xr_nerf_mlp_bwd_workspace_bytes(n, 1, 2), "workspace too small");
    *live_rows = reinterpret_cast<uint32_t*>(workspace);
    *seg_count = *live_rows + n;
    *n_live = *seg_count + xr_div_up(n, LIVE_SEG);
    return XR_OK;
}
// the backward's own list, in its workspace (callers that did not bring one): dead rows of denc_t are zeroed
static int build_live_rows(const float* draw, uint32_t n, const uint32_t* n_dev, float* denc_t, uint32_t ld, void* workspace,
                           hipStream_t stream, const uint32_t** rows, const uint32_t** n_live) {
    uint32_t* list = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* seg = list + n;
    uint32_t* cnt = seg + xr_div_up(n, LIVE_SEG);
    *rows = list; *n_live = cnt;
    return xr_live_rows(draw, n, n_dev, seg, list, cnt, denc_t, ld, 0, stream);
}

// xr_ngp_train_step runs the partial reduce on its helper stream beside the table scatter (only the optimiser reads the MLP
// gradients): it sets this flag around its backward call and issues xr_nerf_mlp_bwd_reduce itself
static thread_local bool g_defer_reduce = false;
void xr_internal_defer_mlp_reduce(bool on) { g_defer_reduce = on; }
int xr_internal_mlp_bwd_reduce(const void* workspace, uint32_t n, int nhd, int nhc, float* grad_w_density, float* grad_w_color, int overwrite, void* stream_) {
    XR_REQUIRE(workspace && grad_w_density && grad_w_color, "null pointer");
    const bool deep = bwd_is_deep(nhd, nhc);
    const uint32_t gwd = deep ? (uint32_t)deep_glb_floats(nhd) : (uint32_t)NetShape<1>::glb_floats;
    const uint32_t GW = gwd + (deep ? (uint32_t)deep_glb_floats(nhc) : (uint32_t)NetShape<2>::glb_floats);
    hipLaunchKernelGGL(k_reduce_partials, dim3(xr_div_up(GW, 64)), dim3(256), 0, (hipStream_t)stream_, (const float*)bwd_partials(workspace, n),
                       deep ? bwd_grid_deep(n) : bwd_grid(n), GW, gwd, grad_w_density, grad_w_color, overwrite);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_sum_partials(const float* partials, uint32_t n_partials, size_t stride, uint32_t n, float* out, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(partials && out && n_partials >= 1 && stride >= n && stride <= 0xffffffffull, "bad argument");
    // (k_reduce_partials addresses partial[b][j] with a row length of gw: the stride is the row length, columns >= n are not touched)
    hipLaunchKernelGGL(k_sum_partials, dim3(xr_div_up(n, 64)), dim3(256), 0, (hipStream_t)stream_, partials, n_partials, (uint32_t)stride, n, out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// the same sum as (1) of an XrAuxWork (xr_aux.h): the caller's launch runs it (overwrite / adam are the caller's to set)
int xr_internal_mlp_bwd_reduce_desc(const void* workspace, uint32_t n, int nhd, int nhc, XrAuxWork* w) {
    XR_REQUIRE(workspace && w, "null pointer");
    const bool deep = bwd_is_deep(nhd, nhc);
    const uint32_t gwd = deep ? (uint32_t)deep_glb_floats(nhd) : (uint32_t)NetShape<1>::glb_floats;
    w->partial = (const float*)bwd_partials(const_cast<void*>(workspace), n);
    w->nb = deep ? bwd_grid_deep(n) : bwd_grid(n);
    w->gw = gwd + (deep ? (uint32_t)deep_glb_floats(nhc) : (uint32_t)NetShape<2>::glb_floats);
    w->split = gwd;
    return XR_OK;
}

static int mlp_bwd_f32(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                               const uint32_t* n_dev, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                               float pad_value, const float* draw, float* denc_t, float* grad_w_density,
                               float* grad_w_color, void* workspace, size_t workspace_bytes, const uint32_t* live_rows,
        const uint32_t* n_live, void* stream_) {
    if (n == 0) return XR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(enc_t && dirs && w_density && w_color && draw && denc_t && grad_w_density && grad_w_color, "null pointer");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && dir_stride >= 3 && ((uintptr_t)draw & 15) == 0, "bad ld / stride / alignment");
    XR_REQUIRE(n_hidden_density >= 1 && n_hidden_density <= XR_MLP_MAX_HIDDEN && n_hidden_color >= 1 && n_hidden_color <= XR_MLP_MAX_HIDDEN,
               "1..8 hidden layers per network");
    XR_REQUIRE(workspace && workspace_bytes >= xr_nerf_mlp_bwd_workspace_bytes(n, n_hidden_density, n_hidden_color), "workspace too small");
    XR_REQUIRE(!live_rows == !n_live, "live_rows and n_live come together");
    const uint32_t* rows = live_rows;                    // the caller's list (xr_live_rows), else the backward's own
    if (!rows && live_rows_enabled()) { const int rc = build_live_rows(draw, n, n_dev, denc_t, ld, workspace, stream, &rows, &n_live); if (rc != XR_OK) return rc; }
    float* partials = bwd_partials(workspace, n);
    if (bwd_is_deep(n_hidden_density, n_hidden_color)) {
        // any depth but the reference config's (1, 2): the streamed kernel (k_nerf_mlp_bwd_deep)
        const int nhd = n_hidden_density, nhc = n_hidden_color;
        const uint32_t gwd = (uint32_t)deep_glb_floats(nhd), GWD = gwd + (uint32_t)deep_glb_floats(nhc);
        const uint32_t gridd = bwd_grid_deep(n);
        const size_t ldsd = (size_t)2 * 2 * DP_PS * sizeof(__bf16) + (size_t)DP_BW * 64 * ST33 * sizeof(float) +
                            (size_t)(nhd + nhc + 2) * DP_BW * 64 * sizeof(uint32_t);
        float* scratch = reinterpret_cast<float*>((char*)partials + bwd_partial_bytes(nhd, nhc));
        auto kd = rows ? k_nerf_mlp_bwd_deep<true> : k_nerf_mlp_bwd_deep<false>;
        if (mlp_set_lds((const void*)kd, ldsd) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
        hipLaunchKernelGGL(kd, dim3(gridd), dim3(DP_BW * 64), ldsd, stream, enc_t, ld, dirs, dir_stride, n, n_dev, w_density, w_color, nhd, nhc, pad_value,
                           (const float4*)draw, denc_t, partials, scratch, rows, n_live);
        if (!g_defer_reduce)
            hipLaunchKernelGGL(k_reduce_partials, dim3(xr_div_up(GWD, 64)), dim3(256), 0, stream, (const float*)partials, gridd, GWD, gwd,
                               grad_w_density, grad_w_color, 0);
        XR_LAUNCH_CHECK();
        return XR_OK;
    }
    constexpr int GW = NetShape<1>::glb_floats + NetShape<2>::glb_floats;
    size_t lds = (NetShape<1>::lds_floats + NetShape<2>::lds_floats + MLP_WAVES * 4 * 32 * ST33) * sizeof(float);
    const uint32_t grid = bwd_grid(n);
    // XR_MLP_BWD_DW (read per call): f32 = fp32 MFMA throughout; b2 = the dW products on the bf16 matrix cores (2-way split);
    // b2x = the dX chain too; h2f (default, round 5) = that with the forward recomputed on two FP16 parts -- the arithmetic of the
    // default forward (XR_MLP_F16X2), product for product, so the ReLU decisions are the forward's; b2f = the recompute on two
    // bf16 parts.  b2f is not the default: a recompute at 2^-16
    // relative accuracy puts a hidden unit whose pre-activation is within ~1e-5 of zero on the other side of its ReLU than
    // the forward had it (a few hundred unit-samples per training step, ~1 with the fp32 recompute) -- harmless to the
    // optimiser, but each such flip is one sample's whole contribution to a weight row, which a 1e-3 * max parity bar on a
    // small batch sees (profiles/NOTES_r01_r03.md 5f).
    const char* bwd_env = getenv("XR_MLP_BWD_DW");
    const int mode = !bwd_env ? 4 : strcmp(bwd_env, "f32") == 0 ? 0 : strcmp(bwd_env, "b2") == 0 ? 1 : strcmp(bwd_env, "b2x") == 0 ? 2
                   : strcmp(bwd_env, "b2f") == 0 ? 3 : strcmp(bwd_env, "h2f") == 0 ? 4 : -1;
    if (mode < 0) { xr_set_error("XR_MLP_BWD_DW=%s: expected f32, b2, b2x, b2f or h2f", bwd_env); return XR_EINVAL; }
    using KernT = void (*)(const float*, uint32_t, const float*, uint32_t, uint32_t, const uint32_t*, const float*, const float*, float,
                           const float4*, float*, float*, const uint32_t*, const uint32_t*);
    static const KernT kerns[2][5] = {{k_nerf_mlp_bwd_1_2<false, 0>, k_nerf_mlp_bwd_1_2<false, 1>, k_nerf_mlp_bwd_1_2<false, 2>, k_nerf_mlp_bwd_1_2<false, 3>,
                                       k_nerf_mlp_bwd_1_2<false, 4>},
                                      {k_nerf_mlp_bwd_1_2<true, 0>, k_nerf_mlp_bwd_1_2<true, 1>, k_nerf_mlp_bwd_1_2<true, 2>, k_nerf_mlp_bwd_1_2<true, 3>,
                                       k_nerf_mlp_bwd_1_2<true, 4>}};
    KernT kern = kerns[rows ? 1 : 0][mode];
    constexpr size_t bt2 = (size_t)2 * (HShape<1>::b_halves + HShape<2>::b_halves) * sizeof(__bf16);
    constexpr size_t ft2 = (size_t)2 * (F2Shape<1>::halves + F2Shape<2>::halves) * sizeof(__bf16);
    constexpr size_t w32 = (NetShape<1>::lds_floats + NetShape<2>::lds_floats) * sizeof(float);
    constexpr size_t st3 = (size_t)MLP_WAVES * 3 * 32 * ST33 * sizeof(float);
    if (mode == 2) lds = w32 + bt2 + st3;
    if (mode >= 3) lds = ft2 + bt2 + st3;
    if (mlp_set_lds((const void*)kern, lds) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n,
                       n_dev, w_density, w_color, pad_value, (const float4*)draw, denc_t, partials, rows, n_live);
    if (!g_defer_reduce)
        hipLaunchKernelGGL(k_reduce_partials, dim3(xr_div_up(GW, 64)), dim3(256), 0, stream, (const float*)partials, grid,
                           (uint32_t)GW, (uint32_t)NetShape<1>::glb_floats, grad_w_density, grad_w_color);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ---- generic single-network entry points (tcnn.Network compatibility surface)
extern "C" size_t xr_mlp_bwd_workspace_bytes(int n_hidden) {
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    return (size_t)cus * (n_hidden == 2 ? NetShape<2>::glb_floats : NetShape<1>::glb_floats) * sizeof(float);
}
extern "C" int xr_mlp_fwd(const float* x, long row_stride, long col_stride, int n_in, float pad_value, uint32_t n,
                          const float* w, int n_hidden, float* y, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(x && w && y && n_in >= 1 && n_in <= 32, "bad argument");
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t grid = min(xr_div_up((n + 31) / 32, MLP_WAVES), (uint32_t)cus * 2u);
    hipStream_t stream = (hipStream_t)stream_;
#define XR_FWD_CASE(NH)                                                                                              \
    case NH: {                                                                                                       \
        const size_t lds = NetShape<NH>::lds_floats * sizeof(float);                                                 \
        XR_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd<NH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_mlp_fwd<NH>, dim3(grid), dim3(MLP_THREADS), lds, stream, x, row_stride, col_stride, n_in, \
                           pad_value, n, w, y);                                                                      \
    } break;
    switch (n_hidden) {
        XR_FWD_CASE(1) XR_FWD_CASE(2) XR_FWD_CASE(3)
        default: xr_set_error("xr_mlp_fwd: n_hidden_layers %d not built (1..3)", n_hidden); return XR_EINVAL;
    }
#undef XR_FWD_CASE
    XR_LAUNCH_CHECK();
    return XR_OK;
}
extern "C" int xr_mlp_bwd(const float* x, long row_stride, long col_stride, int n_in, float pad_value, uint32_t n,
                          const float* w, int n_hidden, const float* dy, float* dx, float* grad_w, void* workspace,
                          size_t workspace_bytes, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(x && w && dy && grad_w && n_in >= 1 && n_in <= 32, "bad argument");
    XR_REQUIRE(n_hidden == 1 || n_hidden == 2, "backward is built for 1 or 2 hidden layers");
    XR_REQUIRE(workspace && workspace_bytes >= xr_mlp_bwd_workspace_bytes(n_hidden), "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t grid = bwd_grid(n);
    const int gw = n_hidden == 2 ? NetShape<2>::glb_floats : NetShape<1>::glb_floats;
    if (n_hidden == 1) {
        const size_t lds = (NetShape<1>::lds_floats + MLP_WAVES * 4 * 32 * ST33) * sizeof(float);
        XR_HIP(hipFuncSetAttribute((const void*)k_mlp_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mlp_bwd<1>, dim3(grid), dim3(MLP_THREADS), lds, stream, x, row_stride, col_stride, n_in, pad_value,
                           n, w, dy, dx, (float*)workspace);
    } else {
        const size_t lds = (NetShape<2>::lds_floats + MLP_WAVES * 4 * 32 * ST33) * sizeof(float);
        XR_HIP(hipFuncSetAttribute((const void*)k_mlp_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mlp_bwd<2>, dim3(grid), dim3(MLP_THREADS), lds, stream, x, row_stride, col_stride, n_in, pad_value,
                           n, w, dy, dx, (float*)workspace);
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(xr_div_up(gw, 64)), dim3(256), 0, stream, (const float*)workspace, grid,
                       (uint32_t)gw, (uint32_t)gw, grad_w, grad_w);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ---- reference-precision mode entry points (same contracts as xr_nerf_mlp_fwd / xr_nerf_mlp_bwd)
static int mlp_fwd_f16(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                                   const uint32_t* n_dev, const uint32_t* rows, const float* w_density, const float* w_color,
                                   int n_hidden_density, int n_hidden_color, float pad_value, float* raw, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(enc_t && w_density && raw, "null pointer");
    XR_REQUIRE(!dirs || (w_color && dir_stride >= 3), "color path needs w_color and dir_stride >= 3");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && ((uintptr_t)raw & 15) == 0, "bad ld / raw alignment");
    XR_REQUIRE(n_hidden_density == 1 && n_hidden_color == 2, "the fp16 mode is built for the (1,2) hidden-layer topology");
    hipStream_t stream = (hipStream_t)stream_;
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t grid = min(xr_div_up((n + 31) / 32, MLP_WAVES), (uint32_t)cus * 2u);     // resident: 2 workgroups per CU
    if (dirs) {
        const size_t lds = (size_t)(HShape<1>::f_halves + HShape<2>::f_halves) * 2;
        hipLaunchKernelGGL(k_nerf_mlp_fwd_h<true>, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev,
                           rows, w_density, w_color, pad_value, (float4*)raw, (const int32_t*)nullptr, (float*)nullptr);
    } else {
        const size_t lds = (size_t)HShape<1>::f_halves * 2;
        hipLaunchKernelGGL(k_nerf_mlp_fwd_h<false>, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev,
                           rows, w_density, w_color, pad_value, (float4*)raw, g_fwd_splat_idx, g_fwd_splat_grid);
    }
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// fp32-accuracy forward on the bf16 matrix cores (3-way operand split; same contract as xr_nerf_mlp_fwd, topology (1, 2))
static int mlp_fwd_bf16x3(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                                      const uint32_t* n_dev, const uint32_t* rows, const float* w_density, const float* w_color,
                                      int n_hidden_density, int n_hidden_color, float pad_value, float* raw, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(enc_t && w_density && raw, "null pointer");
    XR_REQUIRE(!dirs || (w_color && dir_stride >= 3), "color path needs w_color and dir_stride >= 3");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && ((uintptr_t)raw & 15) == 0, "bad ld / raw alignment");
    hipStream_t stream = (hipStream_t)stream_;
    if (!(n_hidden_density == 1 && n_hidden_color == 2))      // the same arithmetic at any depth: the layers' weights streamed through LDS
        return launch_fwd_deep(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, raw, stream);
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t grid = min(xr_div_up((n + 31) / 32, BX_WAVES), (uint32_t)cus);          // resident: one 8-wave workgroup per CU
    if (dirs) {
        const size_t lds = (size_t)3 * (HShape<1>::f_halves + HShape<2>::f_halves) * 2;
        if (mlp_set_lds((const void*)k_nerf_mlp_fwd_b3<true>, lds) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
        hipLaunchKernelGGL(k_nerf_mlp_fwd_b3<true>, dim3(grid), dim3(BX_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev,
                           rows, w_density, w_color, pad_value, (float4*)raw, (const int32_t*)nullptr, (float*)nullptr);
    } else {
        const size_t lds = (size_t)3 * HShape<1>::f_halves * 2;
        hipLaunchKernelGGL(k_nerf_mlp_fwd_b3<false>, dim3(grid), dim3(BX_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev,
                           rows, w_density, w_color, pad_value, (float4*)raw, g_fwd_splat_idx, g_fwd_splat_grid);
    }
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// fp32-accurate forward on the fp16 matrix cores (2-way operand split: k_nerf_mlp_fwd_h2; same contract as xr_nerf_mlp_fwd).  Any
// depth but (1, 2) takes the streamed kernel, which uses the same arithmetic.
static int mlp_fwd_f16x2(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                                     const uint32_t* n_dev, const uint32_t* rows, const float* w_density, const float* w_color,
                                     int n_hidden_density, int n_hidden_color, float pad_value, float* raw, void* stream_) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(enc_t && w_density && raw, "null pointer");
    XR_REQUIRE(!dirs || (w_color && dir_stride >= 3), "color path needs w_color and dir_stride >= 3");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && ((uintptr_t)raw & 15) == 0, "bad ld / raw alignment");
    hipStream_t stream = (hipStream_t)stream_;
    if (!(n_hidden_density == 1 && n_hidden_color == 2))
        return launch_fwd_deep(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, raw, stream);
    const int cus = xr_device_cus() > 0 ? xr_device_cus() : 256;
    const uint32_t grid2 = min(xr_div_up((n + 31) / 32, BX_WAVES), 2u * (uint32_t)cus);   // 56 KiB of weights: two workgroups per CU
    const size_t lds = (size_t)2 * (HShape<1>::f_halves + (dirs ? HShape<2>::f_halves : 0)) * 2;
    auto k = dirs ? (g_range_word ? k_nerf_mlp_fwd_h2<true, true> : k_nerf_mlp_fwd_h2<true, false>)
                  : (g_range_word ? k_nerf_mlp_fwd_h2<false, true> : k_nerf_mlp_fwd_h2<false, false>);
    if (mlp_set_lds((const void*)k, lds) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
    hipLaunchKernelGGL(k, dim3(grid2), dim3(BX_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, pad_value, (float4*)raw,
                       dirs ? (const int32_t*)nullptr : g_fwd_splat_idx, dirs ? (float*)nullptr : g_fwd_splat_grid, g_range_word);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

static int mlp_bwd_f16(const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                                   const uint32_t* n_dev, const float* w_density, const float* w_color, int n_hidden_density,
                                   int n_hidden_color, float pad_value, const float* draw, float* denc_t, float* grad_w_density,
                                   float* grad_w_color, void* workspace, size_t workspace_bytes, const uint32_t* live_rows,
        const uint32_t* n_live, void* stream_) {
    if (n == 0) return XR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(enc_t && dirs && w_density && w_color && draw && denc_t && grad_w_density && grad_w_color, "null pointer");
    XR_REQUIRE(ld >= n && ld <= XR_MLP_MAX_LD && dir_stride >= 3 && ((uintptr_t)draw & 15) == 0, "bad ld / stride / alignment");
    XR_REQUIRE(n_hidden_density == 1 && n_hidden_color == 2, "the fp16 mode is built for the (1,2) hidden-layer topology");
    XR_REQUIRE(workspace && workspace_bytes >= xr_nerf_mlp_bwd_workspace_bytes(n, 1, 2), "workspace too small");
    constexpr int GW = NetShape<1>::glb_floats + NetShape<2>::glb_floats;
    constexpr size_t stage_bytes = (size_t)MLP_WAVES * 4 * 32 * HST * 2;
    static_assert(stage_bytes >= GW * sizeof(float), "stage area doubles as the dW reduction buffer");
    const size_t lds = (size_t)(HShape<1>::f_halves + HShape<2>::f_halves + HShape<1>::b_halves + HShape<2>::b_halves) * 2 + stage_bytes;
    const uint32_t grid = bwd_grid(n);
    XR_REQUIRE(!live_rows == !n_live, "live_rows and n_live come together");
    const uint32_t* rows = live_rows;                    // the caller's list (xr_live_rows), else the backward's own
    if (!rows && live_rows_enabled()) { const int rc = build_live_rows(draw, n, n_dev, denc_t, ld, workspace, stream, &rows, &n_live); if (rc != XR_OK) return rc; }
    auto kern = rows ? k_nerf_mlp_bwd_h<true> : k_nerf_mlp_bwd_h<false>;
    if (mlp_set_lds((const void*)kern, lds) != XR_OK) { xr_set_error("hipFuncSetAttribute failed"); return XR_EHIP; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_THREADS), lds, stream, enc_t, ld, dirs, dir_stride, n, n_dev, w_density,
                       w_color, pad_value, (const float4*)draw, denc_t, bwd_partials(workspace, n), rows, n_live);
    if (!g_defer_reduce)
        hipLaunchKernelGGL(k_reduce_partials, dim3(xr_div_up(GW, 64)), dim3(256), 0, stream, (const float*)bwd_partials(workspace, n), grid, (uint32_t)GW,
                           (uint32_t)NetShape<1>::glb_floats, grad_w_density, grad_w_color);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ---- the two entry points: `arithmetic` selects the kernel family (include/xrnerf_mi355.h: XR_MLP_F32 / _F16 / _BF16X3 / _F16X2)
extern "C" int xr_nerf_mlp_fwd(int arithmetic, const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                               const uint32_t* n_dev, const uint32_t* rows, const float* w_density, const float* w_color, int n_hidden_density,
                               int n_hidden_color, float pad_value, float* raw, void* stream) {
    XR_REQUIRE(arithmetic >= XR_MLP_F32 && arithmetic <= XR_MLP_F16X2, "unknown arithmetic");
    auto fwd = arithmetic == XR_MLP_F16 ? mlp_fwd_f16 : arithmetic == XR_MLP_BF16X3 ? mlp_fwd_bf16x3 : arithmetic == XR_MLP_F16X2 ? mlp_fwd_f16x2 : mlp_fwd_f32;
    return fwd(enc_t, ld, dirs, dir_stride, n, n_dev, rows, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, raw, stream);
}
extern "C" int xr_nerf_mlp_bwd(int arithmetic, const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                               const uint32_t* n_dev, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                               float pad_value, const float* draw, float* denc_t, float* grad_w_density, float* grad_w_color, void* workspace,
                               size_t workspace_bytes, const uint32_t* live_rows, const uint32_t* n_live, void* stream) {
    XR_REQUIRE(arithmetic >= XR_MLP_F32 && arithmetic <= XR_MLP_F16X2, "unknown arithmetic");
    auto bwd = arithmetic == XR_MLP_F16 ? mlp_bwd_f16 : mlp_bwd_f32;          // (the three fp32 forwards share one backward: XR_MLP_BWD_DW)
    return bwd(enc_t, ld, dirs, dir_stride, n, n_dev, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, draw, denc_t, grad_w_density,
               grad_w_color, workspace, workspace_bytes, live_rows, n_live, stream);
}

// K9's density query and K8 in one launch (grid refresh, ngp_grid_sampler.py:103-137 -> hashnerf_mlp.py:107-111 + splat_grid_samples...cu):
// the density network over n encoded points (enc_t feature-major, as xr_hashgrid_fwd writes it), each result's optical thickness
// exp(density) * min_step merged into density_grid_tmp[indices[i]] by an order-free maximum from the forward kernel's epilogue -- no
// [n,4] network output through HBM, no separate 2^20-thread launch.  mlp_mode as in xr_ngp_train_step; topology (1, 2) only.
extern "C" int xr_nerf_density_splat(int mlp_mode, const float* enc_t, uint32_t ld, uint32_t n, const float* w_density, int n_hidden_density,
                                     int n_hidden_color, const int32_t* indices, float* density_grid_tmp, void* stream) {
    if (n == 0) return XR_OK;
    XR_REQUIRE(indices && density_grid_tmp, "null pointer");
    XR_REQUIRE(mlp_mode != 1 || (n_hidden_density == 1 && n_hidden_color == 2), "the fp16 mode is built for the (1,2) hidden-layer topology");
    g_fwd_splat_idx = indices; g_fwd_splat_grid = density_grid_tmp;
    // (`raw` is not written in this mode; the argument only has to pass the alignment check)
    const int rc = xr_nerf_mlp_fwd(mlp_mode, enc_t, ld, nullptr, 0, n, nullptr, nullptr, w_density, nullptr, n_hidden_density, n_hidden_color, 1.0f,
                                   (float*)(((uintptr_t)density_grid_tmp + 15) & ~(uintptr_t)15), stream);
    g_fwd_splat_idx = nullptr; g_fwd_splat_grid = nullptr;
    return rc;
}
