// Small work of a training step that only has to be finished by the end of the table scatter: the fixed-order sum of the MLP
// backward's per-workgroup partials (+ the optimiser update of the two MLP tensors), the two loss scalars, a clear of the live-row
// segment counts.  Rounds 2-5 ran it as four launches on a helper stream forked from and joined into the step's stream -- a fork
// and a join cost the step's queue ~26 us per iteration (profiles/r04_event_cost_probe.txt, r06_trace_single_stream.txt).  Since
// round 6 it rides INSIDE the scatter's binning launch as a few extra workgroups (xr_scatter.hip): one in-order stream, no event.
//
// The device functions below are the one definition of that arithmetic: the stand-alone kernels (k_reduce_partials in xr_mlp.hip,
// k_train_loss_scalars in xr_raymarch.hip, k_adam_multi's update through adam1 / ema1) and the extra workgroups produce the same
// bits, whatever the number of real threads -- the work is laid out on VIRTUAL workgroups of the stand-alone kernels' shapes.
#pragma once
#include "xr_common.h"
#include "xr_adam.h"

struct XrAuxWork {
    // (1) grad[j] (+)= sum_b partial[b][j], j < gw; columns below `split` go to g0, the others to g1
    const float* partial; uint32_t nb, gw, split; float* g0; float* g1; int overwrite;
    // ... followed by the optimiser update of the two tensors where adam != 0 (a0: the tensor behind g0, a1: behind g1)
    int adam; XrAdamArgs a0, a1;
    // (2) out[0] = scale * sum Huber(rgb - target), out[1] = sum ((rgb - target) * alpha)^2      (rgb == nullptr: none)
    const float *rgb, *target, *alpha; uint32_t n_rays; float delta, scale; float* loss;
    // (3) words set to zero                                                                         (clear == nullptr: none)
    uint32_t* clear; uint32_t clear_words;
    // host side: set by the launch that took the work in
    bool done;
};
#define XR_AUX_RED_COLS 64            // columns per virtual reduce workgroup (256 threads = 64 columns x 4 row groups)
#define XR_AUX_LS_VTHREADS 1024       // virtual threads of the loss-scalar sum

// number of REAL-thread workgroups of (1)
template <int REAL> __host__ __device__ inline uint32_t xr_aux_reduce_blocks(uint32_t gw) {
    return (gw + XR_AUX_RED_COLS * (REAL / 256) - 1) / (XR_AUX_RED_COLS * (REAL / 256));
}

// fixed order: each of the 4 row groups of a column sums every 4th partial (16 loads in flight), the group sums are added as
// (0 + 1) + (2 + 3).  red: REAL / 64 rows of 64 floats of LDS.
template <int REAL>
__device__ __forceinline__ void xr_aux_reduce_block(const XrAuxWork& w, uint32_t blk, float (*red)[64]) {
    static_assert(REAL % 256 == 0, "whole virtual workgroups");
    const uint32_t c = threadIdx.x & 63u, rg = (threadIdx.x >> 6) & 3u, vb = threadIdx.x >> 8;
    const uint32_t j = (blk * (REAL / 256) + vb) * XR_AUX_RED_COLS + c, gw = w.gw, nb = w.nb;
    const float* __restrict__ partial = w.partial;
    float s = 0.f;
    if (j < gw) {
        uint32_t b = rg;
        for (; b + 4 * 15 < nb; b += 4 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(b + 4 * u) * gw + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; b < nb; b += 4) s += partial[(size_t)b * gw + j];
    }
    red[threadIdx.x >> 6][c] = s;
    __syncthreads();
    if (rg == 0 && j < gw) {
        const float (*r)[64] = red + 4 * vb;
        const float t = (r[0][c] + r[1][c]) + (r[2][c] + r[3][c]);
        const bool first = j < w.split;
        float* __restrict__ g = first ? w.g0 : w.g1;
        const uint32_t i = first ? j : j - w.split;
        // overwrite: the sums REPLACE what the buffers held (xr_ngp_train_step: no zero-fill of its gradient buffers)
        const float gv = w.overwrite ? t : g[i] + t;
        g[i] = gv;
        if (w.adam) {                                  // k_adam_multi's update of this parameter (same adam1 / ema1, same operands)
            const XrAdamArgs& A = first ? w.a0 : w.a1;
            float p = A.p[i], m = A.m[i], v = A.v[i];
            adam1(p, gv, m, v, A.b1, A.b2, A.step_size, A.bc2s, A.eps, A.wd, A.gs);
            A.p[i] = p; A.m[i] = m; A.v[i] = v;
            if (A.ema) A.ema[i] = ema1(A.ema[i], p, A.mom);
        }
    }
}

// ONE workgroup: virtual thread vt sums rays vt, vt + 1024, ... in that order; the 64 virtual threads of a virtual wave are combined
// by the xor butterfly, the 16 wave sums in index order.  ws, ws2: 16 floats of LDS each.
template <int REAL>
__device__ __forceinline__ void xr_aux_loss_block(const float* __restrict__ rgb, const float* __restrict__ target,
                                                  const float* __restrict__ alpha_mask, uint32_t n_rays, float delta, float scale,
                                                  float* __restrict__ out, float* ws, float* ws2) {
    constexpr int V = XR_AUX_LS_VTHREADS / REAL;
    static_assert(V >= 1 && V * REAL == XR_AUX_LS_VTHREADS && REAL % 64 == 0, "real threads divide the virtual ones");
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float acc = 0.f, mse = 0.f;
        // four rays' loads in flight per virtual thread (the sums keep their order: a plain loop exposed one memory latency per ray)
        for (uint32_t i0 = threadIdx.x + REAL * v; i0 < n_rays; i0 += XR_AUX_LS_VTHREADS * 4) {
            float am[4], r[4][3], t[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = min(i0 + XR_AUX_LS_VTHREADS * u, n_rays - 1);
                am[u] = alpha_mask[i];
#pragma unroll
                for (int c = 0; c < 3; ++c) { r[u][c] = rgb[3 * (size_t)i + c]; t[u][c] = target[3 * (size_t)i + c]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + XR_AUX_LS_VTHREADS * u >= n_rays) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float d = r[u][c] - t[u][c], a = fabsf(d);
                    acc += a > delta ? a - 0.5f * delta : 0.5f / delta * a * a;
                    const float mm = d * am[u]; mse += mm * mm;
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); mse += __shfl_xor(mse, d, 64); }
        if ((threadIdx.x & 63) == 0) { ws[(threadIdx.x >> 6) + (REAL / 64) * v] = acc; ws2[(threadIdx.x >> 6) + (REAL / 64) * v] = mse; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < XR_AUX_LS_VTHREADS / 64; ++k) { a += ws[k]; b += ws2[k]; }
        out[0] = scale * a; out[1] = b;
    }
}

// The work as workgroups of REAL threads: xr_aux_reduce_blocks<REAL>(gw) of them for (1) (short: they fit under a ~20-us launch), ONE
// for (2) + (3) (a ~25-us chain of dependent loads: it belongs into a longer launch).  lds: at least (REAL / 64) * 64 floats.
template <int REAL>
__device__ __forceinline__ void xr_aux_tail_block(const XrAuxWork& w, float* lds) {
    if (w.clear) for (uint32_t q = threadIdx.x; q < w.clear_words; q += REAL) w.clear[q] = 0u;
    if (w.rgb) xr_aux_loss_block<REAL>(w.rgb, w.target, w.alpha, w.n_rays, w.delta, w.scale, w.loss, lds, lds + 16);
}

// the next xr_scatter3 call of this thread takes the work into its binning launch where it has one (w->done tells); nullptr clears
void xr_internal_scatter_aux_work(XrAuxWork* w);
// the MLP backward's partial sums as (1) of an XrAuxWork: xr_mlp.hip
int xr_internal_mlp_bwd_reduce_desc(const void* workspace, uint32_t n, int n_hidden_density, int n_hidden_color, XrAuxWork* w);
