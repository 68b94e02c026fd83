// KiloNeRF rendering path for gfx950 (BASELINE config #5): thousands of 32-wide MLPs on a
// regular grid, each sample evaluated by the network of its cell.
//
// Reference (relative to /root/reference/): xrnerf/models/mlps/kilonerf_mlp.py:138-190 (KiloNerfMLP.forward) =
//   reorder_points_and_dirs (networks/utils/transforms.py:57-151: torch index arithmetic, nonzero, sort,
//   unique_consecutive, 4 gathers) -> kilonerf_cuda.global_to_local -> kilonerf_cuda.compute_fourier_features ->
//   6 x kilonerf_cuda.multimatmul_magma_grouped_static (MAGMA grouped GEMMs, activations through HBM between
//   layers) -> two index_put scatters; then NerfRender (renders/nerf_render.py:45-98).  `kilonerf_cuda` is an external
//   CUDA library that is not in the tree; the semantics are the in-tree PyTorch statements (oracle/kilo_oracle.py).
//
// Here:
//   k_kilo_assign    thread = sample: cell / occupancy / domain tests, network id per sample, per-network counts
//                    through an LDS histogram per workgroup (one global atomic per non-empty bin and workgroup);
//                    inactive samples get their zero raw row in the same pass (no separate fill)
//   k_kilo_offsets   one workgroup: exclusive scans -> segment start per network, tile start per network
//   k_kilo_scatter   counting-sort scatter with workgroup-level range reservation (LDS ranks)
//   k_kilo_mlp       persistent workgroups over (network, 128-sample tile): the network's 25 KB of weights are staged
//                    in LDS once per tile; a wave owns 32 samples and runs every layer as v_mfma_f32_32x32x2_f32 steps
//                    (neurons x samples) with the activations kept in the accumulator registers between layers; local
//                    coordinates and Fourier features are computed in registers (each half-wave evaluates half of the
//                    frequencies); the result goes straight to raw[sample]
//   k_nerf_render    wave = ray: classic NeRF compositing with an fp64 wave product scan
// Bound: fp32 MFMA (157.3 TFLOP/s; 12.2 kflop per evaluated sample + 3 padded K slots per Fourier group) for
// k_kilo_mlp, HBM for the rest.
#include "xr_common.h"
#include "xr_mip_math.h"      // xr_mip_zval: GetZvals' linspace (datasets/pipelines/create.py:502-516)

#define KILO_H 32            // hidden width == direction-layer width of every reference config
#define KILO_TILE 128           // samples per workgroup tile: 4 waves x 32 samples
#define KILO_MAX_LDS_BINS 16384u

struct KiloGrid {
    float gmin[3], gmax[3];
    int fixed_res[3], occ_res[3];
    // wave-uniform values of transforms.py:78-110, evaluated once on the host in the same fp32 operations
    float lo_eps[3], hi_eps[3];      // gmin + 0.001, gmax - 0.001
    float voxel[3], ovoxel[3];       // (gmax - gmin) / fixed_res, (gmax - gmin) / occ_res
};

struct KiloRays {
    const float* pts;                // nullable: explicit sample positions [n,3]
    const float* rays_o; const float* rays_d; const float* z_vals; const float* viewdirs;
    uint32_t n_rays, n_s;
    // z_vals == NULL: the un-jittered GetZvals lattice between near[ray] and far[ray], evaluated where it is needed
    const float* near; const float* far; int lindisp;
};

__device__ inline float kilo_z(const KiloRays& r, uint64_t ray, uint32_t s) {
    return r.z_vals != nullptr ? r.z_vals[ray * r.n_s + s] : xr_mip_zval(r.near[ray], r.far[ray], r.n_s, s, r.lindisp);
}

__device__ inline void kilo_point(const KiloRays& r, uint64_t i, float p[3]) {
    if (r.pts != nullptr) {
        p[0] = r.pts[i * 3]; p[1] = r.pts[i * 3 + 1]; p[2] = r.pts[i * 3 + 2];
    } else {
        const uint64_t ray = i / r.n_s;
        const float z = kilo_z(r, ray, (uint32_t)(i - ray * r.n_s));
        for (int a = 0; a < 3; ++a) p[a] = r.rays_o[ray * 3 + a] + r.rays_d[ray * 3 + a] * z;   // GetPts: mul, then add
    }
}

// transforms.py:69-120 for one sample: network index or -1.  The domain test comes first: every other test is moot
// for a sample that fails it, and most samples of a frame do (no divisions for them).
__device__ inline int kilo_network_of(const KiloGrid& g, const float p[3], const uint8_t* __restrict__ occupancy,
                                      int num_networks) {
    bool inside = true;
    for (int a = 0; a < 3; ++a) inside = inside && (p[a] > g.lo_eps[a]) && (p[a] < g.hi_eps[a]);
    if (!inside) return -1;
    long net = 0, oflat = 0;
    for (int a = 0; a < 3; ++a) {
        const float rel = p[a] - g.gmin[a];
        const long idx = (long)(rel / g.voxel[a]);                      // tensor.to(long): truncation toward zero
        net = net * g.fixed_res[a] + idx;        // row-major strides [r1*r2, r2, 1]; == sum(idx * strides)
        if (occupancy != nullptr) {
            long oi = (long)(rel / g.ovoxel[a]);
            oi = oi < 0 ? 0 : (oi > g.occ_res[a] - 1 ? g.occ_res[a] - 1 : oi);
            oflat = oflat * g.occ_res[a] + oi;
        }
    }
    if (occupancy != nullptr && occupancy[oflat] == 0) return -1;
    if (net < 0 || net >= num_networks) return -1;
    return (int)net;
}

// NOTE on `net = net * res + idx`: with idx possibly negative on an axis this Horner form still equals
// sum(idx * strides) exactly (integer arithmetic), which is what the reference compares against [0, num_networks).

// The lattice passes (assign, scatter, render) visit either every sample of a stretch of the flattened [R*S] lattice
// (spans == NULL: the module-level path with its dense tensors) or, in the fused frame path, only each ray's SPAN:
// the index range of the un-jittered z lattice that can lie inside the global domain (slab test, widened by two samples
// per side; axes the ray is nearly parallel to do not constrain).  Samples outside a span fail the reference's domain
// test, so they are never evaluated, written or read -- about 9 of 10 lattice samples of a Lego frame.
template <typename F>
__device__ __forceinline__ void kilo_for_each_sample(const uint32_t* __restrict__ spans, uint32_t n_s, uint64_t n,
                                                     uint32_t per_block, uint32_t n_rays, F f) {
    if (spans == nullptr) {
        const uint64_t i0 = (uint64_t)blockIdx.x * per_block;
        const uint64_t i1 = i0 + per_block < n ? i0 + per_block : n;
        for (uint64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) f(i);
    } else {
        const uint32_t r0 = blockIdx.x * per_block;                  // per_block counts rays in this mode
        const uint32_t r1 = r0 + per_block < n_rays ? r0 + per_block : n_rays;
        const uint32_t lane = threadIdx.x & 63;
        for (uint32_t ray = r0 + (threadIdx.x >> 6); ray < r1; ray += blockDim.x >> 6) {
            const uint32_t lo = spans[2 * ray], hi = spans[2 * ray + 1];
            for (uint32_t s = lo + lane; s < hi; s += 64) f((uint64_t)ray * n_s + s);
        }
    }
}

__global__ void k_kilo_spans(KiloGrid g, KiloRays r, uint32_t* __restrict__ spans) {
    const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= r.n_rays) return;
    const float near = r.near[ray], far = r.far[ray];
    float z0 = fminf(near, far), z1 = fmaxf(near, far);
    bool empty = false;
    for (int a = 0; a < 3; ++a) {
        const float o = r.rays_o[ray * 3ull + a], d = r.rays_d[ray * 3ull + a];
        if (fabsf(d) >= 1e-3f) {
            const float za = (g.lo_eps[a] - o) / d, zb = (g.hi_eps[a] - o) / d;
            z0 = fmaxf(z0, fminf(za, zb));
            z1 = fminf(z1, fmaxf(za, zb));
        } else if (fabsf(d) == 0.f && (o <= g.lo_eps[a] - 1e-3f || o >= g.hi_eps[a] + 1e-3f)) {
            empty = true;                               // parallel to the slab and clearly outside of it
        }
    }
    uint32_t lo = 0, hi = 0;
    if (!empty && z0 <= z1 && far != near && r.lindisp == 0) {
        const float scale = (float)(r.n_s - 1) / (far - near);
        float a = (z0 - near) * scale, b = (z1 - near) * scale;
        if (a > b) { const float t = a; a = b; b = t; }
        const float fl = floorf(a) - 2.f, fh = ceilf(b) + 3.f;
        lo = fl <= 0.f ? 0u : (fl >= (float)r.n_s ? r.n_s : (uint32_t)fl);
        hi = fh <= 0.f ? 0u : (fh >= (float)r.n_s ? r.n_s : (uint32_t)fh);
        if (hi < lo) hi = lo;
    } else if (!empty && (far == near || r.lindisp != 0)) {
        hi = r.n_s;                                     // no closed form used: keep the whole ray
    }
    spans[2 * ray] = lo; spans[2 * ray + 1] = hi;
}

__global__ void __launch_bounds__(1024) k_kilo_assign(KiloGrid g, KiloRays r, const uint8_t* __restrict__ occupancy,
                                                     uint32_t num_networks, uint64_t n, uint32_t per_block,
                                                     const uint32_t* __restrict__ spans, int32_t* __restrict__ net_of,
                                                     uint32_t* __restrict__ counts,
                                                     float4* __restrict__ raw /* NULL: rows without a network stay unwritten */) {
    extern __shared__ uint32_t s_hist[];
    const bool lds = num_networks <= KILO_MAX_LDS_BINS;
    if (lds) {
        for (uint32_t b = threadIdx.x; b < num_networks; b += blockDim.x) s_hist[b] = 0;
        __syncthreads();
    }
    int any = 0;
    kilo_for_each_sample(spans, r.n_s, n, per_block, r.n_rays, [&](uint64_t i) {
        float p[3];
        kilo_point(r, i, p);
        const int net = kilo_network_of(g, p, occupancy, (int)num_networks);
        net_of[i] = net;
        if (net < 0) { if (raw != nullptr) raw[i] = make_float4(0.f, 0.f, 0.f, 0.f); }   // kilonerf_mlp.py:183-189: zeros elsewhere
        else if (lds) { atomicAdd(&s_hist[net], 1u); any = 1; }
        else atomicAdd(&counts[net], 1u);
    });
    if (lds && __syncthreads_or(any)) {                 // most workgroups of a frame see no occupied sample at all
        for (uint32_t b = threadIdx.x; b < num_networks; b += blockDim.x)
            if (s_hist[b]) atomicAdd(&counts[b], s_hist[b]);
    }
}

// seg_start[N+1] = exclusive scan of counts; tile_start[N+1] = exclusive scan of ceil(counts / TILE); cursor = 0
__global__ void __launch_bounds__(1024) k_kilo_offsets(const uint32_t* __restrict__ counts, uint32_t num_networks,
                                                       uint32_t* __restrict__ seg_start, uint32_t* __restrict__ tile_start,
                                                       uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s_a[1024], s_b[1024];
    __shared__ uint32_t carry_a, carry_b;
    if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < num_networks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < num_networks ? counts[i] : 0u;
        s_a[threadIdx.x] = c;
        s_b[threadIdx.x] = (c + KILO_TILE - 1) / KILO_TILE;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {                  // Hillis-Steele inclusive scan
            uint32_t va = 0, vb = 0;
            if (threadIdx.x >= off) { va = s_a[threadIdx.x - off]; vb = s_b[threadIdx.x - off]; }
            __syncthreads();
            s_a[threadIdx.x] += va; s_b[threadIdx.x] += vb;
            __syncthreads();
        }
        if (i < num_networks) {
            seg_start[i] = carry_a + s_a[threadIdx.x] - c;
            tile_start[i] = carry_b + s_b[threadIdx.x] - (c + KILO_TILE - 1) / KILO_TILE;
            cursor[i] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_a += s_a[1023]; carry_b += s_b[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { seg_start[num_networks] = carry_a; tile_start[num_networks] = carry_b; }
}

__global__ void __launch_bounds__(1024) k_kilo_scatter(const int32_t* __restrict__ net_of, uint32_t num_networks, uint64_t n,
                                                      uint32_t per_block, const uint32_t* __restrict__ spans, uint32_t n_s,
                                                      uint32_t n_rays, const uint32_t* __restrict__ seg_start,
                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
    extern __shared__ uint32_t s_hist[];
    const bool lds = num_networks <= KILO_MAX_LDS_BINS;
    if (!lds) {
        kilo_for_each_sample(spans, n_s, n, per_block, n_rays, [&](uint64_t i) {
            const int net = net_of[i];
            if (net >= 0) order[seg_start[net] + atomicAdd(&cursor[net], 1u)] = (uint32_t)i;
        });
        return;
    }
    for (uint32_t b = threadIdx.x; b < num_networks; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    int any = 0;
    kilo_for_each_sample(spans, n_s, n, per_block, n_rays, [&](uint64_t i) {
        const int net = net_of[i];
        if (net >= 0) { atomicAdd(&s_hist[net], 1u); any = 1; }
    });
    if (!__syncthreads_or(any)) return;                 // nothing to place from this stretch of samples
    // reserve this workgroup's range inside every network segment it touches; the bin then holds the next free slot
    for (uint32_t b = threadIdx.x; b < num_networks; b += blockDim.x) {
        const uint32_t c = s_hist[b];
        if (c) s_hist[b] = seg_start[b] + atomicAdd(&cursor[b], c);
    }
    __syncthreads();
    kilo_for_each_sample(spans, n_s, n, per_block, n_rays, [&](uint64_t i) {
        const int net = net_of[i];
        if (net >= 0) order[atomicAdd(&s_hist[net], 1u)] = (uint32_t)i;
    });
}

// ------------------------------------------------------------------------------------------ the tiny MLPs
// packed parameter block of one network (floats), H = 32, P = 3(2Fp+1), D = 3(2Fd+1), L hidden layers:
//   [W_0 (P x H, input-major) | b_0 (H)] [W_l (H x H) | b_l (H)] l=1..L-1
//   [W_alpha (H) | b_alpha (1) + 3 pad] [W_feature (H x H) | b_feature (H)]
//   [W_dir ((H + D) x H) | b_dir (H)] [W_rgb (H x 4, 4th column zero) | b_rgb (4)]
__host__ __device__ inline uint32_t kilo_param_floats(int pos_freqs, int dir_freqs, int n_hidden) {
    const uint32_t P = 3u * (2u * pos_freqs + 1u), D = 3u * (2u * dir_freqs + 1u), H = KILO_H;
    return (P * H + H) + (uint32_t)(n_hidden - 1) * (H * H + H) + (H + 4u) + (H * H + H) + ((H + D) * H + H) + (H * 4u + 4u);
}

struct KiloMlpArgs {
    KiloRays rays;
    const float* domain_mins; const float* domain_maxs;        // [N,3]
    const float* params; uint32_t param_stride;               // [N, stride]
    uint32_t num_networks;
    int pos_freqs, dir_freqs, n_hidden;
    const uint32_t* seg_start; const uint32_t* tile_start; const uint32_t* order;
    float4* raw;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_32x32x2_f32 (D = A.B + C, M = 32 output neurons, N = 32 samples, K = 2 inputs per instruction):
//   A: lane l holds A[m = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][n = l & 31]
//   C/D: lane l, register r holds D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31]
// A wave owns 32 samples (column = l & 31); the two half-waves (hi = l >> 5) split the K dimension.  The order of the
// K terms is free, so a layer's output registers are fed straight back as the next layer's B operand: at step s the
// lower half supplies neuron drow(s), the upper half neuron drow(s) + 4 -- exactly what each holds in register s --
// and the A operand is the matching weight row.  Activations never leave registers; a weight is read from LDS once
// per 32 samples (one ds_read_b32 per MFMA = 16 flop per LDS byte; the scalar-FMA version of this kernel with
// broadcast ds_read_b128 operands was LDS-bound at 0.5 flop per byte: 11.0 ms per frame instead of this one's time,
// profiles/r01_kilo_*).
__device__ __forceinline__ constexpr int kdrow(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ f32x16 kilo_bias(const float* __restrict__ b, int hi) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = b[kdrow(r) + 4 * hi];
    return v;
}

// acc += W^T h for a [32 in][32 out] slab (input-major rows of 32 floats)
__device__ __forceinline__ f32x16 kilo_dense32(const float* __restrict__ w, f32x16 acc, const f32x16& h, int col, int hi) {
    const float* wl = w + (4 * hi) * KILO_H + col;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = KMFMA(wl[kdrow(s) * KILO_H], h[s], acc);
    return acc;
}

__device__ __forceinline__ f32x16 kilo_relu(f32x16 v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
    return v;
}

// Fourier features of one scalar through a [(2F+1) x 32] weight slab (rows: x | cos(x 2^k), k < F | sin(x 2^k), k < F;
// kilonerf_fourier_embedder.py:33-52).  The lower half-wave feeds x and the even frequencies, the upper half the odd
// ones (one sincos per lane and frequency pair); slots without a feature carry b = 0.
__device__ __forceinline__ f32x16 kilo_feed_fourier(const float* __restrict__ w, f32x16 acc, float x, int n_freq, int col, int hi) {
    acc = KMFMA(w[col], hi == 0 ? x : 0.f, acc);
    for (int m = 0; 2 * m < n_freq; ++m) {
        const int k = 2 * m + hi;
        const bool valid = k < n_freq;
        float sv, cv;
        sincosf(ldexpf(x, k), &sv, &cv);                 // x * 2^k is exact
        const int kk = valid ? k : 0;
        acc = KMFMA(w[(1 + kk) * KILO_H + col], valid ? cv : 0.f, acc);
        acc = KMFMA(w[(1 + n_freq + kk) * KILO_H + col], valid ? sv : 0.f, acc);
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_kilo_mlp(KiloMlpArgs a) {
    extern __shared__ float s_w[];                      // one network's parameter block
    __shared__ uint32_t s_net;
    const uint32_t n_floats = kilo_param_floats(a.pos_freqs, a.dir_freqs, a.n_hidden);
    const uint32_t total_tiles = a.tile_start[a.num_networks];
    const uint32_t P = 3u * (2u * a.pos_freqs + 1u), D = 3u * (2u * a.dir_freqs + 1u);
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        __syncthreads();                                // previous tile's weights are no longer read
        if (threadIdx.x == 0) {
            // last network whose tile_start <= tile (networks without samples own no tile: skip them)
            uint32_t lo = 0, hi_n = a.num_networks - 1;
            while (lo < hi_n) {
                const uint32_t mid = (lo + hi_n + 1) >> 1;
                if (a.tile_start[mid] <= tile) lo = mid; else hi_n = mid - 1;
            }
            s_net = lo;
        }
        __syncthreads();
        const uint32_t net = s_net;
        {
            const float4* src = reinterpret_cast<const float4*>(a.params + (size_t)net * a.param_stride);
            float4* dst = reinterpret_cast<float4*>(s_w);
            for (uint32_t q = threadIdx.x; q < n_floats / 4; q += 256) dst[q] = src[q];
        }
        __syncthreads();
        const uint32_t seg0 = a.seg_start[net], seg1 = a.seg_start[net + 1];
        const uint32_t wave0 = seg0 + (tile - a.tile_start[net]) * KILO_TILE + wave * 32;
        if (wave0 >= seg1) continue;                    // this wave's 32 slots are past the segment (uniform per wave)
        const uint32_t slot = wave0 + col;
        const bool live = slot < seg1;
        const uint32_t i = a.order[live ? slot : seg1 - 1];
        float p[3];
        kilo_point(a.rays, i, p);
        const float* w = s_w;
        // layer 0 on the Fourier features of the local coordinates (transforms.py:35-45: 2 (p - min) / (max - min) - 1)
        f32x16 acc = kilo_bias(w + P * KILO_H, hi);
        for (int c = 0; c < 3; ++c) {
            const float lo = a.domain_mins[net * 3 + c], hi_d = a.domain_maxs[net * 3 + c];
            const float x = 2.f * (p[c] - lo) / (hi_d - lo) - 1.f;
            acc = kilo_feed_fourier(w + c * (2 * a.pos_freqs + 1) * KILO_H, acc, x, a.pos_freqs, col, hi);
        }
        w += P * KILO_H + KILO_H;
        f32x16 h = kilo_relu(acc);
        for (int l = 1; l < a.n_hidden; ++l) {
            h = kilo_relu(kilo_dense32(w, kilo_bias(w + KILO_H * KILO_H, hi), h, col, hi));
            w += KILO_H * KILO_H + KILO_H;
        }
        // alpha: no activation (multi_modules.py:609-610); each half-wave holds 16 of the 32 hidden values
        float alpha = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) alpha = fmaf(w[kdrow(r) + 4 * hi], h[r], alpha);
        alpha = alpha + __shfl_xor(alpha, 32, 64) + w[KILO_H];
        w += KILO_H + 4;
        // feature vector: no activation (multi_modules.py:611-615)
        h = kilo_dense32(w, kilo_bias(w + KILO_H * KILO_H, hi), h, col, hi);
        w += KILO_H * KILO_H + KILO_H;
        // direction layer on [feature | Fourier(viewdir)]
        acc = kilo_dense32(w, kilo_bias(w + (KILO_H + D) * KILO_H, hi), h, col, hi);
        {
            const uint64_t ray = (uint64_t)i / a.rays.n_s;
            for (int c = 0; c < 3; ++c)
                acc = kilo_feed_fourier(w + (KILO_H + c * (2 * a.dir_freqs + 1)) * KILO_H, acc, a.rays.viewdirs[ray * 3 + c],
                                        a.dir_freqs, col, hi);
        }
        w += (KILO_H + D) * KILO_H + KILO_H;
        h = kilo_relu(acc);
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 t = *reinterpret_cast<const float4*>(w + (kdrow(r) + 4 * hi) * 4);
            o0 = fmaf(t.x, h[r], o0); o1 = fmaf(t.y, h[r], o1); o2 = fmaf(t.z, h[r], o2);
        }
        const float4 br = *reinterpret_cast<const float4*>(w + KILO_H * 4);
        o0 = o0 + __shfl_xor(o0, 32, 64) + br.x;
        o1 = o1 + __shfl_xor(o1, 32, 64) + br.y;
        o2 = o2 + __shfl_xor(o2, 32, 64) + br.z;
        if (live && hi == 0) a.raw[i] = make_float4(o0, o1, o2, alpha);
    }
}

// ------------------------------------------------------------------------------------------ fine-tuning gradients
// dL/d(parameters) of the tiny MLPs given dL/draw (AddMultiMatMul.backward, multi_modules.py:215-236 = per network
// grad_biases = column sums, grad_weights = in^T . grad_out, grad_in = grad_out . W^T; chained through
// MultiNetwork.forward :590-668).  Same tiling as the forward: a wave owns 32 samples in the neurons x samples MFMA
// layout and first re-runs the forward (nothing is saved).  Back-propagation through a layer is the forward's MFMA idiom
// with the weight slab read transposed; a weight gradient is an outer product contracted over the wave's 32 samples --
// both operand tiles ([neurons][samples]) go through a per-wave LDS tile so that the sample index can be the MFMA's K --
// and is added to the network's block of the packed gradient buffer with coalesced fp32 atomics (32 consecutive floats
// per row).  Biases: half-wave butterfly sums.  n_hidden <= 2, pos_freqs <= 10, dir_freqs <= 4 (every reference config).
#define KB_ST 33                                 // LDS tile row stride (floats): conflict-free for row- and column-wise access
#define KB_WAVE_FLOATS ((64 + 32) * KB_ST)       // per wave: input tile (64 rows) + delta tile (32 rows)

struct KiloBwdArgs { KiloMlpArgs f; const float4* draw; float* grad; };

__device__ __forceinline__ void ktile_to_lds(float* __restrict__ T, const f32x16& v, int col, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(kdrow(r) + 4 * hi) * KB_ST + col] = v[r];
}
// D[k][j] = sum_n tin[k][n] * tdl[j][n]  (n = the wave's 32 samples)
__device__ __forceinline__ f32x16 kouter(const float* __restrict__ tin, const float* __restrict__ tdl, int col, int hi) {
    // the tiles were written by the other lanes of this wave: lockstep execution orders those stores before the loads
    // below; the barrier pins that order for the compiler (and is the rendezvous of the host emulation, tests/hip_emu)
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = KMFMA(tin[col * KB_ST + 2 * s + hi], tdl[col * KB_ST + 2 * s + hi], acc);
    return acc;
}
__device__ __forceinline__ void ktile_atomic_add(float* __restrict__ g, const f32x16& d, int ldw, int n_rows, int n_cols, int col, int hi) {
    if (col < n_cols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kdrow(r) + 4 * hi;
            if (k < n_rows) atomicAdd(g + k * ldw + col, d[r]);
        }
    }
}
__device__ __forceinline__ float khalf_sum(float v) {           // over the 32 lanes (samples) of a half-wave
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// gb[neuron] += sum over the samples of delta[neuron][sample]
__device__ __forceinline__ void kbias_add(float* __restrict__ gb, const f32x16& d, int col, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float sum = khalf_sum(d[r]);
        if (col == 0) atomicAdd(gb + kdrow(r) + 4 * hi, sum);
    }
}
// delta_in[k][n] = sum_j W[k][j] delta_out[j][n] for an input-major [32 k][32 j] slab (read along j: transposed use)
__device__ __forceinline__ f32x16 kilo_dense32_T(const float* __restrict__ w, const f32x16& d, int col, int hi) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* wl = w + col * KILO_H + 4 * hi;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = KMFMA(wl[kdrow(s)], d[s], acc);
    return acc;
}
// Fourier features of one scalar into rows [0, 2F+1) of an LDS tile (row = feature, column = sample): same split between
// the half-waves as kilo_feed_fourier
__device__ __forceinline__ void kfourier_to_lds(float* __restrict__ T, float x, int n_freq, int col, int hi) {
    if (hi == 0) T[col] = x;
    for (int m = 0; 2 * m < n_freq; ++m) {
        const int k = 2 * m + hi;
        if (k < n_freq) {
            float sv, cv;
            sincosf(ldexpf(x, k), &sv, &cv);
            T[(1 + k) * KB_ST + col] = cv;
            T[(1 + n_freq + k) * KB_ST + col] = sv;
        }
    }
}

#define KB_WAVES 4                               // waves per workgroup = 32-sample parts of a 128-sample tile (round 6: four waves side by side
                                                 // instead of two waves taking two parts each in turn -- the kernel is one long dependent MFMA chain per wave)
template <int NH>
__global__ void __launch_bounds__(64 * KB_WAVES) k_kilo_mlp_bwd(KiloBwdArgs b) {
    extern __shared__ float s_w[];                      // [parameter block | per-wave tiles]
    __shared__ uint32_t s_net;
    const KiloMlpArgs& a = b.f;
    const uint32_t n_floats = kilo_param_floats(a.pos_freqs, a.dir_freqs, NH);
    const uint32_t total_tiles = a.tile_start[a.num_networks];
    const int P = 3 * (2 * a.pos_freqs + 1), D = 3 * (2 * a.dir_freqs + 1);
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const uint32_t wave = threadIdx.x >> 6;
    float* tin = s_w + n_floats + wave * KB_WAVE_FLOATS;
    float* tdl = tin + 64 * KB_ST;
    // offsets inside a parameter / gradient block
    const int o_b0 = P * KILO_H, o_l1 = o_b0 + KILO_H, o_alpha = o_l1 + (NH - 1) * (KILO_H * KILO_H + KILO_H);
    const int o_feat = o_alpha + KILO_H + 4, o_dir = o_feat + KILO_H * KILO_H + KILO_H;
    const int o_bd = o_dir + (KILO_H + D) * KILO_H, o_rgb = o_bd + KILO_H;
    for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t lo = 0, hi_n = a.num_networks - 1;
            while (lo < hi_n) {
                const uint32_t mid = (lo + hi_n + 1) >> 1;
                if (a.tile_start[mid] <= tile) lo = mid; else hi_n = mid - 1;
            }
            s_net = lo;
        }
        __syncthreads();
        const uint32_t net = s_net;
        {
            const float4* src = reinterpret_cast<const float4*>(a.params + (size_t)net * a.param_stride);
            float4* dst = reinterpret_cast<float4*>(s_w);
            for (uint32_t q = threadIdx.x; q < n_floats / 4; q += 64 * KB_WAVES) dst[q] = src[q];
        }
        __syncthreads();
        const float* w = s_w;
        float* g = b.grad + (size_t)net * a.param_stride;
        const uint32_t seg0 = a.seg_start[net], seg1 = a.seg_start[net + 1];
        static_assert(KB_WAVES * 32 == KILO_TILE, "one 32-sample part per wave");
        {
            const uint32_t wave0 = seg0 + (tile - a.tile_start[net]) * KILO_TILE + wave * 32;
            if (wave0 >= seg1) continue;                // (uniform per wave; the loop's barriers are at its top, every wave reaches them)
            const uint32_t slot = wave0 + col;
            const bool live = slot < seg1;
            const uint32_t i = a.order[live ? slot : seg1 - 1];
            float p[3], x[3], v[3];
            kilo_point(a.rays, i, p);
            const uint64_t ray = (uint64_t)i / a.rays.n_s;
            for (int c = 0; c < 3; ++c) {
                const float lo = a.domain_mins[net * 3 + c], hi_d = a.domain_maxs[net * 3 + c];
                x[c] = 2.f * (p[c] - lo) / (hi_d - lo) - 1.f;
                v[c] = a.rays.viewdirs[ray * 3 + c];
            }
            // ---- forward, as k_kilo_mlp
            f32x16 acc = kilo_bias(w + o_b0, hi);
            for (int c = 0; c < 3; ++c) acc = kilo_feed_fourier(w + c * (2 * a.pos_freqs + 1) * KILO_H, acc, x[c], a.pos_freqs, col, hi);
            const f32x16 h0 = kilo_relu(acc);
            f32x16 hl = h0;
            if (NH == 2) hl = kilo_relu(kilo_dense32(w + o_l1, kilo_bias(w + o_l1 + KILO_H * KILO_H, hi), h0, col, hi));
            const f32x16 feat = kilo_dense32(w + o_feat, kilo_bias(w + o_feat + KILO_H * KILO_H, hi), hl, col, hi);
            acc = kilo_dense32(w + o_dir, kilo_bias(w + o_bd, hi), feat, col, hi);
            for (int c = 0; c < 3; ++c)
                acc = kilo_feed_fourier(w + o_dir + (KILO_H + c * (2 * a.dir_freqs + 1)) * KILO_H, acc, v[c], a.dir_freqs, col, hi);
            const f32x16 hd = kilo_relu(acc);
            // ---- backward
            float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) gr = b.draw[i];
            // rgb head: out_c = b_c + sum_k Wr[k][c] hd[k]
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            if (hi == 0) { d[0] = gr.x; d[1] = gr.y; d[2] = gr.z; }          // rows 0..2 of the delta tile (kdrow(r) = r for r < 4)
            ktile_to_lds(tin, hd, col, hi);
            ktile_to_lds(tdl, d, col, hi);
            ktile_atomic_add(g + o_rgb, kouter(tin, tdl, col, hi), 4, KILO_H, 3, col, hi);
            {
                const float s0 = khalf_sum(gr.x), s1 = khalf_sum(gr.y), s2 = khalf_sum(gr.z), s3 = khalf_sum(gr.w);
                if (lane == 0) {
                    atomicAdd(g + o_rgb + KILO_H * 4 + 0, s0); atomicAdd(g + o_rgb + KILO_H * 4 + 1, s1);
                    atomicAdd(g + o_rgb + KILO_H * 4 + 2, s2);
                    atomicAdd(g + o_alpha + KILO_H, s3);                    // b_alpha
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 t = *reinterpret_cast<const float4*>(w + o_rgb + (kdrow(r) + 4 * hi) * 4);
                d[r] = hd[r] > 0.f ? fmaf(t.x, gr.x, fmaf(t.y, gr.y, t.z * gr.z)) : 0.f;
            }
            // direction layer: inputs [feature (32) | Fourier(viewdir) (D)]
            ktile_to_lds(tdl, d, col, hi);
            ktile_to_lds(tin, feat, col, hi);
            ktile_atomic_add(g + o_dir, kouter(tin, tdl, col, hi), KILO_H, KILO_H, KILO_H, col, hi);
            for (int rr = hi; rr < 32; rr += 2) tin[rr * KB_ST + col] = 0.f;     // rows beyond D stay zero
            __builtin_amdgcn_wave_barrier();                                   // zero fill (all lanes) before the feature rows
            for (int c = 0; c < 3; ++c) kfourier_to_lds(tin + c * (2 * a.dir_freqs + 1) * KB_ST, v[c], a.dir_freqs, col, hi);
            ktile_atomic_add(g + o_dir + KILO_H * KILO_H, kouter(tin, tdl, col, hi), KILO_H, D, KILO_H, col, hi);
            kbias_add(g + o_bd, d, col, hi);
            d = kilo_dense32_T(w + o_dir, d, col, hi);                      // dL/dfeature (no activation)
            // feature and alpha heads on hl
            ktile_to_lds(tdl, d, col, hi);
            ktile_to_lds(tin, hl, col, hi);
            ktile_atomic_add(g + o_feat, kouter(tin, tdl, col, hi), KILO_H, KILO_H, KILO_H, col, hi);
            kbias_add(g + o_feat + KILO_H * KILO_H, d, col, hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sum = khalf_sum(hl[r] * gr.w);
                if (col == 0) atomicAdd(g + o_alpha + kdrow(r) + 4 * hi, sum);
            }
            d = kilo_dense32_T(w + o_feat, d, col, hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = hl[r] > 0.f ? fmaf(w[o_alpha + kdrow(r) + 4 * hi], gr.w, d[r]) : 0.f;
            if (NH == 2) {
                ktile_to_lds(tdl, d, col, hi);
                ktile_to_lds(tin, h0, col, hi);
                ktile_atomic_add(g + o_l1, kouter(tin, tdl, col, hi), KILO_H, KILO_H, KILO_H, col, hi);
                kbias_add(g + o_l1 + KILO_H * KILO_H, d, col, hi);
                d = kilo_dense32_T(w + o_l1, d, col, hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = h0[r] > 0.f ? d[r] : 0.f;
            }
            // layer 0 on the Fourier features of the local coordinates: two row tiles of the [P x 32] gradient
            ktile_to_lds(tdl, d, col, hi);
            for (int rr = hi; rr < 64; rr += 2) tin[rr * KB_ST + col] = 0.f;
            __builtin_amdgcn_wave_barrier();
            for (int c = 0; c < 3; ++c) kfourier_to_lds(tin + c * (2 * a.pos_freqs + 1) * KB_ST, x[c], a.pos_freqs, col, hi);
            ktile_atomic_add(g, kouter(tin, tdl, col, hi), KILO_H, P < 32 ? P : 32, KILO_H, col, hi);
            if (P > 32) ktile_atomic_add(g + 32 * KILO_H, kouter(tin + 32 * KB_ST, tdl, col, hi), KILO_H, P - 32, KILO_H, col, hi);
            kbias_add(g + o_b0, d, col, hi);
        }
    }
}

// ------------------------------------------------------------------------------------------ NerfRender.forward
__device__ inline double wave_incl_prod(double v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off, 64);
        if (lane >= off) v *= o;
    }
    return v;
}
__device__ inline float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// renders/nerf_render.py:45-98, raw_noise_std = 0; z are sample POSITIONS; the last interval is 1e10.
// net_of (nullable): rows with net_of < 0 were never written and count as raw = 0 (exactly what they contribute: alpha = 0,
// weight = 0, transmittance factor 1) -- the sparse frame path reads 4 bytes per empty sample instead of 16.
__global__ void __launch_bounds__(256) k_nerf_render(const float4* __restrict__ raw, KiloRays zr,
                                                     const int32_t* __restrict__ net_of,
                                                     const uint32_t* __restrict__ spans /* with net_of: only [lo, hi) was assigned */,
                                                     int white_bkgd,
                                                     float* __restrict__ rgb_out, float* __restrict__ disp_out,
                                                     float* __restrict__ acc_out, float* __restrict__ weights_out) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_s = zr.n_s;
    if (r >= zr.n_rays) return;
    const float dx = zr.rays_d[r * 3ull], dy = zr.rays_d[r * 3ull + 1], dz = zr.rays_d[r * 3ull + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    double carry = 1.0;                                  // prod_{j < sweep} (1 - alpha_j + 1e-10)
    float a_c[3] = {0.f, 0.f, 0.f}, a_w = 0.f, a_z = 0.f;
    const uint32_t s_lo = spans != nullptr ? spans[2 * r] : 0u, s_hi = spans != nullptr ? spans[2 * r + 1] : n_s;
    for (uint32_t base = s_lo & ~63u; base < s_hi; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i >= s_lo && i < s_hi;
        float alpha = 0.f, zi = 0.f;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        double fac = 1.0;
        if (live) {
            const uint64_t g = (uint64_t)r * n_s + i;
            const bool evaluated = net_of == nullptr || net_of[g] >= 0;
            if (evaluated) {
                v = raw[g];
                zi = kilo_z(zr, r, i);
                const float dist = (i + 1 < n_s ? kilo_z(zr, r, i + 1) - zi : 1e10f) * dnorm;
                alpha = 1.f - expf(-(fmaxf(v.w, 0.f) * dist));
                fac = (double)(1.f - alpha + 1e-10f);
            }
        }
        if (__ballot(alpha != 0.f) == 0ull) {           // a sweep without density: factors are exactly 1 (1 - 0 + 1e-10)
            if (live && weights_out != nullptr) weights_out[(uint64_t)r * n_s + i] = 0.f;
            continue;
        }
        const double incl = wave_incl_prod(fac);
        // exclusive product without dividing (a factor can be exactly 0): shift the inclusive scan by one lane
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        if (live) {
            const float w = alpha * (float)(carry * excl);
            if (weights_out != nullptr) weights_out[(uint64_t)r * n_s + i] = w;
            a_w += w;
            a_z += w * zi;
            a_c[0] += w * (1.f / (1.f + expf(-v.x)));
            a_c[1] += w * (1.f / (1.f + expf(-v.y)));
            a_c[2] += w * (1.f / (1.f + expf(-v.z)));
        }
        carry *= __shfl(incl, 63, 64);
    }
    const float acc = wave_sum_f(a_w), depth = wave_sum_f(a_z);
    float col[3];
    for (int c = 0; c < 3; ++c) col[c] = wave_sum_f(a_c[c]);
    if (lane == 0) {
        const float q = depth / acc;
        const float m = (q != q) ? q : fmaxf(1e-10f, q);                 // torch.max propagates the NaN of 0/0
        disp_out[r] = 1.f / m;
        acc_out[r] = acc;
        for (int c = 0; c < 3; ++c) rgb_out[r * 3ull + c] = white_bkgd ? col[c] + (1.f - acc) : col[c];
    }
}

// ------------------------------------------------------------------------------------------ C-ABI
extern "C" uint32_t xr_kilo_param_floats(int pos_freqs, int dir_freqs, int n_hidden) {
    if (pos_freqs < 0 || dir_freqs < 0 || n_hidden < 1) return 0;
    return kilo_param_floats(pos_freqs, dir_freqs, n_hidden);
}

struct KiloWs { int32_t* net_of; uint32_t* order; uint32_t* counts; uint32_t* cursor; uint32_t* seg_start; uint32_t* tile_start; uint32_t* spans; };
static size_t kilo_ws_layout(uint64_t n, uint32_t N, uint32_t span_rays, char* base, KiloWs* ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? base + o : nullptr; };
    char* p;
    p = take(n * 4); if (ws) ws->net_of = (int32_t*)p;
    p = take(n * 4); if (ws) ws->order = (uint32_t*)p;
    p = take((size_t)N * 4); if (ws) ws->counts = (uint32_t*)p;
    p = take((size_t)N * 4); if (ws) ws->cursor = (uint32_t*)p;
    p = take((size_t)(N + 1) * 4); if (ws) ws->seg_start = (uint32_t*)p;
    p = take((size_t)(N + 1) * 4); if (ws) ws->tile_start = (uint32_t*)p;
    p = take((size_t)span_rays * 8); if (ws) ws->spans = span_rays ? (uint32_t*)p : nullptr;
    return off;
}

extern "C" size_t xr_kilo_workspace_bytes(uint64_t n_samples, uint32_t num_networks) {
    return kilo_ws_layout(n_samples, num_networks, 0, nullptr, nullptr);
}

// assignment -> offsets -> scatter -> MLP; dense: rows without a network are zero-filled (the reference's raw tensor),
// otherwise they stay unwritten and the caller consults ws.net_of
static int kilo_mlp_launch(const KiloRays& rays, const float* gmin_host, const float* gmax_host, const int32_t* fixed_res_host,
                           const int32_t* occ_res_host, const uint8_t* occupancy, const float* domain_mins,
                           const float* domain_maxs, const float* params, uint32_t param_stride, uint32_t num_networks,
                           int pos_freqs, int dir_freqs, int n_hidden, float* raw, bool dense, uint32_t* counts_out,
                           void* workspace, size_t workspace_bytes, hipStream_t st, KiloWs* ws_out,
                           const float* draw = nullptr, float* grad = nullptr /* non-null: gradients instead of raw */,
                           bool reuse_assignment = false /* the workspace still holds the assignment / order / segments of these very samples */) {
    // per-ray spans: only where the z lattice is evaluated on the fly and nothing dense is promised to the caller
    const bool use_spans = !dense && rays.pts == nullptr && rays.z_vals == nullptr;
    const bool backward = grad != nullptr;
    XR_REQUIRE(gmin_host && gmax_host && fixed_res_host, "null host pointer");
    XR_REQUIRE(num_networks >= 1, "num_networks must be >= 1");
    XR_REQUIRE(pos_freqs >= 0 && pos_freqs <= 16 && dir_freqs >= 0 && dir_freqs <= 16, "frequency count out of range");
    XR_REQUIRE(n_hidden >= 1 && n_hidden <= 8, "n_hidden must be in [1, 8]");
    const uint32_t n_floats = kilo_param_floats(pos_freqs, dir_freqs, n_hidden);
    XR_REQUIRE(param_stride >= n_floats && param_stride % 4 == 0, "param_stride too small or not a multiple of 4 floats");
    XR_REQUIRE((size_t)n_floats * 4 <= 64 * 1024, "parameter block does not fit the LDS staging buffer");
    XR_REQUIRE(occupancy == nullptr || occ_res_host != nullptr, "occupancy without its resolution");
    const uint64_t n = (uint64_t)rays.n_rays * rays.n_s;
    XR_REQUIRE(n > 0 && n < (1ull << 32), "sample count must be in (0, 2^32)");
    XR_REQUIRE(rays.viewdirs && domain_mins && domain_maxs && params && (backward ? draw != nullptr : raw != nullptr), "null pointer");
    XR_REQUIRE(!backward || (n_hidden <= 2 && pos_freqs <= 10 && dir_freqs <= 4),
               "gradients: n_hidden <= 2, pos_freqs <= 10, dir_freqs <= 4 (the architectures of the reference configs)");
    XR_REQUIRE(rays.pts || (rays.rays_o && rays.rays_d && (rays.z_vals || (rays.near && rays.far))),
               "either pts or (rays_o, rays_d, z_vals | near, far) is required");
    XR_REQUIRE(((uintptr_t)raw & 15) == 0 && ((uintptr_t)params & 15) == 0 && ((uintptr_t)draw & 15) == 0,
               "raw / params / draw must be 16-byte aligned");
    XR_REQUIRE(workspace && workspace_bytes >= kilo_ws_layout(n, num_networks, use_spans ? rays.n_rays : 0, nullptr, nullptr) &&
               ((uintptr_t)workspace & 255) == 0, "workspace too small or not 256-byte aligned");
    KiloGrid g;
    for (int a = 0; a < 3; ++a) {
        g.gmin[a] = gmin_host[a]; g.gmax[a] = gmax_host[a]; g.fixed_res[a] = fixed_res_host[a];
        g.occ_res[a] = occ_res_host ? occ_res_host[a] : 1;
        XR_REQUIRE(g.fixed_res[a] >= 1 && g.occ_res[a] >= 1, "resolution must be positive");
        volatile float size = g.gmax[a] - g.gmin[a];    // volatile: one fp32 rounding per operation, like the tensor ops
        volatile float lo = g.gmin[a] + 0.001f, hi = g.gmax[a] - 0.001f;
        volatile float vx = size / (float)g.fixed_res[a], ovx = size / (float)g.occ_res[a];
        g.lo_eps[a] = lo; g.hi_eps[a] = hi; g.voxel[a] = vx; g.ovoxel[a] = ovx;
    }
    KiloWs ws;
    kilo_ws_layout(n, num_networks, use_spans ? rays.n_rays : 0, (char*)workspace, &ws);
    if (ws_out) *ws_out = ws;
    if (!reuse_assignment) {
    XR_HIP(hipMemsetAsync(ws.counts, 0, (size_t)num_networks * 4, st));
    // work per workgroup: the per-workgroup histogram costs ~3 passes over the bins, so give each one enough samples
    // (span mode: per_block counts rays)
    uint32_t per_block, blocks, threads = 256;
    if (use_spans) {
        hipLaunchKernelGGL(k_kilo_spans, dim3(xr_div_up(rays.n_rays, 256)), dim3(256), 0, st, g, rays, ws.spans);
        XR_LAUNCH_CHECK();
        per_block = 32;                                  // 4 waves x 8 rays (measured: 128 rays 396 us, 64 rays / 1024 threads 540 us)
        blocks = xr_div_up(rays.n_rays, per_block);
    } else {
        per_block = n >= (64ull << 20) ? 16384 : 4096;
        blocks = xr_div_up(n, per_block);
    }
    const size_t hist_lds = num_networks <= KILO_MAX_LDS_BINS ? (size_t)num_networks * 4 : 0;
    hipLaunchKernelGGL(k_kilo_assign, dim3(blocks), dim3(threads), hist_lds, st, g, rays, occupancy, num_networks, n, per_block,
                       ws.spans, ws.net_of, ws.counts, dense ? reinterpret_cast<float4*>(raw) : nullptr);
    XR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kilo_offsets, dim3(1), dim3(1024), 0, st, ws.counts, num_networks, ws.seg_start, ws.tile_start, ws.cursor);
    XR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kilo_scatter, dim3(blocks), dim3(threads), hist_lds, st, ws.net_of, num_networks, n, per_block,
                       ws.spans, rays.n_s, rays.n_rays, ws.seg_start, ws.cursor, ws.order);
    XR_LAUNCH_CHECK();
    }
    KiloMlpArgs a{rays, domain_mins, domain_maxs, params, param_stride, num_networks, pos_freqs, dir_freqs, n_hidden,
                  ws.seg_start, ws.tile_start, ws.order, reinterpret_cast<float4*>(raw)};
    const uint64_t max_tiles = n / KILO_TILE + num_networks;
    int cus = xr_device_cus();
    if (cus <= 0) cus = 256;
    const uint32_t grid = (uint32_t)(max_tiles < (uint64_t)cus * 8 ? max_tiles : (uint64_t)cus * 8);
    if (!backward) {
        hipLaunchKernelGGL(k_kilo_mlp, dim3(grid), dim3(256), (size_t)n_floats * 4, st, a);
    } else {
        KiloBwdArgs bw{a, reinterpret_cast<const float4*>(draw), grad};
        const size_t lds = (size_t)n_floats * 4 + (size_t)KB_WAVES * KB_WAVE_FLOATS * 4;
        static bool attr = false;
        if (!attr) {
            XR_HIP(hipFuncSetAttribute((const void*)k_kilo_mlp_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            XR_HIP(hipFuncSetAttribute((const void*)k_kilo_mlp_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr = true;
        }
        XR_REQUIRE(lds <= 96 * 1024, "parameter block + the waves' tiles do not fit the LDS");
        if (n_hidden == 1) hipLaunchKernelGGL(k_kilo_mlp_bwd<1>, dim3(grid), dim3(64 * KB_WAVES), lds, st, bw);
        else hipLaunchKernelGGL(k_kilo_mlp_bwd<2>, dim3(grid), dim3(64 * KB_WAVES), lds, st, bw);
    }
    XR_LAUNCH_CHECK();
    if (counts_out != nullptr)
        XR_HIP(hipMemcpyAsync(counts_out, ws.counts, (size_t)num_networks * 4, hipMemcpyDeviceToDevice, st));
    return XR_OK;
}

extern "C" int xr_kilo_mlp_forward(const float* pts, const float* rays_o, const float* rays_d, const float* z_vals,
                                   const float* viewdirs, uint32_t n_rays, uint32_t n_samples, const float* gmin_host,
                                   const float* gmax_host, const int32_t* fixed_res_host, const int32_t* occ_res_host,
                                   const uint8_t* occupancy, const float* domain_mins, const float* domain_maxs,
                                   const float* params, uint32_t param_stride, uint32_t num_networks, int pos_freqs,
                                   int dir_freqs, int n_hidden, float* raw, uint32_t* counts_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if ((uint64_t)n_rays * n_samples == 0) return XR_OK;
    KiloRays rays{pts, rays_o, rays_d, z_vals, viewdirs, n_rays, n_samples, nullptr, nullptr, 0};
    return kilo_mlp_launch(rays, gmin_host, gmax_host, fixed_res_host, occ_res_host, occupancy, domain_mins, domain_maxs,
                           params, param_stride, num_networks, pos_freqs, dir_freqs, n_hidden, raw, true, counts_out,
                           workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

// parameter gradients for dL/draw [n_rays*n_samples, 4]; ACCUMULATES into grad_params [N, param_stride] (same block layout
// as params; caller zero-fills); rows of draw that no network evaluated are ignored
extern "C" int xr_kilo_mlp_backward(const float* pts, const float* rays_o, const float* rays_d, const float* z_vals,
                                    const float* viewdirs, uint32_t n_rays, uint32_t n_samples, const float* gmin_host,
                                    const float* gmax_host, const int32_t* fixed_res_host, const int32_t* occ_res_host,
                                    const uint8_t* occupancy, const float* domain_mins, const float* domain_maxs,
                                    const float* params, uint32_t param_stride, uint32_t num_networks, int pos_freqs,
                                    int dir_freqs, int n_hidden, const float* draw, float* grad_params, int reuse_assignment, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if ((uint64_t)n_rays * n_samples == 0) return XR_OK;
    XR_REQUIRE(grad_params != nullptr, "null pointer");
    KiloRays rays{pts, rays_o, rays_d, z_vals, viewdirs, n_rays, n_samples, nullptr, nullptr, 0};
    return kilo_mlp_launch(rays, gmin_host, gmax_host, fixed_res_host, occ_res_host, occupancy, domain_mins, domain_maxs,
                           params, param_stride, num_networks, pos_freqs, dir_freqs, n_hidden, nullptr, false, nullptr,
                           workspace, workspace_bytes, (hipStream_t)stream, nullptr, draw, grad_params, reuse_assignment != 0);
}

// ------------------------------------------------------------------------------------------ parameter blocks <-> the reference's tensors
// The reference keeps one tensor per layer of ALL networks (multi_modules.py:238-340: weight [N, in, out], bias [N, out]); the kernels
// above take one packed block per network.  Going from one form to the other was a dozen torch.cat / slice-copy launches per fine-tuning
// step in each direction; here it is one launch each: a thread per packed float finds its segment in a 16-entry table.
#define KILO_MAX_SEGS 16
struct KiloSegs {
    float* t[KILO_MAX_SEGS];                // tensor of the segment (nullptr: padding)
    uint32_t off[KILO_MAX_SEGS + 1];        // first packed float of the segment
    uint32_t cols[KILO_MAX_SEGS];           // floats per network in the tensor
    uint32_t rgbw[KILO_MAX_SEGS];           // != 0: the rgb weight, [H, 3] in the tensor, [H, 4] (4th column zero) in the block
    uint32_t n_seg, n_floats;
};
static int kilo_segments(float* const* tensors, int pos_freqs, int dir_freqs, int n_hidden, KiloSegs* out) {
    const uint32_t P = 3u * (2u * pos_freqs + 1u), D = 3u * (2u * dir_freqs + 1u), H = KILO_H;
    KiloSegs& S = *out;
    uint32_t n = 0, off = 0;
    auto seg = [&](float* t, uint32_t width, uint32_t cols, uint32_t rgbw) { S.t[n] = t; S.off[n] = off; S.cols[n] = cols; S.rgbw[n] = rgbw; off += width; ++n; };
    int k = 0;
    for (int l = 0; l < n_hidden; ++l) { const uint32_t in = l == 0 ? P : H; seg(tensors[k], in * H, in * H, 0); seg(tensors[k + 1], H, H, 0); k += 2; }
    seg(tensors[k], H, H, 0); seg(tensors[k + 1], 1, 1, 0); seg(nullptr, 3, 0, 0); k += 2;            // alpha: W, b, 3 pad
    seg(tensors[k], H * H, H * H, 0); seg(tensors[k + 1], H, H, 0); k += 2;                            // feature
    seg(tensors[k], (H + D) * H, (H + D) * H, 0); seg(tensors[k + 1], H, H, 0); k += 2;                // direction
    seg(tensors[k], H * 4, H * 3, 1); seg(tensors[k + 1], 3, 3, 0); seg(nullptr, 1, 0, 0);             // rgb: W (3 -> 4 columns), b, 1 pad
    S.off[n] = off; S.n_seg = n; S.n_floats = off;
    return off == kilo_param_floats(pos_freqs, dir_freqs, n_hidden) && n <= KILO_MAX_SEGS ? XR_OK : XR_EINVAL;
}
template <bool UNPACK>
__global__ void __launch_bounds__(256) k_kilo_repack(KiloSegs S, float* __restrict__ blocks, uint32_t param_stride, uint32_t num_networks, int clear) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (e >= S.n_floats) return;
    uint32_t s = 0;
    while (s + 1 < S.n_seg && e >= S.off[s + 1]) ++s;
    const uint32_t idx = e - S.off[s];
    float* b = blocks + (size_t)n * param_stride + e;
    float* t = S.t[s];
    size_t src = 0;
    bool has = t != nullptr;
    if (has) {
        if (S.rgbw[s]) { has = (idx & 3u) != 3u; src = (size_t)n * S.cols[s] + (idx >> 2) * 3u + (idx & 3u); }
        else src = (size_t)n * S.cols[s] + idx;
    }
    if (UNPACK) {
        if (has) t[src] = *b;
        if (clear) *b = 0.f;
    } else {
        *b = has ? t[src] : 0.f;
    }
}
static int kilo_repack(bool unpack, float* const* tensors_host, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden, float* blocks,
                       uint32_t param_stride, int clear, void* stream) {
    XR_REQUIRE(tensors_host && blocks && num_networks >= 1 && num_networks <= 65535, "null pointer / network count");
    XR_REQUIRE(pos_freqs >= 0 && pos_freqs <= 16 && dir_freqs >= 0 && dir_freqs <= 16 && n_hidden >= 1 && 2 * n_hidden + 11 <= KILO_MAX_SEGS, "architecture out of range");
    for (int k = 0; k < 2 * (n_hidden + 4); ++k) XR_REQUIRE(tensors_host[k] != nullptr, "null parameter tensor");
    KiloSegs S;
    XR_REQUIRE(kilo_segments(tensors_host, pos_freqs, dir_freqs, n_hidden, &S) == XR_OK, "segment table does not match the block layout");
    XR_REQUIRE(param_stride >= S.n_floats, "param_stride too small");
    const dim3 grid(xr_div_up(S.n_floats, 256), num_networks);
    if (unpack) hipLaunchKernelGGL(k_kilo_repack<true>, grid, dim3(256), 0, (hipStream_t)stream, S, blocks, param_stride, num_networks, clear);
    else hipLaunchKernelGGL(k_kilo_repack<false>, grid, dim3(256), 0, (hipStream_t)stream, S, blocks, param_stride, num_networks, 0);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
// tensors_host: 2 (n_hidden + 4) device pointers in MultiNetwork.ordered_parameters() order (weight, bias of pts_linears.*, alpha, feature,
// direction, rgb; contiguous fp32) -> blocks [N, param_stride]
extern "C" int xr_kilo_pack_params(const float* const* tensors_host, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                                   float* blocks, uint32_t param_stride, void* stream) {
    return kilo_repack(false, const_cast<float* const*>(reinterpret_cast<const float* const*>(tensors_host)), num_networks, pos_freqs, dir_freqs,
                       n_hidden, blocks, param_stride, 0, stream);
}
// the packed gradient blocks -> one tensor per parameter (same order and shapes); clear_blocks != 0 leaves the blocks zero-filled for the
// next xr_kilo_mlp_backward, which accumulates into them
extern "C" int xr_kilo_unpack_grads(float* blocks, uint32_t param_stride, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                                    float* const* grads_host, int clear_blocks, void* stream) {
    return kilo_repack(true, grads_host, num_networks, pos_freqs, dir_freqs, n_hidden, blocks, param_stride, clear_blocks, stream);
}

extern "C" size_t xr_kilo_render_workspace_bytes(uint32_t n_rays, uint32_t n_samples, uint32_t num_networks) {
    const uint64_t n = (uint64_t)n_rays * n_samples;
    return ((n * 16 + 255) & ~(size_t)255) + kilo_ws_layout(n, num_networks, n_rays, nullptr, nullptr);
}

extern "C" int xr_kilo_render_rays(const float* rays_o, const float* rays_d, const float* viewdirs, const float* near,
                                   const float* far, uint32_t n_rays, uint32_t n_samples, int lindisp,
                                   const float* gmin_host, const float* gmax_host, const int32_t* fixed_res_host,
                                   const int32_t* occ_res_host, const uint8_t* occupancy, const float* domain_mins,
                                   const float* domain_maxs, const float* params, uint32_t param_stride,
                                   uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden, int white_bkgd,
                                   float* rgb, float* disp, float* acc, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    const uint64_t n = (uint64_t)n_rays * n_samples;
    if (n == 0) return XR_OK;
    XR_REQUIRE(near && far && rgb && disp && acc, "null pointer");
    XR_REQUIRE(workspace && workspace_bytes >= xr_kilo_render_workspace_bytes(n_rays, n_samples, num_networks) &&
               ((uintptr_t)workspace & 255) == 0, "workspace too small or not 256-byte aligned");
    const size_t raw_bytes = (n * 16 + 255) & ~(size_t)255;
    float* raw = (float*)workspace;                      // touched only where a network is evaluated
    KiloRays rays{nullptr, rays_o, rays_d, nullptr, viewdirs, n_rays, n_samples, near, far, lindisp};
    KiloWs ws;
    int rc = kilo_mlp_launch(rays, gmin_host, gmax_host, fixed_res_host, occ_res_host, occupancy, domain_mins, domain_maxs,
                             params, param_stride, num_networks, pos_freqs, dir_freqs, n_hidden, raw, false, nullptr,
                             (char*)workspace + raw_bytes, workspace_bytes - raw_bytes, (hipStream_t)stream, &ws);
    if (rc) return rc;
    hipLaunchKernelGGL(k_nerf_render, dim3(xr_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(raw), rays, ws.net_of, ws.spans, white_bkgd, rgb, disp, acc, (float*)nullptr);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_nerf_render_forward(const float* raw, const float* z_vals, const float* rays_d, uint32_t n_rays,
                                      uint32_t n_samples, int white_bkgd, float* rgb, float* disp, float* acc,
                                      float* weights, void* stream) {
    XR_REQUIRE(n_samples >= 1, "n_samples must be >= 1");
    if (n_rays == 0) return XR_OK;
    XR_REQUIRE(raw && z_vals && rays_d && rgb && disp && acc, "null pointer");
    XR_REQUIRE(((uintptr_t)raw & 15) == 0, "raw must be 16-byte aligned");
    KiloRays zr{nullptr, nullptr, rays_d, z_vals, nullptr, n_rays, n_samples, nullptr, nullptr, 0};
    hipLaunchKernelGGL(k_nerf_render, dim3(xr_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(raw), zr, (const int32_t*)nullptr, (const uint32_t*)nullptr, white_bkgd, rgb,
                       disp, acc, weights);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
