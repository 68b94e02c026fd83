// The linear layers of the 8x256 NeRF MLP (configs #1 and #3: xrnerf/models/mlps/nerf_mlp.py:27-94, nn.Linear + F.relu) to fp32
// accuracy on the bf16 matrix cores, with the operands kept SPLIT in HBM.
//
// xr_gemm.hip's k_gemm_b3 splits every fp32 operand element into three bf16 numbers (8 + 8 + 8 significand bits, exact) while its
// panel is staged into LDS: the same activation row is split again by every column tile that reads it, the same weight panel by
// every one of the 1024 row tiles, and the split's ~6 VALU instructions per element sit between the global load and the LDS store of
// every k-step (105-120 TFLOP/s forward; the two backward products stayed on the fp32 MFMA at 64-72).  Here a tensor that feeds a
// product exists as three bf16 PLANES [3][rows][ld] (plane stride ps): it is split ONCE, by the kernel that produces it (the epilogue
// of the layer in front, xr_p3_split for network inputs and incoming gradients), and the product's main loop is the plain bf16
// pattern -- 16-byte global loads -> 16-byte LDS stores -> ds_read_b128 fragments -> v_mfma_f32_32x32x16_bf16 -- with SIX MFMAs
// per fragment pair (the terms above 2^-23 of the product, smallest first, fp32 accumulate).  Per MFMA that is half the staging
// bytes of a bf16 GEMM.
//
//   xr_p3_gemm_nt   C[M,N] = A[M,K] . B[N,K]^T (+ bias, relu, gradient mask)   forward (B = weight) and input gradient (B = weight^T)
//   xr_p3_gemm_tn   C[N',K'] = A[Mk,N']^T . B[Mk,K']  split over Mk            weight gradient (fixed-order partial sums)
//   xr_p3_split     fp32 [M,K] -> planes;  xr_p3_colsum  column sums of planes (bias gradient)
//
// Tile 128 x 128 x 32, 256 threads = 4 waves in 2 x 2, each 64 x 64 = 2 x 2 accumulators of 32 x 32; 48 KB of LDS (3 planes x 2
// operands x 128 rows x 64 B, the four 16-byte slots of a row XOR-swizzled with (row >> 2) & 3: ds_read_b128's 16-lane groups then
// hit 16 distinct slots), three workgroups per CU, which is what hides a workgroup's global-load latency.  The WEIGHT fragment is
// the MFMA's A operand, so a lane of the accumulator holds one sample row and 4 consecutive output columns per register quad:
// the epilogue packs them to 8-byte bf16 runs, transposes through LDS per wave and writes full 128-byte row segments of each plane.
#include "xr_common.h"
#include <cstdlib>

typedef float p3f16 __attribute__((ext_vector_type(16)));
typedef __bf16 p3b8 __attribute__((ext_vector_type(8)));
#define P3MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define P3_T 128                                 // tile edge (rows of A, rows of B)
#define P3_BK 32                                 // k per step
#define P3_PLANE (P3_T * P3_BK)                  // halves per plane of one operand tile
#define P3_ST 72                                 // halves per row of the epilogue's per-wave staging tile (64 columns + 16 B)

__device__ __forceinline__ void p3_split(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;              // exact
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;             // exact
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(uint16_t, bh); m = __builtin_bit_cast(uint16_t, bm); l = __builtin_bit_cast(uint16_t, bl);
}
// a bf16 pattern that is > 0 (not zero, not negative, NaN counts as positive like `x > 0` does not -- masks come from relu outputs)
__device__ __forceinline__ uint32_t p3_pos_mask2(uint32_t two) {
    const uint32_t lo = ((two & 0x8000u) == 0u && (two & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
    const uint32_t hi = ((two & 0x80000000u) == 0u && (two & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
    return lo | hi;
}

// ------------------------------------------------------------------------------------------------ fp32 -> planes
// thread = 8 consecutive columns of one row; columns K .. Kpad - 1 are written as zeros (the products' k-steps are 32 wide)
__global__ void __launch_bounds__(256) k_p3_split(const float* __restrict__ x, uint32_t M, uint32_t K, uint32_t ldx, const float* __restrict__ mask,
                                                  uint32_t ldm, uint16_t* __restrict__ out, uint32_t ldo, size_t ps, uint32_t Kpad) {
    const uint32_t chunks = Kpad / 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)M * chunks) return;
    const uint32_t r = (uint32_t)(id / chunks), c = (uint32_t)(id % chunks) * 8;
    uint16_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        if (c + e < K) {
            v = x[(size_t)r * ldx + c + e];
            if (mask != nullptr && !(mask[(size_t)r * ldm + c + e] > 0.f)) v = 0.f;
        }
        p3_split(v, h[e], m[e], l[e]);
    }
    auto pack = [](const uint16_t (&a)[8]) {
        return make_uint4(a[0] | ((uint32_t)a[1] << 16), a[2] | ((uint32_t)a[3] << 16), a[4] | ((uint32_t)a[5] << 16), a[6] | ((uint32_t)a[7] << 16));
    };
    uint16_t* o = out + (size_t)r * ldo + c;
    *reinterpret_cast<uint4*>(o) = pack(h);
    *reinterpret_cast<uint4*>(o + ps) = pack(m);
    *reinterpret_cast<uint4*>(o + 2 * ps) = pack(l);
}

extern "C" int xr_p3_split(const float* x, uint32_t M, uint32_t K, uint32_t ldx, const float* mask_src, uint32_t ld_mask, void* planes,
                           uint32_t ld_planes, size_t plane_stride, uint32_t K_padded, void* stream) {
    if (M == 0 || K_padded == 0) return XR_OK;
    XR_REQUIRE(x && planes, "null pointer");
    XR_REQUIRE(K_padded % 8 == 0 && K_padded >= K && ld_planes % 8 == 0 && ld_planes >= K_padded && plane_stride % 8 == 0 && ((uintptr_t)planes & 15) == 0,
               "planes: 16-byte aligned rows, padded width a multiple of 8");
    const size_t n = (size_t)M * (K_padded / 8);
    hipLaunchKernelGGL(k_p3_split, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M, K, ldx, mask_src, ld_mask,
                       (uint16_t*)planes, ld_planes, plane_stride, K_padded);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------------------------------------ C = A . B^T
struct P3Nt {
    const uint16_t* A; const uint16_t* B;        // planes [M][lda], [N][ldb]; K is a multiple of 32 (zero-padded columns)
    uint32_t lda, ldb; size_t psa, psb;
    uint32_t M, N, K;
    const float* bias;                           // [N], nullable
    int relu;
    const uint16_t* mask; uint32_t ldmask;       // nullable: the HIGH plane of the layer's forward output [M][ldmask]; C counts where it is > 0
    uint16_t* Cp; uint32_t ldcp; size_t pscp;    // planes out, nullable (N % 8 == 0)
    float* C; uint32_t ldc;                      // fp32 out, nullable (N % 4 == 0)
};

__global__ void __launch_bounds__(256, 3) k_p3_gemm_nt(P3Nt g) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 3 * P3_PLANE];
    uint16_t* sA = lds;
    uint16_t* sB = lds + 3 * P3_PLANE;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const uint32_t n0 = blockIdx.x * P3_T, m0 = blockIdx.y * P3_T;     // column tiles vary fastest: the tiles that share an A row panel run together
    p3f16 acc[2][2];                             // [jn][im]: weights are the MFMA's A operand -> lane = sample row, registers = output columns
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const uint32_t sr = t >> 2, sc = t & 3;      // staging: rows sr and sr + 64 of each plane, 16-byte chunk sc of the 64-byte row
    for (uint32_t k0 = 0; k0 < g.K; k0 += P3_BK) {
        uint4 va[3][2], vb[3][2];
#if defined(P3_PROBE) && P3_PROBE == 1
        if (k0 == 0)
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t ra = m0 + sr + 64 * j, rb = n0 + sr + 64 * j;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                va[p][j] = ra < g.M ? *reinterpret_cast<const uint4*>(g.A + p * g.psa + (size_t)ra * g.lda + k0 + 8 * sc) : make_uint4(0, 0, 0, 0);
                vb[p][j] = rb < g.N ? *reinterpret_cast<const uint4*>(g.B + p * g.psb + (size_t)rb * g.ldb + k0 + 8 * sc) : make_uint4(0, 0, 0, 0);
            }
        }
        if (k0) __syncthreads();                 // the previous tile has been consumed
#if defined(P3_PROBE) && P3_PROBE == 1
        if (k0 == 0)
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t r = sr + 64 * j, slot = sc ^ ((r >> 2) & 3);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                *reinterpret_cast<uint4*>(sA + p * P3_PLANE + r * P3_BK + 8 * slot) = va[p][j];
                *reinterpret_cast<uint4*>(sB + p * P3_PLANE + r * P3_BK + 8 * slot) = vb[p][j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < P3_BK / 16; ++s) {
            p3b8 fx[2][3], fw[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t rx = wm * 64 + 32 * i + l31, rw = wn * 64 + 32 * i + l31;
                const uint32_t ox = rx * P3_BK + 8 * ((2 * s + hi) ^ ((rx >> 2) & 3)), ow = rw * P3_BK + 8 * ((2 * s + hi) ^ ((rw >> 2) & 3));
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fx[i][p] = *reinterpret_cast<const p3b8*>(sA + p * P3_PLANE + ox);
                    fw[i][p] = *reinterpret_cast<const p3b8*>(sB + p * P3_PLANE + ow);
                }
            }
#if defined(P3_PROBE) && P3_PROBE == 2
            for (int jn = 0; jn < 2; ++jn) for (int im = 0; im < 2; ++im) for (int p = 0; p < 3; ++p) { acc[jn][im][p] += (float)fw[jn][p][0] + (float)fx[im][p][1]; }
            if (false)
#endif
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int im = 0; im < 2; ++im) {              // smallest terms first
                    acc[jn][im] = P3MFMA(fw[jn][2], fx[im][0], acc[jn][im]);
                    acc[jn][im] = P3MFMA(fw[jn][0], fx[im][2], acc[jn][im]);
                    acc[jn][im] = P3MFMA(fw[jn][1], fx[im][1], acc[jn][im]);
                    acc[jn][im] = P3MFMA(fw[jn][1], fx[im][0], acc[jn][im]);
                    acc[jn][im] = P3MFMA(fw[jn][0], fx[im][1], acc[jn][im]);
                    acc[jn][im] = P3MFMA(fw[jn][0], fx[im][0], acc[jn][im]);
                }
        }
    }
    // ---- epilogue.  lane: sample row 32 im + l31 of the wave's 64; register r of acc[jn][im]: column 32 jn + (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t n = n0 + wn * 64 + 32 * jn + 8 * q + 4 * hi;
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (g.bias != nullptr)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = n + e < g.N ? g.bias[n + e] : 0.f;
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[jn][im][4 * q + e] + b[e];
                    if (g.relu) v = fmaxf(v, 0.f);
                    acc[jn][im][4 * q + e] = v;
                }
        }
    if (g.C != nullptr) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const uint32_t m = m0 + wm * 64 + 32 * im + l31;
            if (m >= g.M) continue;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t n = n0 + wn * 64 + 32 * jn + 8 * q + 4 * hi;
                    if (n >= g.N) continue;
                    float4 v = make_float4(acc[jn][im][4 * q], acc[jn][im][4 * q + 1], acc[jn][im][4 * q + 2], acc[jn][im][4 * q + 3]);
                    if (g.mask != nullptr) {
                        const uint2 mk = *reinterpret_cast<const uint2*>(g.mask + (size_t)m * g.ldmask + n);
                        const uint32_t s0 = p3_pos_mask2(mk.x), s1 = p3_pos_mask2(mk.y);
                        if (!(s0 & 0xffffu)) v.x = 0.f;
                        if (!(s0 >> 16)) v.y = 0.f;
                        if (!(s1 & 0xffffu)) v.z = 0.f;
                        if (!(s1 >> 16)) v.w = 0.f;
                    }
                    *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + n) = v;
                }
        }
    }
    if (g.Cp == nullptr) return;                 // (uniform)
    __syncthreads();                             // every wave is done with the operand tiles: the LDS becomes four staging tiles
    uint16_t* st = lds + wave * (64 * P3_ST);
    const uint32_t rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint16_t h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint16_t a, b, c;
                        p3_split(acc[jn][im][4 * q + e], a, b, c);
                        h[e] = p == 0 ? a : (p == 1 ? b : c);
                    }
                    *reinterpret_cast<uint2*>(st + (32 * im + l31) * P3_ST + 32 * jn + 8 * q + 4 * hi) =
                        make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const uint32_t row = rrow + 8 * tt;
            const uint32_t m = m0 + wm * 64 + row, n = n0 + wn * 64 + 8 * rchunk;
            uint4 v = *reinterpret_cast<const uint4*>(st + row * P3_ST + 8 * rchunk);
            if (m >= g.M || n >= g.N) continue;
            if (g.mask != nullptr) {             // (re-read per plane: 16 B per lane from L2 instead of 32 registers held across the planes)
                const uint4 w = *reinterpret_cast<const uint4*>(g.mask + (size_t)m * g.ldmask + n);
                v.x &= p3_pos_mask2(w.x); v.y &= p3_pos_mask2(w.y); v.z &= p3_pos_mask2(w.z); v.w &= p3_pos_mask2(w.w);
            }
            *reinterpret_cast<uint4*>(g.Cp + p * g.pscp + (size_t)m * g.ldcp + n) = v;
        }
        __syncthreads();
    }
}

static int p3_check_planes(const void* p, uint32_t ld, size_t ps) {
    return p != nullptr && ld % 8 == 0 && ps % 8 == 0 && ((uintptr_t)p & 15) == 0;
}

// C [M,N] = act(A [M,K] . B [N,K]^T + bias) (* (mask_hi > 0)): A, B bf16 planes with K (a multiple of 32) zero-padded columns;
// outputs: bf16 planes (C_planes, N % 8 == 0) and / or fp32 (C, N % 4 == 0).  mask_hi: the high plane of a relu layer's output at the
// same [m][n] (the input gradient of the layer behind it is masked where that layer's output is not positive).
extern "C" int xr_p3_gemm_nt(const void* A, uint32_t lda, size_t psa, const void* B, uint32_t ldb, size_t psb, uint32_t M, uint32_t N, uint32_t K,
                             const float* bias, int relu, const void* mask_hi, uint32_t ld_mask, void* C_planes, uint32_t ldcp, size_t pscp,
                             float* C, uint32_t ldc, void* stream) {
    if (M == 0 || N == 0) return XR_OK;
    XR_REQUIRE(p3_check_planes(A, lda, psa) && p3_check_planes(B, ldb, psb), "operand planes: 16-byte aligned, leading dimension a multiple of 8");
    XR_REQUIRE(K >= P3_BK && K % P3_BK == 0 && lda >= K && ldb >= K, "K must be a multiple of 32 (zero-padded planes)");
    XR_REQUIRE(C_planes != nullptr || C != nullptr, "no output");
    XR_REQUIRE(C_planes == nullptr || (p3_check_planes(C_planes, ldcp, pscp) && N % 8 == 0 && ldcp >= N), "output planes: N a multiple of 8");
    XR_REQUIRE(C == nullptr || (N % 4 == 0 && ldc % 4 == 0 && ldc >= N && ((uintptr_t)C & 15) == 0), "fp32 output: N a multiple of 4, 16-byte aligned rows");
    XR_REQUIRE(mask_hi == nullptr || (ld_mask % 8 == 0 && ((uintptr_t)mask_hi & 15) == 0 && ld_mask >= N && N % 4 == 0), "mask plane: 16-byte aligned rows");
    XR_REQUIRE(xr_div_up(M, P3_T) <= 65535, "more than 65535 row tiles in one call");
    P3Nt g{(const uint16_t*)A, (const uint16_t*)B, lda, ldb, psa, psb, M, N, K, bias, relu, (const uint16_t*)mask_hi, ld_mask,
           (uint16_t*)C_planes, ldcp, pscp, C, ldc};
    hipLaunchKernelGGL(k_p3_gemm_nt, dim3(xr_div_up(N, P3_T), xr_div_up(M, P3_T)), dim3(256), 0, (hipStream_t)stream, g);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
