// Issue behaviour of v_mfma_f32_32x32x16_bf16 on gfx950: cycles per MFMA when 1 / 2 / 4 accumulators are updated in turn
// (dependent chains of different lengths), with one wave and with two waves per SIMD, and with VALU conversions in between.
// Why: the split forward (csrc/xr_mlp.hip, k_nerf_mlp_fwd_b3) spends ~10.8 K cycles per tile against 4.6 K of MFMA passes.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_chain_probe.hip -o tools/mfma_chain_probe && tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define M(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
template <int NACC, bool CVT>
__global__ void k(float* out, long long* cyc, int iters) {
    b8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = out[threadIdx.x & 63];
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u % NACC] = M(a, b, acc[u % NACC]);
        if (CVT) {          // ~ the split of one 16-value tile: conversions depending on the accumulators
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[0][r] + x;
                const __bf16 h = (__bf16)v; const float r1 = v - (float)h; const __bf16 m = (__bf16)r1;
                x += (float)m * 1e-9f;
            }
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    const long long t1 = wall_clock64();
    float s = x;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
template <int NACC, bool CVT> void run(const char* name, int threads, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, CVT>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    long long h[2];
    if (hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    printf("%-34s waves/SIMD %d: %6.1f shader cycles per MFMA (per wave), %6.1f per MFMA and SIMD   [wall clock x%.2f]\n", name, threads / 256,
           (double)h[0] / (iters * 24.0), (double)h[0] / (iters * 24.0) / (threads / 256), (double)h[1] / (double)h[0]);
}
int main() {
    float* out; long long* cyc;
    if (hipMalloc(&out, 1024 * 4) != hipSuccess || hipMalloc(&cyc, 16) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemset(out, 0, 1024 * 4);
    for (int threads : {256, 512}) {
        if (threads == 256) { run<1, false>("1 accumulator (24-long chain)", 256, out, cyc); run<2, false>("2 accumulators in turn", 256, out, cyc);
                              run<4, false>("4 accumulators in turn", 256, out, cyc); run<4, true>("4 accumulators + 16-value split", 256, out, cyc); }
        else { run<1, false>("1 accumulator (24-long chain)", 512, out, cyc); run<2, false>("2 accumulators in turn", 512, out, cyc);
               run<4, false>("4 accumulators in turn", 512, out, cyc); run<4, true>("4 accumulators + 16-value split", 512, out, cyc); }
    }
    return 0;
}
