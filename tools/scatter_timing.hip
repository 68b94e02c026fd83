// phase timing of the second-generation scatter pair (k_scatter_bin2 / k_scatter_accum2) on 2^18 uniform random samples,
// 11 hashed levels: wall_clock64 (100 MHz) stamps of workgroup 0 + event time of each kernel.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -DSC_TIMING tools/scatter_timing.hip -o tools/scatter_timing
#include "../xrnerf_amd/csrc/xr_encode.hip"
#include <vector>
#include <cstdio>
#include <cstring>
#include <cmath>
void xr_set_error(const char*, ...) {}
int main() {
    const uint32_t n = 1u << 18, nl = 11, l_lo = 5, l_hi = 16, parts = (1u << 19) >> SC_LOG2, nsb = n / SC_BLOCK_SAMPLES;
    printf("SC_LOG2 %d  SC_ACC_THREADS %d  SB_THREADS %d  SB_SPT %d\n", SC_LOG2, SC_ACC_THREADS, SB_THREADS, SB_SPT);
    float scale[16]; uint32_t res[16], off[17];
    xr_hashgrid_meta(16, 19, 16, std::exp2(std::log2(2048.0 / 16) / 15), scale, res, off);
    GridMeta gm; uint32_t hm; fill_meta(&gm, &hm, 16, scale, res, off);
    std::vector<float> x(n * 3), d((size_t)32 * n);
    uint32_t h = 1; auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (h >> 8) * (1.f / 16777216.f); };
    for (auto& v : x) v = rnd();
    for (auto& v : d) v = rnd() - 0.5f;
    float *dx, *dd, *tab; uint32_t* cnt; float4* bins;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dd, d.size() * 4); hipMalloc(&tab, (size_t)off[16] * 8);
    hipMalloc(&cnt, (size_t)nl * parts * nsb * 4); hipMalloc(&bins, (size_t)nl * nsb * SC_SUB_ITEMS2 * 16);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice);
    hipMemset(tab, 0, (size_t)off[16] * 8);
    hipFuncSetAttribute((const void*)k_scatter_accum2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES);
    hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_scatter_bin2, dim3(nl * nsb), dim3(SB_THREADS), 0, 0, gm, l_lo, l_hi, parts, nsb, dx, 3u, dd, n, n, (const uint32_t*)nullptr, cnt, bins, tab);
        hipEventRecord(b);
        hipLaunchKernelGGL(k_scatter_accum2, dim3(nl * parts), dim3(SC_ACC_THREADS), SC_LDS_BYTES, 0, gm, l_lo, l_hi, parts, nsb, cnt, bins, tab);
        hipEventRecord(c); hipEventSynchronize(c);
        float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, b, c);
        long long t[24]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_sc_t), sizeof(t));
        auto us = [&](int i, int j) { return (t[j] - t[i]) / 100.0; };
        printf("bin2 %.1f us | wg0 round0: loads+rank %.2f  sync %.2f  scan %.2f  place %.2f  copy-out %.2f  sync %.2f | whole wg %.2f\n", m1 * 1e3,
               us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(13, 14), us(8, 15));
        printf("accum2 %.1f us | wg0: zero+fills %.2f  stream+atomics %.2f  sync %.2f  flush %.2f | whole wg %.2f\n", m2 * 1e3,
               us(16, 17), us(17, 18), us(18, 19), us(19, 20), us(16, 20));
    }
    return 0;
}
