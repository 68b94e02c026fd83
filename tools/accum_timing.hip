// phase timing of k_scatter_accum (one level, 64 partitions, 64 sample blocks, half-full sub-bins)
#define SC_TIMING
#include "../xrnerf_amd/csrc/xr_encode.hip"
#include <vector>
#include <cstdio>
#include <cstring>
void xr_set_error(const char*, ...) {}
int main() {
    const uint32_t parts = 64, nsb = 64, cap = SC_SUB_ITEMS / parts;
    GridMeta gm{}; gm.n_levels = 1; gm.off[0] = 0; gm.off[1] = 1u << 19; gm.res[0] = 2048; gm.scale[0] = 2047.f;
    std::vector<uint32_t> cnt(parts * nsb, cap / 2);
    std::vector<float4> items((size_t)nsb * SC_SUB_ITEMS);
    uint32_t h = 1;
    for (auto& it : items) { h = h * 1664525u + 1013904223u; uint32_t i0 = (h >> 8) & (SC_ENTRIES - 1); uint32_t pr = i0 | ((i0 ^ 1u) << SC_LOG2); float f; memcpy(&f, &pr, 4); it = make_float4(f, 1.f, 2.f, 0.25f); }
    uint32_t* d_cnt; float4* d_bins; float* d_tab;
    hipMalloc(&d_cnt, cnt.size() * 4); hipMalloc(&d_bins, items.size() * 16); hipMalloc(&d_tab, (size_t)2 << 21);
    hipMemcpy(d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_bins, items.data(), items.size() * 16, hipMemcpyHostToDevice);
    hipMemset(d_tab, 0, (size_t)2 << 21);
    hipFuncSetAttribute((const void*)k_scatter_accum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_scatter_accum, dim3(parts), dim3(SC_THREADS), SC_LDS_BYTES, 0, gm, 0u, 1u, parts, nsb, d_cnt, d_bins, d_tab);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_sc_t), sizeof(t));
        printf("kernel %.1f us | block 0 (100 MHz ticks -> us): zero+counts %.2f  loop %.2f  sync %.2f  flush %.2f\n", ms * 1e3,
               (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0, (t[4] - t[3]) / 100.0);
    }
    return 0;
}
