// micro-probe: what does MI355X sustain for the hash-grid forward's access pattern -- independent random gathers of 8 / 16
// bytes from a table slice -- as a function of (a) the slice size per XCD (L1 / L2 / Infinity-Cache resident), (b) whether
// every XCD gathers from its OWN slice (the level-major mapping of k_hashgrid_fwd) or all XCDs from one big table,
// (c) the load width, (d) two loads hitting the same 128-B line (the x-neighbour pair)?  Gives the roofline the kernel is
// to be compared with: G lane-gathers/s and G distinct 128-B lines/s.
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: 8 x 8-B loads at 8 random entries; 1: 4 x 16-B loads at 4 random aligned pairs; 2: 8 x 8-B loads = 4 random lines,
// entry e and e^1 of each (pairs in separate instructions); 3: 8 x 8-B, 4 random lines, e and e^3 (same line, not adjacent)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float2* __restrict__ tab, uint32_t slice_entries, int affine, uint32_t iters, float* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const float2* t = tab + (affine ? (size_t)(blockIdx.x & 7) * slice_entries : 0);
    const uint32_t mask = slice_entries - 1;
    float acc = 0.f;
    for (uint32_t k = 0; k < iters; ++k) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = hash32(tid * 977u + (k * 8 + j) * 0x9e3779b9u) & mask;
        if (MODE == 0) {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = t[e[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
        } else if (MODE == 1) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(t + (e[j] & ~1u));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        } else {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = t[e[j]]; v[2 * j + 1] = t[e[j] ^ (MODE == 2 ? 1u : 3u)]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
        }
    }
    if (acc == 12345.678f) out[tid] = acc;
}

template <int MODE>
void run(const float2* tab, uint32_t slice_entries, int affine, float* out) {
    const int blocks = 16384; const uint32_t iters = 8;        // 4.2 M threads x 64 gathers... = the kernel's 4.1 M sample-levels x 8
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(tab, slice_entries, affine, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, 256>>>(tab, slice_entries, affine, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double lanes = (double)blocks * 256 * iters * (MODE == 1 ? 4 : 8), lines = (double)blocks * 256 * iters * (MODE == 0 ? 8 : 4);
    printf("mode %d  slice %7.0f KiB  %s  %8.1f us   %6.1f G loads/s  %6.1f G lines/s   %6.0f GB/s useful\n", MODE, slice_entries * 8.0 / 1024,
           affine ? "per-XCD slices" : "one shared table", ms * 1e3, lanes / ms / 1e6, lines / ms / 1e6, lanes * (MODE == 1 ? 16 : 8) / ms / 1e6);
}

int main() {
    float2* tab; float* out;
    hipMalloc(&tab, (size_t)64 << 20); hipMemset(tab, 0, (size_t)64 << 20); hipMalloc(&out, (size_t)16384 * 256 * 4);
    for (uint32_t kib : {16u, 512u, 2048u, 4096u, 8192u}) {
        const uint32_t ent = kib * 1024 / 8;
        run<0>(tab, ent, 1, out); run<1>(tab, ent, 1, out); run<2>(tab, ent, 1, out); run<3>(tab, ent, 1, out);
    }
    for (uint32_t kib : {4096u, 32768u, 65536u}) {
        const uint32_t ent = kib * 1024 / 8;
        run<0>(tab, ent, 0, out); run<1>(tab, ent, 0, out); run<2>(tab, ent, 0, out);
    }
    return 0;
}
