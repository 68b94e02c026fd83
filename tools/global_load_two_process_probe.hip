// Round 6, second reproducer for the two-processes-on-one-GPU scatter deviation.  tools/lds_atomic_two_process_probe.hip cleared the LDS
// atomics (0 errors in 7e4 launches, alone / two queues / two processes), and tools/scatter_determinism_probe.py shows the signature of
// every event: the sub-bin fill counts never differ, and the wrong table floats are FEATURE 0 of the two entries of ~16 items that land in
// ~16 different accumulate workgroups -- i.e. the value `denc_t[2 l][i]` that k_scatter_bin3 LOADED for ~4 samples of one level.
// This kernel loads the way that kernel's fetch() does -- two rows of a [32][ld] float matrix and three floats of a 7-float coordinate
// row per sample, two samples per thread, the next round's loads issued before a phase of returning LDS atomics -- from buffers holding
// a pattern, and checks every loaded value.   err[k]: wrong values of load slot k (0: row 2l, 1: row 2l + 1, 2..4: x, y, z); err[5..]:
// the first wrong value's (slot, lane, index, got, want).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/global_load_two_process_probe tools/global_load_two_process_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define THREADS 512
#define SPT 2
#define BS 2048
#define PARTS 64
__host__ __device__ inline uint32_t pat(uint32_t row, uint32_t i) { return (row * 0x9e3779b1u) ^ (i * 2654435761u) ^ 0x5bd1e995u; }

__global__ __launch_bounds__(THREADS) void k_load(const uint32_t* __restrict__ m /*[32][ld]*/, uint32_t ld, const uint32_t* __restrict__ x /*[n][7]*/,
                                                   uint32_t n, uint32_t n_lv, uint32_t* err) {
    __shared__ uint32_t s_cnt[PARTS];
    const uint32_t l = blockIdx.x % n_lv, sb = blockIdx.x / n_lv, b0 = sb * BS;
    const uint32_t* d0p = m + (size_t)(2 * l) * ld;
    const uint32_t* d1p = d0p + ld;
    uint32_t nv[SPT][5], lv[SPT][5];
    auto fetch = [&](uint32_t rb0) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const uint32_t i = min(rb0 + s * THREADS + threadIdx.x, n - 1);
            const uint32_t* xp = x + (size_t)i * 7;
            nv[s][0] = d0p[i]; nv[s][1] = d1p[i]; nv[s][2] = xp[0]; nv[s][3] = xp[1]; nv[s][4] = xp[2];
        }
    };
    fetch(b0);
    uint32_t sink = 0;
    for (uint32_t r = 0; r < BS / (THREADS * SPT); ++r) {
        const uint32_t rb0 = b0 + r * THREADS * SPT;
        if (rb0 >= n) break;
#pragma unroll
        for (int s = 0; s < SPT; ++s)
#pragma unroll
            for (int k = 0; k < 5; ++k) lv[s][k] = nv[s][k];
        for (uint32_t p = threadIdx.x; p < PARTS; p += THREADS) s_cnt[p] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPT; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) sink += atomicAdd(&s_cnt[((lv[s][2] + c) * 2654435761u) >> 26], 1u);     // the ranking phase
        if (rb0 + THREADS * SPT < min(b0 + BS, n)) fetch(rb0 + THREADS * SPT);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const uint32_t i = min(rb0 + s * THREADS + threadIdx.x, n - 1);
            const uint32_t want[5] = {pat(2 * l, i), pat(2 * l + 1, i), pat(100, 7 * i), pat(100, 7 * i + 1), pat(100, 7 * i + 2)};
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (lv[s][k] != want[k]) {
                    if (atomicAdd(&err[k], 1u) == 0u && atomicAdd(&err[5], 1u) == 0u) {
                        err[6] = k * 16 + s; err[7] = threadIdx.x; err[8] = i; err[9] = lv[s][k]; err[10] = want[k]; err[11] = l;
                    }
                }
        }
        __syncthreads();
    }
    if (sink == 0xffffffffu) err[15] = 1;
}

int main(int argc, char** argv) {
    const int seconds = argc > 1 ? atoi(argv[1]) : 30;
    const uint32_t n = 262144, ld = 262144, n_lv = 13;
    uint32_t *m, *x, *e;
    CK(hipMalloc(&m, (size_t)32 * ld * 4)); CK(hipMalloc(&x, (size_t)n * 7 * 4)); CK(hipMalloc(&e, 64)); CK(hipMemset(e, 0, 64));
    uint32_t* h = (uint32_t*)malloc((size_t)32 * ld * 4);
    for (uint32_t r = 0; r < 32; ++r) for (uint32_t i = 0; i < ld; ++i) h[(size_t)r * ld + i] = pat(r, i);
    CK(hipMemcpy(m, h, (size_t)32 * ld * 4, hipMemcpyHostToDevice));
    for (uint32_t j = 0; j < n * 7; ++j) h[j] = pat(100, j);
    CK(hipMemcpy(x, h, (size_t)n * 7 * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventRecord(a, 0));
    long launches = 0; float ms = 0;
    while (ms < seconds * 1000.0f) {
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_load, dim3(n_lv * (n / BS)), dim3(THREADS), 0, 0, m, ld, x, n, n_lv, e);
        launches += 20;
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    uint32_t r[16]; CK(hipMemcpy(r, e, 64, hipMemcpyDeviceToHost));
    printf("loads as k_scatter_bin3's fetch: %ld launches x %u workgroups: wrong values per slot [row 2l, row 2l+1, x, y, z] = %u %u %u %u %u\n",
           launches, n_lv * (n / BS), r[0], r[1], r[2], r[3], r[4]);
    if (r[5]) printf("   first: slot %u (sample %u of the thread) thread %u index %u level %u: got %08x want %08x   (pattern of index %u row %u is %08x)\n",
                     r[6] / 16, r[6] % 16, r[7], r[8], r[11], r[9], r[10], r[8], 2 * r[11], pat(2 * r[11], r[8]));
    return 0;
}
