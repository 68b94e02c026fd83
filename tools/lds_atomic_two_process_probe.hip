// Which LDS atomic of the table scatter goes wrong when two PROCESSES share the GPU (profiles/r05_two_processes_one_gpu_scatter_probe.txt:
// ~1 launch in 600 wrong by <= 64 table floats of a level, spread over the WHOLE level = over many accumulate workgroups)?
// Two self-checking kernels with the scatter's own atomic sequences and 2^13-entry partitions:
//   rank : k_scatter_bin3's ranking -- every thread takes ranks with RETURNING u32 LDS atomics (ds_add_rtn_u32) on 64 partition counters,
//          the workgroup places one word per (partition, rank) slot, then checks that every slot was written exactly once and that the
//          counters equal the number of atomics issued.  A replayed or dropped atomic shows as a hole / a double slot / a wrong count.
//   f64  : k_scatter_accum3's accumulation -- RETURNLESS ds_add_f64 of small integers (exact in any order) into 2^13 x 2 doubles, every
//          grand total then compared with its closed form (all addends positive: one lost or doubled instruction moves it).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe tools/lds_atomic_two_process_probe.hip ; run one copy, then two side by side,
//   then one copy with `streams` = 2 (two queues of ONE process, what RCCL beside the step is).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define PARTS 64
#define THREADS 512
#define IPT 8                    // ranks per thread and round (the bin kernel: 2 samples x 4 corner pairs)
#define ROUNDS 16

__global__ __launch_bounds__(THREADS) void k_rank(uint32_t seed, uint32_t* err /*[0] holes, [1] doubles, [2] wrong counts*/) {
    __shared__ uint32_t s_cnt[PARTS], s_off[PARTS + 1], s_slot[THREADS * IPT];
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        for (uint32_t p = threadIdx.x; p < PARTS; p += THREADS) s_cnt[p] = 0;
        for (uint32_t q = threadIdx.x; q < THREADS * IPT; q += THREADS) s_slot[q] = 0;
        __syncthreads();
        uint32_t part[IPT], rank[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            part[k] = ((threadIdx.x * IPT + k + r * 977u + blockIdx.x * 131u + seed) * 2654435761u) >> 26;     // 64 partitions, hash-spread like the levels' items
            rank[k] = atomicAdd(&s_cnt[part[k]], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t s = 0; for (int p = 0; p < PARTS; ++p) { s_off[p] = s; s += s_cnt[p]; } s_off[PARTS] = s; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < IPT; ++k) { const uint32_t q = s_off[part[k]] + rank[k]; if (q < THREADS * IPT) atomicAdd(&s_slot[q], 1u); else atomicAdd(&err[1], 1u); }
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < THREADS * IPT; q += THREADS) { if (s_slot[q] == 0) atomicAdd(&err[0], 1u); else if (s_slot[q] > 1) atomicAdd(&err[1], 1u); }
        if (threadIdx.x == 0 && s_off[PARTS] != THREADS * IPT) atomicAdd(&err[2], 1u);
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_f64(uint32_t seed, uint32_t items_per_thread, double want, uint32_t* err /*[3] wrong totals*/) {
    extern __shared__ double s_acc[];                 // [8192][2]
    for (uint32_t q = threadIdx.x; q < 16384; q += 1024) s_acc[q] = 0.0;
    __syncthreads();
    // item j of the workgroup (j = t + 1024 i) adds (j & 7) + 1 to both features of entries e0(j), e1(j): returnless, four per item
    for (uint32_t i = 0; i < items_per_thread; ++i) {
        const uint32_t j = threadIdx.x + 1024u * i, h = (j + seed) * 2654435761u, e0 = h >> 19, e1 = (e0 + 1u) & 8191u;
        const double v = (double)((j & 7u) + 1u);
        atomicAdd(&s_acc[2 * e0], v); atomicAdd(&s_acc[2 * e0 + 1], 2.0 * v); atomicAdd(&s_acc[2 * e1], 3.0 * v); atomicAdd(&s_acc[2 * e1 + 1], 4.0 * v);
    }
    __syncthreads();
    // every value is a positive integer: a lost or doubled wave instruction moves the grand total (compared exactly)
    __shared__ double s_tot;
    if (threadIdx.x == 0) s_tot = 0.0;
    __syncthreads();
    double t = 0; for (uint32_t q = threadIdx.x; q < 16384; q += 1024) t += s_acc[q];
    atomicAdd(&s_tot, t);
    __syncthreads();
    if (threadIdx.x == 0 && s_tot != want) atomicAdd(&err[3], 1u);
}

int main(int argc, char** argv) {
    const int seconds = argc > 1 ? atoi(argv[1]) : 20, n_streams = argc > 2 ? atoi(argv[2]) : 1;
    uint32_t* d; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
    CK(hipFuncSetAttribute((const void*)k_f64, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t st[4]; for (int s = 0; s < n_streams; ++s) CK(hipStreamCreate(&st[s]));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double want = 0; for (uint32_t j = 0; j < 1024u * 64u; ++j) want += 10.0 * (double)((j & 7u) + 1u);
    for (int which = 0; which < 2; ++which) {
        long launches = 0; float ms = 0; uint32_t seed = 1;
        CK(hipEventRecord(a, st[0]));
        while (ms < seconds * 500.0f) {
            for (int r = 0; r < 10; ++r)
                for (int s = 0; s < n_streams; ++s, ++seed) {
                    if (which == 0) hipLaunchKernelGGL(k_rank, dim3(2048), dim3(THREADS), 0, st[s], seed, d);
                    else hipLaunchKernelGGL(k_f64, dim3(1024), dim3(1024), 131072, st[s], seed, 64u, want, d);
                }
            launches += 10 * n_streams;
            for (int s = 1; s < n_streams; ++s) CK(hipStreamSynchronize(st[s]));
            CK(hipEventRecord(b, st[0])); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        uint32_t h[4]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        if (which == 0) printf("rank  (returning ds_add_rtn_u32, %d queue%s): %ld launches x 2048 workgroups x %d rounds: %u holes, %u doubly-filled / out-of-range slots, %u wrong totals\n",
                               n_streams, n_streams > 1 ? "s" : "", launches, ROUNDS, h[0], h[1], h[2]);
        else printf("f64   (returnless ds_add_f64, %d queue%s):    %ld launches x 1024 workgroups x 262144 atomics: %u wrong totals\n", n_streams, n_streams > 1 ? "s" : "", launches, h[3]);
        fflush(stdout);
    }
    return 0;
}
