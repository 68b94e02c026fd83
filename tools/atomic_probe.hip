// micro-probe: throughput of scattered global atomics on MI355X (which knobs matter?)
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// kind: 0 f32 add, 1 u64 add, 2 pk_f16 add, 3 f32 x2 adjacent (pair), 4 f64 add
template <int KIND, bool PARTITION>
__global__ void probe(void* buf, uint32_t entries_mask, uint32_t per_thread) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t xcd = blockIdx.x & 7;
    for (uint32_t k = 0; k < per_thread; ++k) {
        uint32_t h = hash32(tid * 977u + k * 0x9e3779b9u);
        uint32_t e = h & entries_mask;                       // entry of 8 bytes
        if (PARTITION) e = (e & (entries_mask >> 3)) | (xcd * ((entries_mask + 1) >> 3));   // each XCD its own eighth
        if (KIND == 0) unsafeAtomicAdd((float*)buf + 2 * (size_t)e, 1.0f);
        else if (KIND == 1) atomicAdd((unsigned long long*)buf + e, 0x0000000100000001ull);
        else if (KIND == 2) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 v = {(_Float16)1.0f, (_Float16)1.0f};
            __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)((uint32_t*)buf + 2 * (size_t)e), v);
        } else if (KIND == 3) { unsafeAtomicAdd((float*)buf + 2 * (size_t)e, 1.0f); unsafeAtomicAdd((float*)buf + 2 * (size_t)e + 1, 1.0f); }
        else if (KIND == 4) unsafeAtomicAdd((double*)buf + e, 1.0);
        else if (KIND >= 5) {
            // G adjacent lanes hit G adjacent dwords of one random G*4-byte aligned chunk
            const uint32_t G = KIND == 5 ? 2 : (KIND == 6 ? 4 : 16);
            uint32_t hh = hash32((tid / G) * 977u + k * 0x9e3779b9u);
            size_t chunk = hh & (((entries_mask + 1) * 2 / G) - 1);
            unsafeAtomicAdd((float*)buf + chunk * G + (tid % G), 1.0f);
        }
    }
}
template <int KIND, bool PART>
float run(void* buf, size_t bytes, uint32_t entries, int blocks, uint32_t per_thread) {
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<KIND, PART><<<blocks, 256>>>(buf, entries - 1, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<KIND, PART><<<blocks, 256>>>(buf, entries - 1, per_thread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const size_t bytes = 256ull << 20;
    void* buf; hipMalloc(&buf, bytes);
    const int blocks = 8192; const uint32_t per = 64;
    const double ops = (double)blocks * 256 * per;
    const char* names[] = {"f32", "u64", "pk_f16", "f32 pair", "f64"};
    for (uint32_t log_e : {12u, 16u, 19u, 22u, 24u}) {   // entries of 8 B: 32 KB, 512 KB, 4 MB, 32 MB, 128 MB
        uint32_t entries = 1u << log_e;
        float t0 = run<0, false>(buf, bytes, entries, blocks, per), t1 = run<1, false>(buf, bytes, entries, blocks, per);
        float t2 = run<2, false>(buf, bytes, entries, blocks, per), t3 = run<3, false>(buf, bytes, entries, blocks, per);
        float t4 = run<4, false>(buf, bytes, entries, blocks, per);
        float p0 = run<0, true>(buf, bytes, entries, blocks, per), p1 = run<1, true>(buf, bytes, entries, blocks, per);
        float a5 = run<5, false>(buf, bytes, entries, blocks, per), a6 = run<6, false>(buf, bytes, entries, blocks, per), a7 = run<7, false>(buf, bytes, entries, blocks, per);
        printf("   adjacent-lane f32: pairs %.1f  quads %.1f  16-lane lines %.1f Gops/s\n", ops / a5 / 1e6, ops / a6 / 1e6, ops / a7 / 1e6);
        printf("region %8.1f KB: Gops/s  f32 %.1f  u64 %.1f  pk_f16 %.1f  f32pair(ops counted x2) %.1f  f64 %.1f | XCD-partitioned f32 %.1f u64 %.1f\n",
               entries * 8 / 1024.0, ops / t0 / 1e6, ops / t1 / 1e6, ops / t2 / 1e6, 2 * ops / t3 / 1e6, ops / t4 / 1e6, ops / p0 / 1e6, ops / p1 / 1e6);
    }
    return 0;
}
