// LDS read-modify-write throughput probe for the scatter accumulate design (MI355X).
// One workgroup of 1024 threads per CU, 128 KiB of LDS, R rounds of 4 operations per thread on
// pseudo-random (or lane-linear) addresses.  Prints ns per wave-level instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N_F (32768)
template <int MODE, int LINEAR>
__global__ __launch_bounds__(1024) void k(uint32_t rounds, float* out) {
    extern __shared__ float s[];
    for (uint32_t e = threadIdx.x; e < N_F; e += 1024) s[e] = 0.f;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = h * 1664525u + 1013904223u;
            const uint32_t a = LINEAR ? ((threadIdx.x & 63) + 64 * ((h >> 8) & 511)) & (N_F - 1) : (h >> 9) & (N_F - 1);
            if (MODE == 0) atomicAdd(&s[a], 1.0f);                                         // ds_add_f32
            else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(s) + a, 1u);        // ds_add_u32
            else if (MODE == 2) { float v = s[a]; s[a] = v + 1.0f; }                       // plain RMW (racy: probe only)
            else if (MODE == 3) acc += __uint_as_float(atomicAdd(reinterpret_cast<uint32_t*>(s) + a, 1u));   // returning
            else if (MODE == 4) { float2* p = reinterpret_cast<float2*>(s) + (a >> 1); float2 v = *p; v.x += 1.f; v.y += 2.f; *p = v; }   // 8-byte RMW
            else if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (a >> 1), 1ull);        // ds_add_u64
            else if (MODE == 6) { s[a] = 1.0f; }                                           // plain store
            else if (MODE == 7) atomicAdd(reinterpret_cast<double*>(s) + (a >> 1), 1.0);                       // ds_add_f64
            else if (MODE == 8) { unsigned long long* p = reinterpret_cast<unsigned long long*>(s) + (a >> 1); acc += (float)atomicCAS(p, 0ull, (unsigned long long)h); }   // ds_cmpst_rtn_b64
            else if (MODE == 9) { uint32_t* p = reinterpret_cast<uint32_t*>(s) + a; acc += (float)atomicCAS(p, 0u, h); }       // ds_cmpst_rtn_b32
            else if (MODE == 10) atomicMax(reinterpret_cast<uint32_t*>(s) + a, h);                           // ds_max_u32
            else if (MODE == 11) { unsigned long long* p = reinterpret_cast<unsigned long long*>(s) + (a >> 1); unsigned long long v = *p; *p = v + h; }     // 8-byte integer RMW
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5] + acc;
}
template <int MODE, int LINEAR>
static void run(const char* name) {
    float* out; hipMalloc(&out, 4096);
    hipFuncSetAttribute((const void*)k<MODE, LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, N_F * 4);
    const uint32_t rounds = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, LINEAR>), dim3(256), dim3(1024), N_F * 4, 0, rounds, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, LINEAR>), dim3(256), dim3(1024), N_F * 4, 0, rounds, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double wave_instr_per_cu = 16.0 * rounds * 4;
    printf("%-34s %8.3f ms  %7.1f ns / wave-instr / CU  (%.2f lane-ops/ns/CU)\n", name, ms, ms * 1e6 / wave_instr_per_cu, 64.0 * wave_instr_per_cu / (ms * 1e6));
    hipFree(out);
}
int main() {
    run<0, 0>("ds_add_f32 random");
    run<0, 1>("ds_add_f32 lane-linear");
    run<1, 0>("ds_add_u32 random");
    run<1, 1>("ds_add_u32 lane-linear");
    run<3, 0>("ds_add_rtn_u32 random");
    run<5, 0>("ds_add_u64 random");
    run<2, 0>("plain 4B read+write random");
    run<2, 1>("plain 4B read+write lane-linear");
    run<4, 0>("plain 8B read+write random");
    run<6, 0>("plain 4B store random");
    run<7, 0>("ds_add_f64 random");
    run<8, 0>("ds_cmpst_rtn_b64 random");
    run<9, 0>("ds_cmpst_rtn_b32 random");
    run<10, 0>("ds_max_u32 random");
    run<11, 0>("plain 8B int read+write random");
    return 0;
}
