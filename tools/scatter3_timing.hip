// NOTE (round 6): written against the two-kernel scatter of rounds 3-5 (k_scatter_accum3 / k_scatter_dense_rl as kernels of their own).  The
// library's accumulate launch is k_scatter_acc now (run-length + accumulate workgroups in one launch, integer LDS sums): this file builds
// against the history (git show abe25a6:xrnerf_amd/csrc/xr_scatter.hip), not against HEAD; its records are profiles/r03_* / r04_*.
// phase timing of the third-generation accumulate kernel (k_scatter_accum3<S3_LOG2, 1024>) on ray-like samples (20 consecutive steps of
// sqrt(3)/1024 per ray), all 16 levels of the Lego geometry: wall_clock64 (100 MHz) stamps of workgroups 0, 96, 192, ... + event
// times of the kernels.  build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -mllvm -simplifycfg-sink-common=false -DS3_TIMING \
//         tools/scatter3_timing.hip -o tools/scatter3_timing && tools/scatter3_timing [n]
#include "../xrnerf_amd/csrc/xr_scatter.hip"
#include <vector>
#include <cstdio>
#include <cmath>
void xr_set_error(const char*, ...) {}
extern "C" int xr_device_cus(void) { return 256; }
extern "C" void xr_hashgrid_meta(int n_levels, int log2_hashmap_size, int base_resolution, double per_level_scale, float* scale,
                                 uint32_t* resolution, uint32_t* offset) {
    const float log2b = log2f((float)per_level_scale);
    uint32_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float s = exp2f((float)l * log2b) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(s) + 1u;
        scale[l] = s; resolution[l] = res; offset[l] = off;
        const double cube = (double)res * res * res;
        uint32_t n = cube > 2147483647.0 ? 2147483647u : (uint32_t)cube;
        n = (n + 7u) / 8u * 8u;
        if (n > (1u << log2_hashmap_size)) n = 1u << log2_hashmap_size;
        off += n;
    }
    offset[n_levels] = off;
}
// co-run probe (argv[3] = 1): the accumulate kernel WITHOUT the update on one stream, a plain optimiser sweep over the same 12.2 M parameters
// (reads g p m v ema, writes p m v ema: 36 B per parameter) on a second stream, both started together: what an accumulate kernel whose
// update phase runs under the next partition's atomics could reach at best
__global__ __launch_bounds__(256) void k_adam_probe(const float4* __restrict__ g, float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                                    float4* __restrict__ e, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 G = g[i], P = p[i], M = m[i], V = v[i], E = e[i];
        auto one = [](float g_, float& p_, float& m_, float& v_, float& e_) {
            m_ = 0.9f * m_ + 0.1f * g_; v_ = 0.99f * v_ + 0.01f * g_ * g_;
            p_ -= 1e-2f * m_ / (sqrtf(v_) + 1e-15f); e_ = 0.95f * e_ + 0.05f * p_;
        };
        one(G.x, P.x, M.x, V.x, E.x); one(G.y, P.y, M.y, V.y, E.y); one(G.z, P.z, M.z, V.z, E.z); one(G.w, P.w, M.w, V.w, E.w);
        p[i] = P; m[i] = M; v[i] = V; e[i] = E;
    }
}
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 121776u;
    const bool fuse = argc > 2 && atoi(argv[2]) != 0;          // the optimiser update inside the accumulate kernel
    float scale[16]; uint32_t res[16], off[17];
    xr_hashgrid_meta(16, 19, 16, std::exp2(std::log2(2048.0 / 16) / 15), scale, res, off);
    GridMeta gm; uint32_t hm; fill_meta(&gm, &hm, 16, scale, res, off);
    std::vector<float> x((size_t)n * 3), d((size_t)32 * n);
    uint32_t h = 1; auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (h >> 8) * (1.f / 16777216.f); };
    float o[3] = {0, 0, 0}, dir[3] = {0, 0, 1};
    for (uint32_t i = 0; i < n; ++i) {
        if (i % 20 == 0) {
            float nn = 0;
            for (int k = 0; k < 3; ++k) { o[k] = 0.25f + 0.5f * rnd(); dir[k] = rnd() - 0.5f; nn += dir[k] * dir[k]; }
            nn = 1.f / sqrtf(nn + 1e-9f);
            for (int k = 0; k < 3; ++k) dir[k] *= nn;
        }
        for (int k = 0; k < 3; ++k) { float v = o[k] + 0.0016915f * (i % 20) * dir[k]; x[3 * (size_t)i + k] = v < 0 ? 0 : (v > 1 ? 1 : v); }
    }
    for (auto& v : d) v = rnd() - 0.5f;
    S3Layout P;
    if (!s3_layout(1u << 18, gm, hm, 1, &P)) { printf("no layout\n"); return 1; }
    printf("n %u  binned levels %u  accumulate workgroups %u  nsb %u  run-length levels %u\n", n, P.bin.n_lv, P.bin.acc_blocks, P.bin.nsb, P.rl.n_lv);
    float *dx, *dd, *tab; void* ws; uint32_t* ndev;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dd, d.size() * 4); hipMalloc(&tab, (size_t)off[16] * 8); hipMalloc(&ndev, 4);
    const size_t wsb = P.counts_bytes + P.bins_bytes + P.ovf_bytes + P.slabs_bytes;
    hipMalloc(&ws, wsb);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ndev, &n, 4, hipMemcpyHostToDevice);
    uint32_t* counts = (uint32_t*)ws;
    float4* bins = (float4*)((char*)ws + P.counts_bytes);
    float4* ovf = (float4*)((char*)ws + P.counts_bytes + P.bins_bytes);
    if (fuse) {
        float* st[4];
        for (auto& q : st) { hipMalloc(&q, (size_t)off[16] * 8); hipMemset(q, 0, (size_t)off[16] * 8); }
        P.bin.fuse = 1u;
        P.bin.ad = XrAdamArgs{st[0], st[1], st[2], st[3], 0.9f, 0.99f, 1e-2f, 0.1f, 1e-15f, 1e-6f, 0.05f, 1.f};
        printf("fused optimiser update ON\n");
    }
    hipFuncSetAttribute((const void*)k_scatter_accum3<S3_LOG2, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS_BYTES);
    hipFuncSetAttribute((const void*)k_scatter_accum3<S3_LOG2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS_BYTES);
    hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
    // one workgroup per partition, 1024 threads, then 512 (the persistent forms of round 4: commit "Scatter: persistent accumulate kernel")
    struct Form { const char* name; int kind, threads; } forms[] = {{"accum3 x1024", 3, 1024}, {"accum3 x512", 3, 512}};
    for (const Form& f : forms) {
        float best1 = 1e9f, best2 = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            S3Plan pb = P.bin;
            hipEventRecord(a);
            hipLaunchKernelGGL(k_scatter_bin3<2048>, dim3(pb.n_lv * pb.nsb), dim3(S3_BIN_THREADS), 0, 0, pb, dx, 3u, dd, n, 1u << 18, ndev,
                               (const uint32_t*)nullptr, counts, bins, ovf);
            hipEventRecord(b);
            if (f.kind == 3 && f.threads == 1024)
                hipLaunchKernelGGL((k_scatter_accum3<S3_LOG2, 1024>), dim3(pb.acc_blocks), dim3(1024), S3_LDS_BYTES, 0, pb, (const uint32_t*)counts,
                                   (const float4*)bins, (const float4*)ovf, tab);
            else
                hipLaunchKernelGGL((k_scatter_accum3<S3_LOG2, 512>), dim3(pb.acc_blocks), dim3(512), S3_LDS_BYTES, 0, pb, (const uint32_t*)counts,
                                   (const float4*)bins, (const float4*)ovf, tab);
            hipEventRecord(c); hipEventSynchronize(c);
            float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, b, c);
            if (rep >= 2) { best1 = m1 < best1 ? m1 : best1; best2 = m2 < best2 ? m2 : best2; }
        }
        printf("%-16s bin3 %.1f us  accumulate %.1f us (best of 4)\n", f.name, best1 * 1e3, best2 * 1e3);
#ifdef S3_TIMING
        if (f.kind == 3 && f.threads == 1024) {
            long long t[S3_T_BLOCKS][8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_s3_t), sizeof(t));
            long long t0 = t[0][0];
            for (int w = 0; w < S3_T_BLOCKS; ++w) {
                auto us = [&](int i, int j) { return (t[w][j] - t[w][i]) / 100.0; };
                printf("  wg %3d: start +%6.2f | fills %.2f  first fetch issued %.2f  zero+sync %.2f  loop %.2f  overflow check %.2f+sync  write %.2f | whole %.2f us\n",
                       w * 96, (t[w][0] - t0) / 100.0, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(0, 6));
            }
        }
#endif
    }
    if (argc > 3 && atoi(argv[3]) != 0 && !fuse) {
        float4* st[5]; const size_t n4 = (size_t)off[16] * 2 / 4;
        for (auto& q : st) { hipMalloc(&q, n4 * 16); hipMemset(q, 0, n4 * 16); }
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipEvent_t e0, e1, e2, f1, f2; for (auto* ev : {&e0, &e1, &e2, &f1, &f2}) hipEventCreate(ev);
        S3Plan pb = P.bin;
        for (int grid : {512, 1024, 2048, 8192}) {
            float alone_acc = 1e9f, alone_adam = 1e9f, both = 1e9f, both_acc = 1e9f, both_adam = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                for (int mode = 0; mode < 3; ++mode) {          // 0: accumulate alone, 1: sweep alone, 2: together
                    hipDeviceSynchronize();
                    hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
                    hipEventRecord(f1, s2);
                    if (mode != 1) hipLaunchKernelGGL((k_scatter_accum3<S3_LOG2, 512>), dim3(pb.acc_blocks), dim3(512), S3_LDS_BYTES, s1, pb, (const uint32_t*)counts,
                                                      (const float4*)bins, (const float4*)ovf, tab);
                    hipEventRecord(e1, s1);
                    if (mode != 0) hipLaunchKernelGGL(k_adam_probe, dim3(grid), dim3(256), 0, s2, (const float4*)st[0], st[1], st[2], st[3], st[4], n4);
                    hipEventRecord(f2, s2);
                    hipStreamWaitEvent(s1, f2, 0); hipEventRecord(e2, s1); hipEventSynchronize(e2);
                    float ta, tb, tw; hipEventElapsedTime(&ta, e0, e1); hipEventElapsedTime(&tb, f1, f2); hipEventElapsedTime(&tw, e0, e2);
                    if (rep < 2) continue;
                    if (mode == 0) alone_acc = fminf(alone_acc, ta);
                    if (mode == 1) alone_adam = fminf(alone_adam, tb);
                    if (mode == 2) { both = fminf(both, tw); both_acc = fminf(both_acc, ta); both_adam = fminf(both_adam, tb); }
                }
            }
            printf("co-run, sweep grid %5d x 256: accumulate alone %.1f us  sweep alone %.1f us  together %.1f us (accumulate %.1f, sweep %.1f)\n", grid,
                   alone_acc * 1e3, alone_adam * 1e3, both * 1e3, both_acc * 1e3, both_adam * 1e3);
        }
    }
    return 0;
}
