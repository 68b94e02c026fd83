// What does a cross-stream ordering point cost on the issuing queue?  A chain of dependent ~20-us kernels on stream A, with, between
// consecutive kernels: nothing / hipEventRecord / hipStreamWaitEvent on an event of stream B that completed long ago / record + a wait by
// stream B (a fork) / a fork and a join (B runs a short kernel in between).  Time per link of the chain, event-timed over 200 links.
//   hipcc --offload-arch=gfx950 -O2 tools/event_cost_probe.hip -o tools/event_cost_probe && tools/event_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4); hipMemset(d, 0, 4096 * 256 * 4);
    hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    hipEvent_t t0, t1, old, ev[256], ev2[256];
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipEventCreateWithFlags(&old, hipEventDisableTiming);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, B, d + 64, 10); hipEventRecord(old, B); hipDeviceSynchronize();
    const int links = 200;
    const char* names[] = {"plain chain", "+ hipEventRecord", "+ wait on a long-completed event", "+ fork (record, other stream waits and runs a kernel)",
                           "+ fork and join (wait for the other stream's kernel)", "+ 2 records + 2 waits (satisfied)"};
    for (int spin_n : {2000, 20000}) {
        float base = 0;
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipDeviceSynchronize();
                hipEventRecord(t0, A);
                for (int i = 0; i < links; ++i) {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, A, d, spin_n);
                    if (mode == 1) hipEventRecord(ev[i], A);
                    if (mode == 2) hipStreamWaitEvent(A, old, 0);
                    if (mode == 3 || mode == 4) { hipEventRecord(ev[i], A); hipStreamWaitEvent(B, ev[i], 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, B, d + 4096 * 128, 10); }
                    if (mode == 4) { hipEventRecord(ev2[i], B); hipStreamWaitEvent(A, ev2[i], 0); }
                    if (mode == 5) { hipEventRecord(ev[i], A); hipStreamWaitEvent(A, old, 0); hipEventRecord(ev2[i], A); hipStreamWaitEvent(A, old, 0); }
                }
                hipEventRecord(t1, A); hipEventSynchronize(t1);
                float ms; hipEventElapsedTime(&ms, t0, t1);
                if (ms < best) best = ms;
            }
            const float us = best * 1e3f / links;
            if (mode == 0) base = us;
            printf("kernel %5d iterations: %-62s %7.2f us per link (%+.2f)\n", spin_n, names[mode], us, us - base);
        }
    }
    return 0;
}
