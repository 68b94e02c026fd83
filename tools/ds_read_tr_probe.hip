// Lane mapping of gfx950's transposing LDS load ds_read_b64_tr_b16 (each lane supplies the address of 4 consecutive 16-bit
// elements and receives 4 elements): lane l's address is row l of a [64][64] array of element ids, so the output tells for
// every (lane, element) which lane's row and which of its 4 elements it came from.  Needed before the fused-MLP backward can
// stage its dW operands with vector stores and read them back transposed.
//   hipcc --offload-arch=gfx950 -O2 tools/ds_read_tr_probe.hip -o tools/ds_read_tr_probe && tools/ds_read_tr_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int mode) {
    __shared__ uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    // mode 0: lane l -> row l, elements 0..3;  mode 1: lane l -> row (l & 15), elements 4 (l >> 4) .. +3 (a 16 x 16 tile)
    const int row = mode == 0 ? threadIdx.x : (threadIdx.x & 15), c0 = mode == 0 ? 0 : 4 * (threadIdx.x >> 4);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + 64 * row + c0));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (uint16_t)v[e];
}
int main() {
    uint16_t* d = nullptr;
    if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) { printf("no device\n"); return 1; }
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        uint16_t h[256];
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        printf("mode %d: lane -> (source row, source column) of its 4 elements\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d: (%2u,%2u) (%2u,%2u) (%2u,%2u) (%2u,%2u)\n", l, h[4 * l] / 64, h[4 * l] % 64, h[4 * l + 1] / 64, h[4 * l + 1] % 64,
                   h[4 * l + 2] / 64, h[4 * l + 2] % 64, h[4 * l + 3] / 64, h[4 * l + 3] % 64);
    }
    return 0;
}
