// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP-on-CPU shim (see hip/hip_runtime.h).  Workgroups run one after another;
// inside a workgroup every GPU thread is a ucontext fiber, resumed round-robin; a fiber yields only inside a barrier or a
// wave collective.  A collective completes when every lane of the wave that has not left the kernel has arrived.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <cstdarg>
#include <cstdio>
#include <vector>

namespace emu {
struct Wave {
    int active = 0, arrived = 0, gen = 0;
    uint64_t slot[64];
    bool pred[64];
    float a[64], b[64];
    bool present[64];
};
struct Lane {
    ucontext_t ctx;
    std::vector<char> stack;
    uint3e tid;
    int lane, wave;
    bool done = false;
};
static std::vector<Lane> g_lanes;
static std::vector<Wave> g_waves;
static ucontext_t g_main;
static Lane* g_cur = nullptr;
static uint3e g_bid;
static dim3 g_bdim, g_gdim;
static int g_block_active = 0, g_block_arrived = 0, g_block_gen = 0, g_block_or = 0, g_block_or_result = 0;
static const std::function<void()>* g_body = nullptr;
static std::vector<char> g_dyn;

Lane* cur() { return g_cur; }
const uint3e& tid() { return g_cur->tid; }
const uint3e& bid() { return g_bid; }
const dim3& bdim() { return g_bdim; }
const dim3& gdim() { return g_gdim; }
int lane_id() { return g_cur->lane; }
void* dyn_lds() { return g_dyn.data(); }

static void yield() { swapcontext(&g_cur->ctx, &g_main); }

static void wave_sync() {                       // all lanes of the wave that are still running
    Wave& w = g_waves[g_cur->wave];
    const int gen = w.gen;
    if (++w.arrived >= w.active) { w.arrived = 0; ++w.gen; return; }
    while (w.gen == gen) yield();
}

void wave_barrier() { wave_sync(); }

void block_barrier() {
    const int gen = g_block_gen;
    if (++g_block_arrived >= g_block_active) { g_block_arrived = 0; g_block_or_result = g_block_or; g_block_or = 0; ++g_block_gen; return; }
    while (g_block_gen == gen) yield();
}
int block_or(int v) {
    if (v) g_block_or = 1;
    block_barrier();
    return g_block_or_result;
}

uint64_t wave_exchange(uint64_t mine, int src) {
    Wave& w = g_waves[g_cur->wave];
    w.slot[g_cur->lane] = mine;
    w.present[g_cur->lane] = true;
    wave_sync();
    const uint64_t v = (src >= 0 && src < 64 && w.present[src]) ? w.slot[src] : mine;
    wave_sync();                                 // nobody overwrites a slot before everybody has read
    w.present[g_cur->lane] = false;
    return v;
}
uint64_t wave_ballot(bool p) {
    Wave& w = g_waves[g_cur->wave];
    w.pred[g_cur->lane] = p;
    w.present[g_cur->lane] = true;
    wave_sync();
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (w.present[l] && w.pred[l]) m |= 1ull << l;
    wave_sync();
    w.present[g_cur->lane] = false;
    return m;
}
// v_mfma_f32_32x32x2_f32: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
// as an fmaf chain over k = 0, 1 (the hardware's exact-fp32 behaviour)
void wave_mfma_32x32x2(float a, float b, float* c16) {
    Wave& w = g_waves[g_cur->wave];
    const int l = g_cur->lane;
    w.a[l] = a; w.b[l] = b; w.present[l] = true;
    wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c16[r];
        for (int k = 0; k < 2; ++k) {
            const float av = w.present[i + 32 * k] ? w.a[i + 32 * k] : 0.f;
            const float bv = w.present[j + 32 * k] ? w.b[j + 32 * k] : 0.f;
            acc = fmaf(av, bv, acc);
        }
        c16[r] = acc;
    }
    wave_sync();
    w.present[l] = false;
}

static void lane_leaves(Lane* me) {
    me->done = true;
    Wave& w = g_waves[me->wave];
    --w.active;
    if (w.active > 0 && w.arrived >= w.active) { w.arrived = 0; ++w.gen; }          // the others were only waiting for us
    --g_block_active;
    if (g_block_active > 0 && g_block_arrived >= g_block_active) {
        g_block_arrived = 0; g_block_or_result = g_block_or; g_block_or = 0; ++g_block_gen;
    }
}
static void trampoline() {
    (*g_body)();
    lane_leaves(g_cur);
    swapcontext(&g_cur->ctx, &g_main);
}

void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const std::function<void()>& body) {
    const unsigned threads = block.x * block.y * block.z;
    g_body = &body;
    g_bdim = block; g_gdim = grid;
    g_dyn.assign(dyn_lds_bytes + 64, 0);
    g_lanes.resize(threads);
    for (auto& l : g_lanes) if (l.stack.size() < (256u << 10)) l.stack.resize(256u << 10);
    const unsigned n_waves = (threads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_bid = uint3e{bx, by, bz};
                g_waves.assign(n_waves, Wave());
                g_block_active = (int)threads; g_block_arrived = 0; g_block_or = 0;
                for (unsigned t = 0; t < threads; ++t) {
                    Lane& l = g_lanes[t];
                    l.tid = uint3e{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    l.lane = (int)(t & 63); l.wave = (int)(t >> 6); l.done = false;
                    ++g_waves[l.wave].active;
                    getcontext(&l.ctx);
                    l.ctx.uc_stack.ss_sp = l.stack.data();
                    l.ctx.uc_stack.ss_size = l.stack.size();
                    l.ctx.uc_link = &g_main;
                    makecontext(&l.ctx, trampoline, 0);
                }
                for (auto& w : g_waves) for (int i = 0; i < 64; ++i) w.present[i] = false;
                bool any = true;
                while (any) {
                    any = false;
                    for (unsigned t = 0; t < threads; ++t) {
                        if (g_lanes[t].done) continue;
                        any = true;
                        g_cur = &g_lanes[t];
                        swapcontext(&g_main, &g_lanes[t].ctx);
                    }
                }
            }
    g_cur = nullptr;
}
}  // namespace emu

// pieces of libxrnerf_mi355.so that live in other translation units
// (weak: the translation unit under test may be the one that defines them)
static char g_err[512];
__attribute__((weak)) void xr_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" __attribute__((weak)) const char* xr_last_error(void) { return g_err; }
extern "C" __attribute__((weak)) int xr_device_cus(void) { return 2; }
