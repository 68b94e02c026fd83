// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP-on-CPU shim (see hip/hip_runtime.h).  Inside a workgroup every GPU thread
// is a fiber (own stack, hand-written x86-64 context switch); a fiber gives up the CPU only inside a barrier or a wave
// collective.  A collective completes when every lane of the wave that has not left the kernel has arrived; a lane that
// arrives early hands the CPU straight to the next runnable lane of ITS wave (one switch per lane and collective), and
// only when none is runnable back to the round-robin loop.  Workgroups run one after another on the calling thread, or,
// for entry points the caller declares free of inter-workgroup atomics (emu_set_threads), on several host threads --
// all scheduler state and the kernels' `__shared__` variables are thread-local.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <algorithm>
#include <cstdio>
#include <thread>
#include <vector>

namespace emu {
struct Wave {
    int active = 0, arrived = 0, gen = 0;
    // collectives are double-buffered by the lane's collective count: a lane can be at most one collective ahead of the
    // slowest lane of its wave (it cannot pass the next rendezvous before everybody has read this one)
    uint64_t slot[2][64];
    bool pred[2][64];
    float a[2][64], b[2][64];
    float a8[2][64][8], b8[2][64][8];
    int stamp[2][64];
};
// Context switch: callee-saved registers + stack pointer, x86-64 SysV (glibc's swapcontext makes a sigprocmask system
// call per switch).
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");

enum { WAIT_NONE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2 };
struct Lane {
    void* sp = nullptr;
    std::vector<char> stack;
    uint3e tid;
    int lane, wave;
    bool done = false;
    int wait_kind = WAIT_NONE, wait_gen = 0;
    int coll = 0;                           // collectives executed so far
};
struct State {
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    void* main_sp = nullptr;
    Lane* cur = nullptr;
    uint3e bid;
    int block_active = 0, block_arrived = 0, block_gen = 0, block_or = 0, block_or_result = 0;
    std::vector<char> dyn;
    unsigned n_threads = 0;                 // lanes of the running workgroup
};
static thread_local State S;
static dim3 g_bdim, g_gdim;                 // per launch, read-only while it runs
static const std::function<void()>* g_body = nullptr;
static int g_threads = 1;
static const size_t STACK = 128u << 10;

Lane* cur() { return S.cur; }
const uint3e& tid() { return S.cur->tid; }
const uint3e& bid() { return S.bid; }
const dim3& bdim() { return g_bdim; }
const dim3& gdim() { return g_gdim; }
int lane_id() { return S.cur->lane; }
void* dyn_lds() { return S.dyn.data(); }

static inline bool runnable(const Lane& l) {
    if (l.done) return false;
    if (l.wait_kind == WAIT_WAVE) return S.waves[l.wave].gen != l.wait_gen;
    if (l.wait_kind == WAIT_BLOCK) return S.block_gen != l.wait_gen;
    return true;
}
static inline void to_main() { emu_switch(&S.cur->sp, S.main_sp); }
static inline void to_lane(Lane* me, Lane* next) { S.cur = next; emu_switch(&me->sp, next->sp); }

static void wave_sync() {                       // all lanes of the wave that are still running
    Lane* me = S.cur;
    Wave& w = S.waves[me->wave];
    const int gen = w.gen;
    if (++w.arrived >= w.active) { w.arrived = 0; ++w.gen; return; }
    me->wait_kind = WAIT_WAVE; me->wait_gen = gen;
    while (w.gen == gen) {
        Lane* next = nullptr;
        Lane* base = &S.lanes[(size_t)me->wave * 64];
        const int n_in_wave = (int)std::min<unsigned>(64u, S.n_threads - (unsigned)me->wave * 64u);
        for (int k = 1; k < n_in_wave; ++k) {
            Lane* c = base + (me->lane + k) % n_in_wave;
            if (runnable(*c)) { next = c; break; }
        }
        if (next) to_lane(me, next); else to_main();
    }
    me->wait_kind = WAIT_NONE;
}

void wave_barrier() { wave_sync(); }

void block_barrier() {
    Lane* me = S.cur;
    const int gen = S.block_gen;
    if (++S.block_arrived >= S.block_active) {
        S.block_arrived = 0; S.block_or_result = S.block_or; S.block_or = 0; ++S.block_gen; return;
    }
    me->wait_kind = WAIT_BLOCK; me->wait_gen = gen;
    while (S.block_gen == gen) to_main();
    me->wait_kind = WAIT_NONE;
}
int block_or(int v) {
    if (v) S.block_or = 1;
    block_barrier();
    return S.block_or_result;
}

uint64_t wave_exchange(uint64_t mine, int src) {
    Lane* me = S.cur;
    Wave& w = S.waves[me->wave];
    const int id = ++me->coll, p = id & 1;
    w.slot[p][me->lane] = mine;
    w.stamp[p][me->lane] = id;
    wave_sync();
    return (src >= 0 && src < 64 && w.stamp[p][src] == id) ? w.slot[p][src] : mine;
}
uint64_t wave_ballot(bool pr) {
    Lane* me = S.cur;
    Wave& w = S.waves[me->wave];
    const int id = ++me->coll, p = id & 1;
    w.pred[p][me->lane] = pr;
    w.stamp[p][me->lane] = id;
    wave_sync();
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (w.stamp[p][l] == id && w.pred[p][l]) m |= 1ull << l;
    return m;
}
// v_mfma_f32_32x32x2_f32: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
// as an fmaf chain over k = 0, 1 (the hardware's exact-fp32 behaviour)
void wave_mfma_32x32x2(float a, float b, float* c16) {
    Lane* me = S.cur;
    Wave& w = S.waves[me->wave];
    const int l = me->lane, id = ++me->coll, p = id & 1;
    w.a[p][l] = a; w.b[p][l] = b; w.stamp[p][l] = id;
    wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c16[r];
        for (int k = 0; k < 2; ++k) {
            const float av = w.stamp[p][i + 32 * k] == id ? w.a[p][i + 32 * k] : 0.f;
            const float bv = w.stamp[p][j + 32 * k] == id ? w.b[p][j + 32 * k] : 0.f;
            acc = fmaf(av, bv, acc);
        }
        c16[r] = acc;
    }
}

// v_mfma_f32_32x32x16_f16: A[i = l & 31][k-slot (l >> 5, e)], B[k-slot (l >> 5, e)][j = l & 31], the C/D layout of the 32x32 family;
// operands arrive already widened to fp32 (products of two halves are exact there), the sum is taken in fp32
void wave_mfma_32x32x16(const float* a8, const float* b8, float* c16) {
    Lane* me = S.cur;
    Wave& w = S.waves[me->wave];
    const int l = me->lane, id = ++me->coll, p = id & 1;
    for (int e = 0; e < 8; ++e) { w.a8[p][l][e] = a8[e]; w.b8[p][l][e] = b8[e]; }
    w.stamp[p][l] = id;
    wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c16[r];
        for (int h = 0; h < 2; ++h) {
            const bool pa = w.stamp[p][i + 32 * h] == id, pb = w.stamp[p][j + 32 * h] == id;
            for (int e = 0; e < 8; ++e)
                acc += (pa ? w.a8[p][i + 32 * h][e] : 0.f) * (pb ? w.b8[p][j + 32 * h][e] : 0.f);
        }
        c16[r] = acc;
    }
}

static void lane_leaves(Lane* me) {
    me->done = true;
    Wave& w = S.waves[me->wave];
    --w.active;
    if (w.active > 0 && w.arrived >= w.active) { w.arrived = 0; ++w.gen; }          // the others were only waiting for us
    --S.block_active;
    if (S.block_active > 0 && S.block_arrived >= S.block_active) {
        S.block_arrived = 0; S.block_or_result = S.block_or; S.block_or = 0; ++S.block_gen;
    }
}
static void trampoline() {
    (*g_body)();
    lane_leaves(S.cur);
    for (;;) to_main();      // never resumed
}

static void run_block(unsigned threads, unsigned n_waves, dim3 block, unsigned bx, unsigned by, unsigned bz) {
    S.bid = uint3e{bx, by, bz};
    S.n_threads = threads;
    S.waves.assign(n_waves, Wave());
    for (auto& w : S.waves) for (int q = 0; q < 2; ++q) for (int i = 0; i < 64; ++i) w.stamp[q][i] = -1;
    S.block_active = (int)threads; S.block_arrived = 0; S.block_or = 0;
    for (unsigned t = 0; t < threads; ++t) {
        Lane& l = S.lanes[t];
        l.tid = uint3e{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        l.lane = (int)(t & 63); l.wave = (int)(t >> 6); l.done = false; l.wait_kind = WAIT_NONE; l.coll = 0;
        ++S.waves[l.wave].active;
        // initial frame: six zeroed callee-saved registers, then the entry address `ret` jumps to; the slot holding
        // that address is 16-byte aligned so that the trampoline starts with the ABI's rsp % 16 == 8
        uintptr_t top = ((uintptr_t)l.stack.data() + l.stack.size() - 64) & ~(uintptr_t)15;
        void** frame = (void**)top;
        frame[0] = (void*)&trampoline;
        frame[1] = nullptr;
        for (int r = 1; r <= 6; ++r) frame[-r] = nullptr;
        l.sp = (void*)(frame - 6);
    }
    bool any = true;
    while (any) {
        any = false;
        for (unsigned t = 0; t < threads; ++t) {
            if (S.lanes[t].done) continue;
            any = true;
            if (!runnable(S.lanes[t])) continue;
            S.cur = &S.lanes[t];
            emu_switch(&S.main_sp, S.lanes[t].sp);
        }
    }
    S.cur = nullptr;
}

static void run_range(unsigned threads, dim3 grid, dim3 block, size_t dyn_lds_bytes, unsigned first, unsigned step) {
    S.dyn.assign(dyn_lds_bytes + 64, 0);
    if (S.lanes.size() < threads) S.lanes.resize(threads);
    for (unsigned t = 0; t < threads; ++t) if (S.lanes[t].stack.size() < STACK) S.lanes[t].stack.resize(STACK);
    const unsigned n_waves = (threads + 63) / 64;
    const unsigned long long total = (unsigned long long)grid.x * grid.y * grid.z;
    for (unsigned long long b = first; b < total; b += step)
        run_block(threads, n_waves, block, (unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
}

void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const std::function<void()>& body) {
    const unsigned threads = block.x * block.y * block.z;
    g_body = &body;
    g_bdim = block; g_gdim = grid;
    const unsigned long long total = (unsigned long long)grid.x * grid.y * grid.z;
    const unsigned T = (unsigned)std::min<unsigned long long>((unsigned long long)std::max(g_threads, 1), total);
    if (T <= 1) { run_range(threads, grid, block, dyn_lds_bytes, 0, 1); return; }
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < T; ++k) pool.emplace_back([=]() { run_range(threads, grid, block, dyn_lds_bytes, k, T); });
    for (auto& th : pool) th.join();
}
}  // namespace emu

// host threads for the workgroups of the following launches of THIS library (1 = the calling thread).  Only for entry
// points without inter-workgroup atomics or ordering: the shim's atomics are plain read-modify-writes.
extern "C" void emu_set_threads(int n) { emu::g_threads = n < 1 ? 1 : n; }

// pieces of libxrnerf_mi355.so that live in other translation units
// (weak: the translation unit under test may be the one that defines them)
static char g_err[512];
__attribute__((weak)) void xr_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" __attribute__((weak)) const char* xr_last_error(void) { return g_err; }
extern "C" __attribute__((weak)) int xr_device_cus(void) { return 2; }
