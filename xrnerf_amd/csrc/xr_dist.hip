// Gradient exchange of the data-parallel training step from native code: RCCL (backend "nccl" on ROCm: rings / trees over xGMI)
// driven directly, so that the loop between two grid refreshes (xr_ngp_loop_run) serves N > 1 ranks without returning to the
// interpreter once per iteration.  The reference gets the exchange implicitly from MMDistributedDataParallel
// (/root/reference/xrnerf/core/apis/train.py:28-38): an all-reduce of every parameter's gradient behind the backward pass.
//
// The communicator is created from a unique id the CALLER distributes (torch.distributed's store / a broadcast): nothing here
// talks to the network before that.  librccl is dlopen'ed on first use -- a single-GPU process never loads it.
// Every collective is enqueued on the communicator's own stream, ordered behind the point of the caller's stream at which the
// gradient bucket is complete (an event), so it runs under whatever the caller enqueues next; xr_rccl `finish` orders the
// caller's stream behind all of them.
#include "xr_common.h"
#include <dlfcn.h>
#include <mutex>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers still builds the library (a single-GPU process never calls in here): the few
// types this file hands through are NCCL's stable ABI -- a 128-byte id, an opaque communicator, the float / sum enumerators.
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;

std::mutex g_api_lock;
int load_rccl(const char* path) {
    std::lock_guard<std::mutex> hold(g_api_lock);       // (two threads creating their communicators at once)
    if (g_api.lib) return XR_OK;
    // a copy the process has already mapped (torch.distributed's) is reused; else the given path, else the system library
    void* h = nullptr;
    const char* names[] = {path, "librccl.so.1", "librccl.so"};
    for (const char* nm : names) if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char* nm : names) if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { xr_set_error("xr_rccl: cannot load librccl (%s)", dlerror()); return XR_EHIP; }
    RcclApi api;                                        // filled completely before it is published: a missing symbol leaves g_api untouched
#define XR_SYM(field, name) \
    *(void**)(&api.field) = dlsym(h, name); \
    if (!api.field) { xr_set_error("xr_rccl: librccl has no %s", name); return XR_EHIP; }
    XR_SYM(GetUniqueId, "ncclGetUniqueId") XR_SYM(CommInitRank, "ncclCommInitRank") XR_SYM(CommDestroy, "ncclCommDestroy")
    XR_SYM(AllReduce, "ncclAllReduce") XR_SYM(ReduceScatter, "ncclReduceScatter") XR_SYM(AllGather, "ncclAllGather")
    XR_SYM(GetErrorString, "ncclGetErrorString")
#undef XR_SYM
    api.lib = h;
    g_api = api;
    return XR_OK;
}

struct XrRccl {
    ncclComm_t comm = nullptr;
    hipStream_t cs = nullptr;                 // the collectives' own stream
    hipEvent_t ready = nullptr, done = nullptr;
    int world = 1, rank = 0;
    bool pending = false;
    // how long the caller's stream waits in `finish` (what the exchange adds to a step after everything that overlapped): a ring of
    // event pairs recorded around the wait while timing is on
    static constexpr int RING = 64;
    hipEvent_t t0[RING] = {}, t1[RING] = {};
    bool timing = false;
    unsigned n_timed = 0;
};
#define XR_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { xr_set_error("%s: %s", __func__, g_api.GetErrorString(r_)); return XR_EHIP; } } while (0)

// order the communicator's stream behind `stream` as it stands now
int fork_from(XrRccl* C, void* stream) {
    XR_HIP(hipEventRecord(C->ready, (hipStream_t)stream));
    XR_HIP(hipStreamWaitEvent(C->cs, C->ready, 0));
    C->pending = true;
    return XR_OK;
}
int ex_all_reduce(void* ctx, float* buf, size_t n, void* stream) {
    XrRccl* C = (XrRccl*)ctx;
    int rc = fork_from(C, stream);
    if (rc != XR_OK) return rc;
    XR_NCCL(g_api.AllReduce(buf, buf, n, ncclFloat, ncclSum, C->comm, C->cs));
    return XR_OK;
}
int ex_reduce_scatter(void* ctx, const float* send, float* recv, size_t n_recv, void* stream) {
    XrRccl* C = (XrRccl*)ctx;
    int rc = fork_from(C, stream);
    if (rc != XR_OK) return rc;
    XR_NCCL(g_api.ReduceScatter(send, recv, n_recv, ncclFloat, ncclSum, C->comm, C->cs));
    return XR_OK;
}
int ex_all_gather(void* ctx, const float* send, float* recv, size_t n_send, void* stream) {
    XrRccl* C = (XrRccl*)ctx;
    int rc = fork_from(C, stream);
    if (rc != XR_OK) return rc;
    XR_NCCL(g_api.AllGather(send, recv, n_send, ncclFloat, C->comm, C->cs));
    return XR_OK;
}
int ex_finish(void* ctx, void* stream) {
    XrRccl* C = (XrRccl*)ctx;
    if (!C->pending) return XR_OK;
    XR_HIP(hipEventRecord(C->done, C->cs));
    const unsigned slot = C->n_timed % XrRccl::RING;
    if (C->timing) XR_HIP(hipEventRecord(C->t0[slot], (hipStream_t)stream));
    XR_HIP(hipStreamWaitEvent((hipStream_t)stream, C->done, 0));
    if (C->timing) { XR_HIP(hipEventRecord(C->t1[slot], (hipStream_t)stream)); C->n_timed += 1; }
    C->pending = false;
    return XR_OK;
}
}  // namespace

// 128 bytes for ncclCommInitRank: generated on ONE rank, handed to the others by the caller
extern "C" int xr_rccl_unique_id(const char* librccl_path, void* id128) {
    XR_REQUIRE(id128, "null pointer");
    int rc = load_rccl(librccl_path);
    if (rc != XR_OK) return rc;
    ncclUniqueId id;
    XR_NCCL(g_api.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return XR_OK;
}
// collective call (every rank of the job): a communicator over the current device, its stream and two events.  The one place of this
// library that creates a handle: communicators cannot be caller-provided memory.
extern "C" int xr_rccl_destroy(void* handle);
extern "C" void* xr_rccl_create(const char* librccl_path, const void* id128, int world_size, int rank) {
    if (!id128 || world_size < 1 || rank < 0 || rank >= world_size) { xr_set_error("xr_rccl_create: bad argument"); return nullptr; }
    if (load_rccl(librccl_path) != XR_OK) return nullptr;
    XrRccl* C = new XrRccl();
    C->world = world_size; C->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_api.CommInitRank(&C->comm, world_size, id, rank);
    if (r != ncclSuccess) { xr_set_error("xr_rccl_create: %s", g_api.GetErrorString(r)); delete C; return nullptr; }
    if (hipStreamCreateWithFlags(&C->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&C->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&C->done, hipEventDisableTiming) != hipSuccess) {
        xr_set_error("xr_rccl_create: cannot create the stream / events");
        (void)xr_rccl_destroy(C);                       // releases whatever of the three exists, and the communicator
        return nullptr;
    }
    return C;
}
extern "C" int xr_rccl_destroy(void* handle) {
    if (!handle) return XR_OK;
    XrRccl* C = (XrRccl*)handle;
    if (C->cs) (void)hipStreamSynchronize(C->cs);
    if (C->comm) (void)g_api.CommDestroy(C->comm);
    if (C->ready) (void)hipEventDestroy(C->ready);
    if (C->done) (void)hipEventDestroy(C->done);
    for (int i = 0; i < XrRccl::RING; ++i) { if (C->t0[i]) (void)hipEventDestroy(C->t0[i]); if (C->t1[i]) (void)hipEventDestroy(C->t1[i]); }
    if (C->cs) (void)hipStreamDestroy(C->cs);
    delete C;
    return XR_OK;
}
// measured exposure.  Reports (mean_ms / max_ms / count non-null; after the caller synchronised the device) the mean / max wait of the last
// <= 64 finishes that waited for something and how many there were, THEN sets the recording state: timing_on != 0 starts a new record
// (events created at the first use), 0 stops recording.
extern "C" int xr_rccl_exposed_ms(void* handle, int timing_on, float* mean_ms, float* max_ms, int* count) {
    XR_REQUIRE(handle, "null handle");
    XR_REQUIRE((mean_ms && max_ms && count) || (!mean_ms && !max_ms && !count), "the three results come together");
    XrRccl* C = (XrRccl*)handle;
    if (mean_ms) {
        const unsigned n = C->n_timed < (unsigned)XrRccl::RING ? C->n_timed : (unsigned)XrRccl::RING;
        float sum = 0.f, mx = 0.f;
        for (unsigned i = 0; i < n; ++i) {
            float ms = 0.f;
            XR_HIP(hipEventElapsedTime(&ms, C->t0[i], C->t1[i]));
            sum += ms; mx = ms > mx ? ms : mx;
        }
        *mean_ms = n ? sum / n : 0.f; *max_ms = mx; *count = (int)C->n_timed;
    }
    if (timing_on && !C->t0[0])
        for (int i = 0; i < XrRccl::RING; ++i) { XR_HIP(hipEventCreate(&C->t0[i])); XR_HIP(hipEventCreate(&C->t1[i])); }
    C->timing = timing_on != 0; C->n_timed = 0;
    return XR_OK;
}
// the exchange hooks of xr_ngp_loop_desc served by this communicator
extern "C" int xr_rccl_exchange(void* handle, xr_grad_exchange* out) {
    XR_REQUIRE(handle && out, "null pointer");
    XrRccl* C = (XrRccl*)handle;
    out->all_reduce = ex_all_reduce; out->reduce_scatter = ex_reduce_scatter; out->all_gather = ex_all_gather; out->finish = ex_finish;
    out->ctx = C; out->world_size = C->world; out->rank = C->rank;
    return XR_OK;
}
