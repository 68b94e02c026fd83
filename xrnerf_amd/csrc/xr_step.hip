// One call = one training step's device work of HashNerfNetwork (networks/hashnerf.py:32-52 without the optimiser):
// hash-grid encode -> fused MLP -> K3 + 5*Huber + masked MSE + K4 -> MLP backward -> table scatter, enqueued back to back on
// one stream from native code.  Measured on the GPU box (tools/hosttime2.py): issued from Python -- ten ctypes calls, a dozen
// tensor allocations, ~1400 interpreter-level calls per iteration -- the step costs 0.64 ms of HOST time against 0.66 ms of
// kernels, i.e. the trainer was about to be bound by the interpreter rather than by the MI355X.  Everything here is the
// existing entry points of this library called in sequence (same kernels, same results); buffers are caller-owned, nothing
// is allocated or synchronised.
#include "xr_common.h"
#include <cstdlib>
#include <initializer_list>

// Timing events for callers without a HIP binding of their own (bench.py brackets one stage of the native step with them)
extern "C" void* xr_timing_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
// an event for ordering only (hipEventDisableTiming: no timestamp is taken when it completes); same destroy / wait calls
extern "C" void* xr_order_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? (void*)e : nullptr;
}
extern "C" int xr_timing_event_destroy(void* e) { return e && hipEventDestroy((hipEvent_t)e) != hipSuccess ? XR_EHIP : XR_OK; }
extern "C" int xr_timing_event_elapsed_ms(void* a, void* b, float* ms) {
    XR_REQUIRE(a && b && ms, "null pointer");
    XR_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return XR_OK;
}

// order `stream` behind a recorded event (a caller without a HIP binding holds events from xr_timing_event_create)
extern "C" int xr_stream_wait_event(void* stream, void* event) {
    XR_REQUIRE(event, "null event");
    XR_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return XR_OK;
}

// record an event of this library on a stream (callers without a HIP binding: bench.py's per-iteration boundary events)
extern "C" int xr_event_record(void* event, void* stream) {
    XR_REQUIRE(event, "null event");
    XR_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return XR_OK;
}

static bool stage_is(const char* timed, const char* name) { return timed && strcmp(timed, name) == 0; }

extern "C" int xr_ngp_train_step(
    const float* table, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color, float pad_value,
    int mlp_mode, int n_levels, const float* scale_host, const uint32_t* resolution_host, const uint32_t* offset_host,
    const float* coords, uint32_t n_rows, const uint32_t* n_dev, const int32_t* rays_numsteps, const int32_t* rays_numsteps_compacted,
    uint32_t n_rays, const float* bg_color, const float* target, const float* alpha_mask, const float* density_grid_mean,
    int rgb_activation, int density_activation, float huber_delta, float loss_scale,
    float* enc_t, uint32_t ld, float* raw, float* draw, float* denc_t, float* rgb_out,
    float* zero_block, size_t zero_floats, float* grad_w_density, float* grad_w_color, float* loss_mse, uint32_t* live_seg_count,
    float* grad_table, size_t table_floats, int zero_draw,
    void* ws_mlp_bwd, size_t ws_mlp_bwd_bytes, void* ws_scatter, size_t ws_scatter_bytes, int scatter_level0,
    const float* xyz_planes, uint32_t plane_stride, const xr_adam_fuse* table_adam, const xr_adam_fuse* w_density_adam,
    const xr_adam_fuse* w_color_adam, const char* mark_entry, void* mark_event,
    const char* timed_entry, void* timing_begin, void* timing_end, void* stream_) {
    XR_REQUIRE(table && w_density && w_color && coords && rays_numsteps && rays_numsteps_compacted && bg_color && target &&
               alpha_mask && density_grid_mean && enc_t && raw && draw && denc_t && rgb_out && zero_block && grad_w_density &&
               grad_w_color && loss_mse && (grad_table || table_adam), "null pointer");
    // table_adam: the scatter applies the optimiser's update to the table instead of writing its gradient (xr_hashgrid_bwd_adam);
    // checked before anything is enqueued
    XR_REQUIRE(!table_adam || (scatter_level0 == 0 && table_adam->param == table), "the fused table update takes the whole table of this step");
    XR_REQUIRE(!w_density_adam == !w_color_adam, "the two MLP tensors' updates come together");
    XR_REQUIRE(!w_density_adam || (w_density_adam->param == w_density && w_color_adam->param == w_color && w_density_adam->step == w_color_adam->step &&
                                   w_density_adam->n > 0 && w_color_adam->n > 0),
               "mlp_adam: the step's own weight tensors, one step count");
    XR_REQUIRE(!table_adam || xr_hashgrid_bwd_adam_supported(n_rows, n_levels, scale_host, resolution_host, offset_host),
               "the fused table update needs a non-atomic scatter path for every level at this row capacity");
    XR_REQUIRE(n_rows > 0 && n_rays > 0 && ld >= n_rows, "bad sizes");
    XR_REQUIRE(mlp_mode >= 0 && mlp_mode <= 2, "mlp_mode is 0 (fp32 MFMA), 1 (fp16) or 2 (fp32 forward on split bf16 operands)");
    XR_REQUIRE(scatter_level0 >= 0 && scatter_level0 < n_levels, "scatter_level0 outside [0, n_levels)");
    XR_REQUIRE(!timed_entry || (timing_begin && timing_end), "a timed entry point needs its two events");
    XR_REQUIRE(!mark_entry || mark_event, "a marked entry point needs its event");
    hipStream_t stream = (hipStream_t)stream_;
    // The reduction of the MLP backward's per-workgroup partials (first read by the optimiser) rides on the helper stream the
    // table scatter forks anyway for its small dense levels, in front of them: one launch and one dependent-kernel boundary
    // (~5 us each on this part) less on the caller's stream, no event of its own.  -DXR_STEP_REDUCE_AUX=0: on the caller's stream.
    // (Round 2 measured a SEPARATE fork / join for it: 0.562 ms against 0.542 ms per iteration -- the extra cross-queue joins
    // cost more than the kernel.)  The zero-fill of the table gradient is gone: the scatter WRITES it (XR_SCATTER_OVERWRITE).
#ifndef XR_STEP_REDUCE_AUX
#define XR_STEP_REDUCE_AUX 1
#endif
#ifndef XR_STEP_SHARE_FORK
#define XR_STEP_SHARE_FORK 1
#endif
    const bool reduce_aux = XR_STEP_REDUCE_AUX != 0;
    auto begin = [&](const char* name) -> int { if (stage_is(timed_entry, name)) XR_HIP(hipEventRecord((hipEvent_t)timing_begin, stream)); return XR_OK; };
    // mark_entry / mark_event: the event is recorded on `stream` right behind the named entry point's launches (the trainer
    // starts the next batch's side-stream march from there instead of beside the fused-MLP forward)
    auto end = [&](const char* name) -> int {
        if (stage_is(timed_entry, name)) XR_HIP(hipEventRecord((hipEvent_t)timing_end, stream));
        if (mark_event && stage_is(mark_entry, name)) XR_HIP(hipEventRecord((hipEvent_t)mark_event, stream));
        return XR_OK;
    };
    int rc;
    // coordinate rows {pos3, dt, dir3}: positions and directions are consumed in place (row stride 7)
    if ((rc = begin("xr_hashgrid_fwd")) != XR_OK) return rc;
    // positions: K1's three planes when the caller has them (coalesced loads: 91 -> 83 us at 2.6e5 samples), else the rows
    if (xyz_planes) rc = xr_hashgrid_fwd2(table, xyz_planes, 1, plane_stride, n_rows, n_dev, nullptr, n_levels, scale_host, resolution_host, offset_host, enc_t, ld, stream_);
    else rc = xr_hashgrid_fwd(table, coords, 7, n_rows, n_dev, nullptr, n_levels, scale_host, resolution_host, offset_host, enc_t, ld, stream_);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_hashgrid_fwd")) != XR_OK || (rc = begin("xr_nerf_mlp_fwd")) != XR_OK) return rc;
    const bool f16_mlp = mlp_mode == 1;
    auto mlp_fwd = mlp_mode == 1 ? xr_nerf_mlp_fwd_f16 : mlp_mode == 2 ? xr_nerf_mlp_fwd_bf16x3 : xr_nerf_mlp_fwd;
    rc = mlp_fwd(enc_t, ld, coords + 4, 7, n_rows, n_dev, nullptr, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, raw,
                 stream_);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_nerf_mlp_fwd")) != XR_OK) return rc;
    // nothing of zero_block is zero-filled any more: the MLP gradients are WRITTEN by the reduction of their partials, the loss
    // scalars by xr_train_loss_scalars, and the live-row counts are handed over zeroed and zeroed again after use (below)
    (void)zero_block; (void)zero_floats;
    if (zero_draw) XR_HIP(hipMemsetAsync(draw, 0, (size_t)n_rows * 4 * sizeof(float), stream));
    // rows with an exactly-zero dL/d(raw) (T == 0 behind a surface) are skipped by the MLP backward AND the scatter: one list.
    // The compositor counts the live rows per 1024-row segment while it writes them (no separate counting launch).
    // XR_MLP_LIVE=0 (measurement, read once): no list -- both run over every marched row, same results.
    static const bool live_on = []() { const char* e = getenv("XR_MLP_LIVE"); return !(e && e[0] == '0'); }();
    uint32_t *rows = nullptr, *seg = nullptr, *n_live = nullptr;
    if (live_on) {
        rc = xr_nerf_mlp_bwd_list_slots(ws_mlp_bwd, ws_mlp_bwd_bytes, n_rows, &rows, &seg, &n_live);
        if (rc != XR_OK) return rc;
        // the per-segment counts start at zero.  A caller's own array (live_seg_count) is zero on entry by contract and is cleared
        // again on the scatter's helper stream once the ranking pass has read it: no fill on this stream.  Without one: the
        // slot in the backward's workspace, cleared here.
        if (live_seg_count) seg = live_seg_count;
        else XR_HIP(hipMemsetAsync(seg, 0, xr_live_rows_segments(n_rows) * sizeof(uint32_t), stream));
    }
    if ((rc = begin("xr_composite_train")) != XR_OK) return rc;
    // (the two loss scalars are a function of rgb_out: one fixed-order sum on the scatter's helper stream, see below)
    rc = xr_composite_train2(raw, coords, rays_numsteps, rays_numsteps_compacted, bg_color, target, alpha_mask, density_grid_mean,
                             n_rays, rgb_activation, density_activation, huber_delta, loss_scale, rgb_out, nullptr, draw, seg, stream_);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_composite_train")) != XR_OK) return rc;
    if ((rc = begin("xr_live_rows")) != XR_OK) return rc;
    if (live_on) {
        rc = xr_live_rows2(draw, n_rows, n_dev, seg, rows, n_live, nullptr, ld, 1, stream_);
        if (rc != XR_OK) return rc;
    }
    if ((rc = end("xr_live_rows")) != XR_OK || (rc = begin("xr_nerf_mlp_bwd")) != XR_OK) return rc;
    xr_internal_defer_mlp_reduce(true);                 // (the reduce is issued below: helper stream, or this one)
    rc = f16_mlp ? xr_nerf_mlp_bwd_f16(enc_t, ld, coords + 4, 7, n_rows, n_dev, w_density, w_color, n_hidden_density, n_hidden_color,
                                       pad_value, draw, denc_t, grad_w_density, grad_w_color, ws_mlp_bwd, ws_mlp_bwd_bytes, rows, n_live, stream_)
                 : xr_nerf_mlp_bwd(enc_t, ld, coords + 4, 7, n_rows, n_dev, w_density, w_color, n_hidden_density, n_hidden_color,
                                   pad_value, draw, denc_t, grad_w_density, grad_w_color, ws_mlp_bwd, ws_mlp_bwd_bytes, rows, n_live, stream_);
    xr_internal_defer_mlp_reduce(false);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_nerf_mlp_bwd")) != XR_OK) return rc;
    struct TailArgs { void* ws; uint32_t n; float *gd, *gc; const float *rgb, *target, *alpha; uint32_t n_rays; float delta, scale; float* loss;
                      const xr_adam_fuse *ad, *ac; uint32_t* seg_clear; size_t seg_bytes; }
        ta = {ws_mlp_bwd, n_rows, grad_w_density, grad_w_color, rgb_out, target, alpha_mask, n_rays, huber_delta, loss_scale, loss_mse,
              w_density_adam, w_color_adam, (live_on && live_seg_count) ? live_seg_count : nullptr, xr_live_rows_segments(n_rows) * sizeof(uint32_t)};
    XrAuxPrologue pro = {[](hipStream_t st, void* a) -> int {
                             auto* r = (TailArgs*)a;
                             int rc1 = xr_internal_mlp_bwd_reduce(r->ws, r->n, r->gd, r->gc, 1, st);      // writes the two gradient buffers
                             if (rc1 != XR_OK) return rc1;
                             if (r->seg_clear) XR_HIP(hipMemsetAsync(r->seg_clear, 0, r->seg_bytes, st));      // read by the ranking pass long ago
                             if (r->ad) {                      // the MLP tensors' optimiser update, right behind their gradients
                                 float* p[2] = {r->ad->param, r->ac->param}; const float* g[2] = {r->gd, r->gc};
                                 float* m[2] = {r->ad->m, r->ac->m}; float* v[2] = {r->ad->v, r->ac->v}; float* e[2] = {r->ad->ema, r->ac->ema};
                                 const size_t nn[2] = {(size_t)r->ad->n, (size_t)r->ac->n};
                                 rc1 = xr_adam_step_multi(2, p, g, m, v, (e[0] && e[1]) ? e : nullptr, nn, r->ad->step, r->ad->lr, r->ad->beta1, r->ad->beta2,
                                                          r->ad->eps, r->ad->weight_decay, r->ad->ema_momentum, r->ad->grad_scale, st);
                                 if (rc1 != XR_OK) return rc1;
                             }
                             return xr_train_loss_scalars(r->rgb, r->target, r->alpha, r->n_rays, r->delta, r->scale, r->loss, st);
                         }, &ta, false};
    if (reduce_aux) xr_internal_scatter_aux_prologue(&pro);
    // the mark behind the MLP backward is the last thing on `stream`: the scatter orders its helper stream behind that event instead
    // of recording one of its own (-DXR_STEP_SHARE_FORK=0: its own)
    const bool share_fork = XR_STEP_SHARE_FORK != 0;
    if (share_fork && mark_event && stage_is(mark_entry, "xr_nerf_mlp_bwd") && !stage_is(timed_entry, "xr_hashgrid_bwd"))
        xr_internal_scatter_fork_event(mark_event);
    if ((rc = begin("xr_hashgrid_bwd")) != XR_OK) return rc;
    // data-parallel callers scatter the levels below scatter_level0 themselves (xr_hashgrid_bwd with the same row list, found
    // through xr_nerf_mlp_bwd_list_slots) AFTER handing the finer levels' gradient slice to the collective: table offsets are
    // absolute, so the level metadata is simply passed from that level on
    // the gradient slices of the scattered levels are written, not added to: no zero-fill of the 48.8-MB table gradient
    if (table_adam)
        rc = xr_hashgrid_bwd_adam(coords, 7, denc_t, ld, n_rows, live_on ? n_live : n_dev, rows, n_levels, scale_host, resolution_host, offset_host,
                                  ws_scatter, ws_scatter_bytes, table_adam, stream_);
    else
        rc = xr_hashgrid_bwd2(coords, 7, denc_t + (size_t)2 * scatter_level0 * ld, ld, n_rows, live_on ? n_live : n_dev, rows, n_levels - scatter_level0,
                              scale_host + scatter_level0, resolution_host + scatter_level0, offset_host + scatter_level0, grad_table,
                              ws_scatter, ws_scatter_bytes, XR_SCATTER_OVERWRITE, stream_);
    xr_internal_scatter_aux_prologue(nullptr);
    xr_internal_scatter_fork_event(nullptr);
    if (rc != XR_OK) return rc;
    if (!pro.done && (rc = pro.fn(stream, pro.arg)) != XR_OK) return rc;     // the scatter did not fork (or the build keeps them on this stream)
    if ((rc = end("xr_hashgrid_bwd")) != XR_OK) return rc;
    return XR_OK;
}

// The next batch's side-stream work as one call: HashBatchSample + RandomBGColor (xr_make_batch) -> K1 (xr_rays_sampler) ->
// K2's clipped counts (xr_clip_numsteps) -> asynchronous copy of K1's (rays, samples) counter to pinned host memory.  The two
// hidden generators of the reference (`static pcg32 rng{9121}` per translation unit) and the batch generator are advanced
// here from their call indices.  From Python this sequence cost ~200 us of interpreter time per iteration.
extern "C" int xr_ngp_prefetch(const float* rays_rgb_rows, uint32_t n_rays, uint64_t batch_seed, uint64_t batch_call_index,
                               float* rays_o, float* rays_d, float* target, float* alpha, float* bg, int32_t* img_ids,
                               const uint8_t* bitfield, float aabb0, float aabb1, float near_distance, float cone_angle,
                               uint32_t max_samples, uint64_t k1_call_index, float* coords_out, int32_t* rays_index,
                               int32_t* rays_numsteps, uint32_t* counter2, void* workspace, size_t workspace_bytes,
                               uint32_t max_compacted, int32_t* numsteps_clipped, uint32_t* n_valid_dev,
                               uint32_t* counter_host_pinned, float* xyz_planes, uint32_t plane_stride, void* stream_) {
    uint64_t st, inc;
    xr_pcg32_host_state(batch_seed, batch_call_index, &st, &inc);
    int rc = xr_make_batch(rays_rgb_rows, n_rays, st, inc, rays_o, rays_d, target, alpha, bg, img_ids, stream_);
    if (rc != XR_OK) return rc;
    xr_pcg32_host_state(9121, k1_call_index, &st, &inc);
    rc = xr_rays_sampler2(rays_o, rays_d, bitfield, n_rays, aabb0, aabb1, near_distance, cone_angle, max_samples, st, inc, coords_out,
                          rays_index, rays_numsteps, counter2, xyz_planes, plane_stride, 0, workspace, workspace_bytes, stream_);
    if (rc != XR_OK) return rc;
    rc = xr_clip_numsteps(rays_numsteps, counter2, n_rays, max_compacted, numsteps_clipped, n_valid_dev, max_compacted, 1, stream_);
    if (rc != XR_OK) return rc;
    if (counter_host_pinned)
        XR_HIP(hipMemcpyAsync(counter_host_pinned, counter2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    return XR_OK;
}

// ------------------------------------------------------------------------------------------------ the loop between two refreshes
// k iterations of the trainer's steady state as one call (contract: include/xrnerf_mi355.h, xr_ngp_loop_run).  What one iteration
// enqueues, and where -- the same as xrnerf_amd.train.Trainer.step drives through Python (the trajectory test compares the two):
//   main stream   wait(march of `it` done) -> xr_ngp_train_step(updates inside; mark behind the MLP backward) -> record done(it)
//   side stream   [wait(bitfield), wait(done(it - 1)), wait(mark(it))] -> xr_ngp_prefetch for it + 2 -> record march-done
//                 -> copy of K1's counter to the pinned ring
// A march is never issued across a refresh (iterations = 0 mod f re-write the bitfield it reads and change the batch size): the
// first iteration after a refresh is marched as soon as it is known ("at once"), the second behind the mark of the iteration the
// caller just ran, all others two iterations ahead.
struct XrLoop {
    hipEvent_t march_done[XR_NGP_MARCH_SETS] = {};
    hipEvent_t iter_done[XR_NGP_MARCH_SETS] = {};
    hipEvent_t done_prev[2] = {nullptr, nullptr};          // end of iterations iter - 2, iter - 1 (own or the caller's)
    hipEvent_t join_covers = nullptr;                      // the march event the last step's helper-stream join waited for (inside a window)
};

extern "C" void* xr_ngp_loop_create(void) {
    XrLoop* L = new XrLoop();
    for (int i = 0; i < XR_NGP_MARCH_SETS; ++i)
        if (hipEventCreateWithFlags(&L->march_done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L->iter_done[i], hipEventDisableTiming) != hipSuccess) { delete L; return nullptr; }
    return L;
}
extern "C" int xr_ngp_loop_destroy(void* loop) {
    if (!loop) return XR_OK;
    XrLoop* L = (XrLoop*)loop;
    for (int i = 0; i < XR_NGP_MARCH_SETS; ++i) { if (L->march_done[i]) (void)hipEventDestroy(L->march_done[i]); if (L->iter_done[i]) (void)hipEventDestroy(L->iter_done[i]); }
    delete L;
    return XR_OK;
}

// Hand-over between this loop and a caller that issues marches itself (Trainer's per-iteration path): the event the main stream waits
// on for a march in `set` -- (a) handed out so that the caller can wait for a march this loop issued, (b) recorded on the side
// stream NOW for a march the caller issued there earlier (the stream is in order: the record sits behind that march).
extern "C" void* xr_ngp_loop_march_event(void* loop, uint32_t set) { return (loop && set < (uint32_t)XR_NGP_MARCH_SETS) ? (void*)((XrLoop*)loop)->march_done[set] : nullptr; }
extern "C" int xr_ngp_loop_adopt_march(void* loop, uint32_t set, void* side_stream) {
    XR_REQUIRE(loop && set < (uint32_t)XR_NGP_MARCH_SETS, "bad set");
    XR_HIP(hipEventRecord(((XrLoop*)loop)->march_done[set], (hipStream_t)side_stream));
    return XR_OK;
}

#ifndef XR_LOOP_BATCH_FIRST
#define XR_LOOP_BATCH_FIRST 0     // 1: the batch assembly ahead of the march's start point (measured equal: the march then covers the whole MLP backward, 61 -> 77 us, and less of the scatter)
#endif
// the march of iteration `target` on the side stream: mirror of NGPGridSampler.prefetch_native + Trainer._issue
static int xr_loop_issue_march(XrLoop* L, const xr_ngp_loop_desc& D, xr_ngp_loop_state& S, uint32_t n_rays, hipEvent_t buffer_free, hipEvent_t start) {
    hipStream_t side = (hipStream_t)D.side_stream;
    if (D.bitfield_event) XR_HIP(hipStreamWaitEvent(side, (hipEvent_t)D.bitfield_event, 0));
    if (buffer_free) XR_HIP(hipStreamWaitEvent(side, buffer_free, 0));
    const uint32_t set = (++S.march_launches) % (uint32_t)XR_NGP_MARCH_SETS;
    const xr_ngp_march_set& M = D.march[set];
    if (S.cur_ray + n_rays > D.n_table_rays) S.cur_ray = 0;
    // xr_ngp_prefetch's three calls (XR_LOOP_BATCH_FIRST=1 puts the batch assembly ahead of the start point -- it needs the set's buffers
    // only; measured equal, profiles/r04_march_chain_ab.txt)
    uint64_t st, inc;
    xr_pcg32_host_state(D.batch_seed, S.batches_drawn, &st, &inc);
    if (!XR_LOOP_BATCH_FIRST && start) XR_HIP(hipStreamWaitEvent(side, start, 0));
    int rc = xr_make_batch(D.rays_rgb_rows + (size_t)S.cur_ray * 11, n_rays, st, inc, M.rays_o, M.rays_d, M.target, M.alpha, M.bg, M.img_ids, side);
    if (rc != XR_OK) return rc;
    if (XR_LOOP_BATCH_FIRST && start) XR_HIP(hipStreamWaitEvent(side, start, 0));
    xr_pcg32_host_state(9121, S.k1_calls, &st, &inc);
    rc = xr_rays_sampler2(M.rays_o, M.rays_d, D.bitfield, n_rays, D.aabb0, D.aabb1, D.near_distance, D.cone_angle, D.max_samples, st, inc, M.coords,
                          M.rays_index, M.rays_numsteps, M.counter2, M.xyz_planes, M.plane_stride, 0, D.ws_k1, D.ws_k1_bytes, side);
    if (rc != XR_OK) return rc;
    rc = xr_clip_numsteps(M.rays_numsteps, M.counter2, n_rays, D.max_compacted, M.numsteps_clipped, M.n_valid, D.max_compacted, 1, side);
    if (rc != XR_OK) return rc;
    S.cur_ray += n_rays; S.batches_drawn += 1; S.k1_calls += 1;
    // the main stream waits for the MARCH only: the event sits in front of the counter's device-to-host copy
    XR_HIP(hipEventRecord(L->march_done[set], side));
    if (D.counter_host_pinned) {
        XR_HIP(hipMemcpyAsync(D.counter_host_pinned + 2u * (S.pinned_next % D.n_pinned), M.counter2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, side));
        S.pinned_next += 1;
    }
    S.queue_set[S.queued] = set;
    S.queued += 1;
    return XR_OK;
}

#ifndef XR_LOOP_MERGE_WAITS
#define XR_LOOP_MERGE_WAITS 1        // the wait for the next march rides on the helper stream's join of the step in front (0: a wait of its own)
#endif
#ifndef XR_LOOP_DONE_EVERY
#define XR_LOOP_DONE_EVERY 0         // 1: record the end-of-iteration event in every iteration (the A/B of tools/build_variant.sh)
#endif
extern "C" int xr_ngp_loop_run(void* loop, const xr_ngp_loop_desc* desc, xr_ngp_loop_state* state, uint32_t k, uint32_t n_rays,
                               uint32_t update_grid_freq, const float* lr, const float* ema_momentum, void* ext_done_prev2, void* ext_done_prev1,
                               const char* timed_entry, void* const* timing_events, void* const* iter_events) {
    XR_REQUIRE(loop && desc && state && lr && ema_momentum, "null pointer");
    XrLoop* L = (XrLoop*)loop;
    const xr_ngp_loop_desc& D = *desc;
    xr_ngp_loop_state& S = *state;
    const uint32_t f = update_grid_freq;
    XR_REQUIRE(f >= 2 && k >= 1 && n_rays >= 1 && n_rays <= D.n_table_rays, "bad sizes");
    XR_REQUIRE(S.iter % f != 0 && (S.iter % f) + (uint64_t)k <= (uint64_t)f, "the window crosses a grid refresh (an iteration = 0 mod update_grid_freq is the caller's)");
    XR_REQUIRE(S.queued <= 2 && D.mark_event && D.stream != D.side_stream && D.n_pinned >= 1, "bad loop state");
    XR_REQUIRE(!timed_entry || timing_events, "a timed entry point needs its events");
    XR_REQUIRE(D.adam_table.param == D.table && D.adam_w_density.param == D.w_density && D.adam_w_color.param == D.w_color, "the updates name the step's tensors");
    hipStream_t stream = (hipStream_t)D.stream;
    if (ext_done_prev1) { L->done_prev[0] = ext_done_prev2 ? (hipEvent_t)ext_done_prev2 : L->done_prev[1]; L->done_prev[1] = (hipEvent_t)ext_done_prev1; }
    else if (ext_done_prev2) L->done_prev[0] = (hipEvent_t)ext_done_prev2;
    int rc;
    L->join_covers = nullptr;
    // marches the caller's iteration left to this loop (it runs refresh iterations without issuing any: see Trainer): iteration `iter`
    // at once (ordered behind the end of iter - 2, as if issued during iter - 1), iter + 1 behind the mark of iter - 1
    if (S.queued == 0) {
        if ((rc = xr_loop_issue_march(L, D, S, n_rays, L->done_prev[0], nullptr)) != XR_OK) return rc;
        if ((S.iter + 1) % f != 0 && k >= 1)
            if ((rc = xr_loop_issue_march(L, D, S, n_rays, L->done_prev[0], D.mark_entry ? (hipEvent_t)D.mark_event : nullptr)) != XR_OK) return rc;
    }
    xr_adam_fuse at = D.adam_table, ad = D.adam_w_density, ac = D.adam_w_color;
    for (uint32_t j = 0; j < k; ++j) {
        const uint64_t it = S.iter;
        if (iter_events) XR_HIP(hipEventRecord((hipEvent_t)iter_events[j], stream));
        XR_REQUIRE(S.queued >= 1, "no march queued for this iteration");
        const uint32_t mset = S.queue_set[0];
        S.queue_set[0] = S.queue_set[1]; S.queued -= 1;
        const xr_ngp_march_set& M = D.march[mset];
        // (the previous step's scatter may have put this wait on its helper stream, in front of the join this stream waits for anyway)
        if (L->join_covers != L->march_done[mset]) XR_HIP(hipStreamWaitEvent(stream, L->march_done[mset], 0));
        L->join_covers = nullptr;
        hipEvent_t next_march = (XR_LOOP_MERGE_WAITS && S.queued >= 1) ? L->march_done[S.queue_set[0]] : nullptr;
        S.step_turn ^= 1u;
        const xr_ngp_step_set& B = D.step[S.step_turn & 1u];
        S.adam_step += 1;
        at.step = ad.step = ac.step = S.adam_step;
        at.lr = ad.lr = ac.lr = lr[j];
        at.ema_momentum = ad.ema_momentum = ac.ema_momentum = ema_momentum[j];
        rc = xr_ngp_train_step(D.table, D.w_density, D.w_color, D.n_hidden_density, D.n_hidden_color, D.pad_value, D.mlp_mode, D.n_levels, D.scale_host,
                               D.resolution_host, D.offset_host, M.coords, D.n_rows, M.n_valid, M.rays_numsteps, M.numsteps_clipped, n_rays, M.bg, M.target,
                               M.alpha, D.density_grid_mean, D.rgb_activation, D.density_activation, D.huber_delta, D.loss_scale, B.enc_t, D.ld, B.raw,
                               B.draw, B.denc_t, B.rgb_out, B.zero_block, B.zero_floats, B.grad_w_density, B.grad_w_color, B.loss_mse, B.live_seg_count,
                               nullptr, 0, 0, D.ws_mlp_bwd, D.ws_mlp_bwd_bytes, D.ws_scatter, D.ws_scatter_bytes, 0, M.xyz_planes, M.plane_stride, &at,
                               &ad, &ac, D.mark_entry, D.mark_entry ? D.mark_event : nullptr, timed_entry, timed_entry ? timing_events[2 * j] : nullptr,
                               timed_entry ? timing_events[2 * j + 1] : nullptr, D.stream);
        if (next_march && xr_internal_scatter_join_also_taken()) L->join_covers = next_march;
        xr_internal_scatter_join_also(nullptr);
        if (rc != XR_OK) return rc;
        // Trainer._on_sampled, depth 2: iteration it + 1 at once if nothing is queued for it, then it + 2 behind this step's mark
        if (S.queued == 0 && (it + 1) % f != 0)
            if ((rc = xr_loop_issue_march(L, D, S, n_rays, L->done_prev[1], nullptr)) != XR_OK) return rc;
        if (S.queued == 1 && (it + 1) % f != 0 && (it + 2) % f != 0)
            if ((rc = xr_loop_issue_march(L, D, S, n_rays, L->done_prev[1], D.mark_entry ? (hipEvent_t)D.mark_event : nullptr)) != XR_OK) return rc;
        // "iteration `it` is over" orders the marches that are issued AT ONCE (the first ones of a window, the caller's after a refresh) behind
        // the steps that last read their buffer sets.  Inside a window every march starts behind the mark of a LATER step on this in-order
        // stream, which implies it: only the last two iterations of the window record the event (an event record holds the queue for ~3 us,
        // profiles/r04_event_cost_probe.txt).  Without a mark (march_after = start) every iteration records it.
        if (XR_LOOP_DONE_EVERY || !D.mark_entry || j + 2 >= k) {
            hipEvent_t done = L->iter_done[it % (uint64_t)XR_NGP_MARCH_SETS];
            XR_HIP(hipEventRecord(done, stream));
            L->done_prev[0] = L->done_prev[1]; L->done_prev[1] = done;
        }
        S.last_march_set = mset; S.last_step_set = S.step_turn & 1u;
        S.iter = it + 1;
    }
    if (iter_events) XR_HIP(hipEventRecord((hipEvent_t)iter_events[k], stream));
    return XR_OK;
}
