// One call = one training step's device work of HashNerfNetwork (networks/hashnerf.py:32-52 without the optimiser):
// hash-grid encode -> fused MLP -> K3 + 5*Huber + masked MSE + K4 -> MLP backward -> table scatter, enqueued back to back on
// one stream from native code.  Measured on the GPU box (tools/hosttime2.py): issued from Python -- ten ctypes calls, a dozen
// tensor allocations, ~1400 interpreter-level calls per iteration -- the step costs 0.64 ms of HOST time against 0.66 ms of
// kernels, i.e. the trainer was about to be bound by the interpreter rather than by the MI355X.  Everything here is the
// existing entry points of this library called in sequence (same kernels, same results); buffers are caller-owned, nothing
// is allocated or synchronised.
#include "xr_common.h"
#include "xr_aux.h"
#include <cstdlib>
#include <initializer_list>

// Timing events for callers without a HIP binding of their own (bench.py brackets one stage of the native step with them)
extern "C" void* xr_timing_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
extern "C" int xr_timing_event_destroy(void* e) { return e && hipEventDestroy((hipEvent_t)e) != hipSuccess ? XR_EHIP : XR_OK; }
extern "C" int xr_timing_event_elapsed_ms(void* a, void* b, float* ms) {
    XR_REQUIRE(a && b && ms, "null pointer");
    XR_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return XR_OK;
}
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
    const xr_adam_fuse* w_color_adam, const char* timed_entry, void* timing_begin, void* timing_end, void* stream_) {
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
    XR_REQUIRE(!table_adam || xr_internal_hashgrid_bwd_adam_supported(n_rows, n_levels, scale_host, resolution_host, offset_host),
               "the fused table update needs a non-atomic scatter path for every level at this row capacity");
    XR_REQUIRE(n_rows > 0 && n_rays > 0 && ld >= n_rows, "bad sizes");
    XR_REQUIRE(mlp_mode >= 0 && mlp_mode <= 3, "mlp_mode is 0 (fp32 MFMA), 1 (fp16), 2 (fp32 forward on 3-way split bf16 operands) or 3 (on 2-way split fp16 operands)");
    XR_REQUIRE(scatter_level0 >= 0 && scatter_level0 < n_levels, "scatter_level0 outside [0, n_levels)");
    XR_REQUIRE(!timed_entry || (timing_begin && timing_end), "a timed entry point needs its two events");
    hipStream_t stream = (hipStream_t)stream_;
    // The reduction of the MLP backward's per-workgroup partials (first read by the optimiser) rides on the helper stream the
    // table scatter forks anyway for its small dense levels, in front of them: one launch and one dependent-kernel boundary
    // (~5 us each on this part) less on the caller's stream, no event of its own.  -DXR_STEP_REDUCE_AUX=0: on the caller's stream.
    // (Round 2 measured a SEPARATE fork / join for it: 0.562 ms against 0.542 ms per iteration -- the extra cross-queue joins
    // cost more than the kernel.)  The zero-fill of the table gradient is gone: the scatter WRITES it (XR_SCATTER_OVERWRITE).
#ifndef XR_STEP_REDUCE_AUX
#define XR_STEP_REDUCE_AUX 1
#endif
    const bool reduce_aux = XR_STEP_REDUCE_AUX != 0;
    auto begin = [&](const char* name) -> int { if (stage_is(timed_entry, name)) XR_HIP(hipEventRecord((hipEvent_t)timing_begin, stream)); return XR_OK; };
    auto end = [&](const char* name) -> int {
        if (stage_is(timed_entry, name)) XR_HIP(hipEventRecord((hipEvent_t)timing_end, stream));
        return XR_OK;
    };
    int rc;
    // coordinate rows {pos3, dt, dir3}: positions and directions are consumed in place (row stride 7)
    if ((rc = begin("xr_hashgrid_fwd")) != XR_OK) return rc;
    // positions: K1's three planes when the caller has them (coalesced loads: 91 -> 83 us at 2.6e5 samples), else the rows
    if (xyz_planes) rc = xr_hashgrid_fwd(table, xyz_planes, 1, plane_stride, n_rows, n_dev, nullptr, n_levels, scale_host, resolution_host, offset_host, enc_t, ld, stream_);
    else rc = xr_hashgrid_fwd(table, coords, 7, 1, n_rows, n_dev, nullptr, n_levels, scale_host, resolution_host, offset_host, enc_t, ld, stream_);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_hashgrid_fwd")) != XR_OK || (rc = begin("xr_nerf_mlp_fwd")) != XR_OK) return rc;
    rc = xr_nerf_mlp_fwd(mlp_mode, enc_t, ld, coords + 4, 7, n_rows, n_dev, nullptr, w_density, w_color, n_hidden_density, n_hidden_color, pad_value,
                         raw, stream_);
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
        else XR_HIP(hipMemsetAsync(seg, 0, XR_LIVE_ROWS_SEGMENTS(n_rows) * sizeof(uint32_t), stream));
    }
    if ((rc = begin("xr_composite_train")) != XR_OK) return rc;
    // (the two loss scalars are a function of rgb_out: one fixed-order sum on the scatter's helper stream, see below)
    rc = xr_composite_train(raw, coords, rays_numsteps, rays_numsteps_compacted, bg_color, target, alpha_mask, density_grid_mean,
                             n_rays, rgb_activation, density_activation, huber_delta, loss_scale, rgb_out, nullptr, draw, seg, stream_);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_composite_train")) != XR_OK) return rc;
    if ((rc = begin("xr_live_rows")) != XR_OK) return rc;
    if (live_on) {
        rc = xr_live_rows(draw, n_rows, n_dev, seg, rows, n_live, nullptr, ld, 1, stream_);
        if (rc != XR_OK) return rc;
    }
    if ((rc = end("xr_live_rows")) != XR_OK || (rc = begin("xr_nerf_mlp_bwd")) != XR_OK) return rc;
    xr_internal_defer_mlp_reduce(true);                 // (the reduce is issued below: helper stream, or this one)
    rc = xr_nerf_mlp_bwd(mlp_mode, enc_t, ld, coords + 4, 7, n_rows, n_dev, w_density, w_color, n_hidden_density, n_hidden_color, pad_value, draw, denc_t,
                         grad_w_density, grad_w_color, ws_mlp_bwd, ws_mlp_bwd_bytes, rows, n_live, stream_);
    xr_internal_defer_mlp_reduce(false);
    if (rc != XR_OK) return rc;
    if ((rc = end("xr_nerf_mlp_bwd")) != XR_OK) return rc;
    // The small work behind the backward -- the fixed-order sum of its per-workgroup partials into the two gradient buffers (written),
    // the two MLP tensors' optimiser update right behind their gradients, the loss scalars, the clear of the caller's live-row segment
    // counts (read by the ranking pass long ago) -- rides inside the scatter's binning launch as a few extra workgroups (xr_aux.h).
    // Where the scatter has no such launch for this call it runs here, behind it, as launches of its own.
    XrAuxWork aux; memset(&aux, 0, sizeof(aux));
    if ((rc = xr_internal_mlp_bwd_reduce_desc(ws_mlp_bwd, n_rows, n_hidden_density, n_hidden_color, &aux)) != XR_OK) return rc;
    aux.g0 = grad_w_density; aux.g1 = grad_w_color; aux.overwrite = 1;
    const bool aux_adam = w_density_adam && w_color_adam && (uint32_t)w_density_adam->n == aux.split && (uint32_t)w_color_adam->n == aux.gw - aux.split;
    if (aux_adam) {
        const xr_adam_fuse* ad[2] = {w_density_adam, w_color_adam};
        XrAdamArgs* A[2] = {&aux.a0, &aux.a1};
        // the constants exactly as xr_adam_step_multi hands them to its kernel: ONE set (the first tensor's) for both tensors, the EMA
        // copies only where both tensors have one
        const xr_adam_fuse& H = *w_density_adam;
        const float bc1 = 1.f - powf(H.beta1, (float)H.step), bc2 = 1.f - powf(H.beta2, (float)H.step);
        const bool ema = ad[0]->ema && ad[1]->ema;
        for (int k = 0; k < 2; ++k)
            *A[k] = XrAdamArgs{ad[k]->param, ad[k]->m, ad[k]->v, ema ? ad[k]->ema : nullptr, H.beta1, H.beta2, H.lr / bc1, sqrtf(bc2), H.eps,
                               H.weight_decay, H.ema_momentum, H.grad_scale};
        aux.adam = 1;
    }
    aux.rgb = rgb_out; aux.target = target; aux.alpha = alpha_mask; aux.n_rays = n_rays; aux.delta = huber_delta; aux.scale = loss_scale; aux.loss = loss_mse;
    if (live_on && live_seg_count) { aux.clear = live_seg_count; aux.clear_words = XR_LIVE_ROWS_SEGMENTS(n_rows); }
    if (reduce_aux) xr_internal_scatter_aux_work(&aux);
    if ((rc = begin("xr_hashgrid_bwd")) != XR_OK) return rc;
    // data-parallel callers scatter the levels below scatter_level0 themselves (xr_hashgrid_bwd with the same row list, found
    // through xr_nerf_mlp_bwd_list_slots) AFTER handing the finer levels' gradient slice to the collective: table offsets are
    // absolute, so the level metadata is simply passed from that level on
    // the gradient slices of the scattered levels are written, not added to: no zero-fill of the 48.8-MB table gradient
    if (table_adam)
        rc = xr_hashgrid_bwd_adam(coords, 7, denc_t, ld, n_rows, live_on ? n_live : n_dev, rows, n_levels, scale_host, resolution_host, offset_host,
                                  ws_scatter, ws_scatter_bytes, table_adam, stream_);
    else
        rc = xr_hashgrid_bwd(coords, 7, denc_t + (size_t)2 * scatter_level0 * ld, ld, n_rows, live_on ? n_live : n_dev, rows, n_levels - scatter_level0,
                              scale_host + scatter_level0, resolution_host + scatter_level0, offset_host + scatter_level0, grad_table,
                              ws_scatter, ws_scatter_bytes, XR_SCATTER_OVERWRITE, stream_);
    xr_internal_scatter_aux_work(nullptr);
    if (rc != XR_OK) return rc;
    if (!aux.done) {                                  // the scatter had no binning launch for this call (or the build keeps them apart)
        if ((rc = xr_internal_mlp_bwd_reduce(ws_mlp_bwd, n_rows, n_hidden_density, n_hidden_color, grad_w_density, grad_w_color, 1, stream_)) != XR_OK) return rc;
        if (aux.clear) XR_HIP(hipMemsetAsync(aux.clear, 0, (size_t)aux.clear_words * sizeof(uint32_t), stream));
        if (w_density_adam) {
            float* p[2] = {w_density_adam->param, w_color_adam->param}; const float* g[2] = {grad_w_density, grad_w_color};
            float* m[2] = {w_density_adam->m, w_color_adam->m}; float* v[2] = {w_density_adam->v, w_color_adam->v};
            float* e[2] = {w_density_adam->ema, w_color_adam->ema};
            const size_t nn[2] = {(size_t)w_density_adam->n, (size_t)w_color_adam->n};
            rc = xr_adam_step_multi(2, p, g, m, v, (e[0] && e[1]) ? e : nullptr, nn, w_density_adam->step, w_density_adam->lr, w_density_adam->beta1,
                                    w_density_adam->beta2, w_density_adam->eps, w_density_adam->weight_decay, w_density_adam->ema_momentum,
                                    w_density_adam->grad_scale, stream_);
            if (rc != XR_OK) return rc;
        }
        if ((rc = xr_train_loss_scalars(rgb_out, target, alpha_mask, n_rays, huber_delta, loss_scale, loss_mse, stream_)) != XR_OK) return rc;
    } else if (w_density_adam && !aux_adam) {         // (tensor sizes that are not the backward's gradient widths: the update as its own launch)
        float* p[2] = {w_density_adam->param, w_color_adam->param}; const float* g[2] = {grad_w_density, grad_w_color};
        float* m[2] = {w_density_adam->m, w_color_adam->m}; float* v[2] = {w_density_adam->v, w_color_adam->v};
        float* e[2] = {w_density_adam->ema, w_color_adam->ema};
        const size_t nn[2] = {(size_t)w_density_adam->n, (size_t)w_color_adam->n};
        rc = xr_adam_step_multi(2, p, g, m, v, (e[0] && e[1]) ? e : nullptr, nn, w_density_adam->step, w_density_adam->lr, w_density_adam->beta1,
                                w_density_adam->beta2, w_density_adam->eps, w_density_adam->weight_decay, w_density_adam->ema_momentum,
                                w_density_adam->grad_scale, stream_);
        if (rc != XR_OK) return rc;
    }
    if ((rc = end("xr_hashgrid_bwd")) != XR_OK) return rc;
    return XR_OK;
}

// A refresh window's marches as ONE series of launches (contract: include/xrnerf_mi355.h, xr_ngp_window_march).  Between two grid
// refreshes the occupancy bitfield and the batch size are constant (ngp_grid_sampler.py:194-197 refreshes at iter % 16 == 0,
// :268-281 changes the batch size at iter % 16 == 15) and K1 reads no weights (ray_sampler.cu:5-116): every march of the window is known the
// moment the refresh ends.  Round 4 issued them one by one, two iterations ahead of their step, on a side stream -- fifteen
// 12.9 K-ray launches per window (200 waves on 1024 SIMDs, ~290 us of latency each) beside the steps, costing the kernels they ran
// beside 15-40 us per iteration plus two event operations on the step's queue.  K1's count pass takes the same ~170 us from 4 K to
// 65 K rays (profiles/r04_k1_lanes_per_ray_ab.txt), so the window's other marches are one launch of each K1 pass right behind the
// refresh, and the normal iterations have no side stream at all.
extern "C" int xr_ngp_window_march(const xr_ngp_window* W, uint32_t first_chunk, uint32_t n_chunks, uint32_t batches_ready, uint32_t n_rays,
                                   const float* rays_rgb_rows, uint64_t n_table_rays, uint64_t* cur_ray, uint64_t batch_seed,
                                   uint64_t batch_call_index, const uint8_t* bitfield, float aabb0, float aabb1, float near_distance,
                                   float cone_angle, uint32_t max_samples, uint64_t k1_call_index, uint32_t max_compacted, void* workspace,
                                   size_t workspace_bytes, uint32_t* counter_host_pinned, void* stream_) {
    XR_REQUIRE(W && cur_ray && bitfield && W->rays_o && W->rays_d && W->target && W->alpha && W->bg && W->img_ids && W->rays_index &&
               W->rays_numsteps && W->numsteps_clipped && W->coords && W->counter2 && W->n_valid, "null pointer");
    XR_REQUIRE(n_chunks >= 1 && first_chunk + n_chunks <= (uint32_t)XR_NGP_WINDOW && batches_ready <= n_chunks, "chunks outside the window");
    XR_REQUIRE(n_rays >= 1 && n_rays <= W->ray_stride && W->coords_stride >= max_samples, "the window's buffers are too small");
    XR_REQUIRE(batches_ready == n_chunks || (rays_rgb_rows && n_rays <= n_table_rays), "no ray table to draw the batches from");
    const size_t r0 = (size_t)first_chunk * W->ray_stride;
    uint64_t st, inc;
    int rc;
    if (batches_ready < n_chunks) {
        // HashBatchSample's cursor (datasets.DeviceRayTable.next_batch): a batch that would run over the end of the table starts at row 0
        uint64_t row0[XR_NGP_WINDOW];
        uint64_t cur = *cur_ray;
        for (uint32_t c = batches_ready; c < n_chunks; ++c) {
            if (cur + n_rays > n_table_rays) cur = 0;
            row0[c - batches_ready] = cur;
            cur += n_rays;
        }
        *cur_ray = cur;
        const size_t rb = r0 + (size_t)batches_ready * W->ray_stride;
        xr_pcg32_host_state(batch_seed, batch_call_index, &st, &inc);
        rc = xr_make_batch_series(rays_rgb_rows, row0, n_rays, n_chunks - batches_ready, W->ray_stride, st, inc, W->rays_o + 3 * rb, W->rays_d + 3 * rb,
                                  W->target + 3 * rb, W->alpha + rb, W->bg + 3 * rb, W->img_ids + rb, stream_);
        if (rc != XR_OK) return rc;
    }
    xr_pcg32_host_state(9121, k1_call_index, &st, &inc);
    rc = xr_rays_sampler_series(W->rays_o + 3 * r0, W->rays_d + 3 * r0, W->ray_stride, bitfield, n_rays, n_chunks, aabb0, aabb1, near_distance, cone_angle,
                                max_samples, st, inc, W->coords + 7 * (size_t)first_chunk * W->coords_stride, W->coords_stride, W->rays_index + r0,
                                W->rays_numsteps + 2 * r0, W->counter2 + 2 * first_chunk,
                                W->xyz_planes ? W->xyz_planes + 3 * (size_t)first_chunk * W->plane_stride : nullptr, W->plane_stride, workspace,
                                workspace_bytes, stream_);
    if (rc != XR_OK) return rc;
    rc = xr_clip_numsteps(W->rays_numsteps + 2 * r0, W->counter2 + 2 * first_chunk, n_rays, n_chunks, W->ray_stride, max_compacted,
                                 W->numsteps_clipped + 2 * r0, W->n_valid + 2 * first_chunk, stream_);
    if (rc != XR_OK) return rc;
    if (counter_host_pinned)
        XR_HIP(hipMemcpyAsync(counter_host_pinned + 2 * first_chunk, W->counter2 + 2 * first_chunk, (size_t)n_chunks * 2 * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, (hipStream_t)stream_));
    return XR_OK;
}

// ------------------------------------------------------------------------------------------------ the loop between two refreshes
// k iterations of the trainer's steady state as one call (contract: include/xrnerf_mi355.h, xr_ngp_loop_run): iteration `it` steps on
// the marched batch in chunk it % XR_NGP_WINDOW of the window, with the three updates inside.  One in-order stream, no events: the
// marches were enqueued before (xr_ngp_window_march; the caller orders `stream` behind them once).
extern "C" int xr_ngp_loop_run(const xr_ngp_loop_desc* desc, xr_ngp_loop_state* state, uint32_t k, uint32_t n_rays, const float* lr,
                               const float* ema_momentum, const char* timed_entry, void* const* timing_events, void* const* iter_events) {
    XR_REQUIRE(desc && state && lr && ema_momentum, "null pointer");
    const xr_ngp_loop_desc& D = *desc;
    xr_ngp_loop_state& S = *state;
    const xr_ngp_window& W = D.window;
    XR_REQUIRE(k >= 1 && k <= (uint32_t)XR_NGP_WINDOW && n_rays >= 1 && n_rays <= W.ray_stride, "bad sizes");
    XR_REQUIRE(!timed_entry || timing_events, "a timed entry point needs its events");
    const xr_grad_exchange* X = D.exchange;
    XR_REQUIRE(D.adam_w_density.param == D.w_density && D.adam_w_color.param == D.w_color, "the updates name the step's tensors");
    XR_REQUIRE(X ? (D.dp_mode == 1 ? (D.shard_grad && D.table_padded && D.shard_floats > 0 && D.adam_table.n == D.shard_floats &&
                                        D.adam_table.param == D.table_padded + (uint64_t)X->rank * D.shard_floats && X->reduce_scatter && X->all_gather)
                                    : (D.dp_mode == 0 && D.adam_table.param == D.table))
                 : D.adam_table.param == D.table, "the table update names the step's table (zero1: this rank's shard of it)");
    XR_REQUIRE(!X || (X->all_reduce && X->finish && X->world_size >= 1 && D.step[0].grad_table && D.step[1].grad_table &&
                      D.split_level >= 0 && D.split_level < D.n_levels && (D.dp_mode == 0 || D.split_level == 0)), "bad exchange");
    XR_REQUIRE(W.coords && W.rays_numsteps && W.numsteps_clipped && W.n_valid && W.bg && W.target && W.alpha && W.coords_stride >= D.n_rows, "bad window");
    hipStream_t stream = (hipStream_t)D.stream;
    int rc;
    xr_adam_fuse at = D.adam_table, ad = D.adam_w_density, ac = D.adam_w_color;
    for (uint32_t j = 0; j < k; ++j) {
        const uint64_t it = S.iter;
        if (iter_events) XR_HIP(hipEventRecord((hipEvent_t)iter_events[j], stream));
        const size_t c = (size_t)(it % (uint64_t)XR_NGP_WINDOW), r0 = c * W.ray_stride;
        S.step_turn ^= 1u;
        const xr_ngp_step_set& B = D.step[S.step_turn & 1u];
        S.adam_step += 1;
        at.step = ad.step = ac.step = S.adam_step;
        at.lr = ad.lr = ac.lr = lr[j];
        at.ema_momentum = ad.ema_momentum = ac.ema_momentum = ema_momentum[j];
        const float* coords = W.coords + 7 * c * W.coords_stride;
        const size_t table_floats = 2 * (size_t)D.offset_host[D.n_levels];
        rc = xr_ngp_train_step(D.table, D.w_density, D.w_color, D.n_hidden_density, D.n_hidden_color, D.pad_value, D.mlp_mode, D.n_levels, D.scale_host,
                               D.resolution_host, D.offset_host, coords, D.n_rows, W.n_valid + 2 * c, W.rays_numsteps + 2 * r0,
                               W.numsteps_clipped + 2 * r0, n_rays, W.bg + 3 * r0, W.target + 3 * r0, W.alpha + r0, D.density_grid_mean, D.rgb_activation,
                               D.density_activation, D.huber_delta, D.loss_scale, B.enc_t, D.ld, B.raw, B.draw, B.denc_t, B.rgb_out, B.zero_block,
                               B.zero_floats, B.grad_w_density, B.grad_w_color, B.loss_mse, B.live_seg_count, X ? B.grad_table : nullptr,
                               X ? table_floats : 0, 0, D.ws_mlp_bwd, D.ws_mlp_bwd_bytes, D.ws_scatter, D.ws_scatter_bytes, X ? D.split_level : 0,
                               W.xyz_planes ? W.xyz_planes + 3 * c * W.plane_stride : nullptr, W.plane_stride, X ? nullptr : &at, X ? nullptr : &ad,
                               X ? nullptr : &ac, timed_entry, timed_entry ? timing_events[2 * j] : nullptr,
                               timed_entry ? timing_events[2 * j + 1] : nullptr, D.stream);
        if (rc != XR_OK) return rc;
        if (X) {
            // ---- data parallel: the buckets to the exchange as they complete, then one optimiser launch on the summed gradients
            // (the two MLP gradients are adjacent in the set's block: one bucket)
            XR_REQUIRE(B.grad_w_color == B.grad_w_density + D.adam_w_density.n, "the two MLP gradients form one bucket");
            if ((rc = X->all_reduce(X->ctx, B.grad_w_density, (size_t)(D.adam_w_density.n + D.adam_w_color.n), D.stream)) != XR_OK) return rc;
            if (D.dp_mode == 1) {
                if ((rc = X->reduce_scatter(X->ctx, B.grad_table, D.shard_grad, (size_t)D.shard_floats, D.stream)) != XR_OK) return rc;
            } else if (D.split_level > 0) {
                const size_t cut = 2 * (size_t)D.offset_host[D.split_level];
                if ((rc = X->all_reduce(X->ctx, B.grad_table + cut, table_floats - cut, D.stream)) != XR_OK) return rc;
                // the coarser levels underneath that collective, on the step's own row list
                uint32_t *rows = nullptr, *seg = nullptr, *n_live = nullptr;
                static const bool live_on = []() { const char* e = getenv("XR_MLP_LIVE"); return !(e && e[0] == '0'); }();
                if (live_on && (rc = xr_nerf_mlp_bwd_list_slots(D.ws_mlp_bwd, D.ws_mlp_bwd_bytes, D.n_rows, &rows, &seg, &n_live)) != XR_OK) return rc;
                rc = xr_hashgrid_bwd(coords, 7, B.denc_t, D.ld, D.n_rows, live_on ? n_live : W.n_valid + 2 * c, live_on ? rows : nullptr, D.split_level,
                                      D.scale_host, D.resolution_host, D.offset_host, B.grad_table, D.ws_scatter, D.ws_scatter_bytes, XR_SCATTER_OVERWRITE, D.stream);
                if (rc != XR_OK) return rc;
                if ((rc = X->all_reduce(X->ctx, B.grad_table, cut, D.stream)) != XR_OK) return rc;
            } else if ((rc = X->all_reduce(X->ctx, B.grad_table, table_floats, D.stream)) != XR_OK) return rc;
            if ((rc = X->finish(X->ctx, D.stream)) != XR_OK) return rc;
            float* p[3] = {at.param, ad.param, ac.param};
            const float* g[3] = {D.dp_mode == 1 ? D.shard_grad : B.grad_table, B.grad_w_density, B.grad_w_color};
            float* m[3] = {at.m, ad.m, ac.m}; float* v[3] = {at.v, ad.v, ac.v}; float* e[3] = {at.ema, ad.ema, ac.ema};
            const size_t nn[3] = {(size_t)at.n, (size_t)ad.n, (size_t)ac.n};
            rc = xr_adam_step_multi(3, p, g, m, v, (e[0] && e[1] && e[2]) ? e : nullptr, nn, at.step, at.lr, at.beta1, at.beta2, at.eps, at.weight_decay,
                                    at.ema_momentum, 1.0f / (float)X->world_size, D.stream);
            if (rc != XR_OK) return rc;
            if (D.dp_mode == 1) {          // every rank's updated shard -> the full table, in place
                if ((rc = X->all_gather(X->ctx, at.param, D.table_padded, (size_t)D.shard_floats, D.stream)) != XR_OK) return rc;
                if ((rc = X->finish(X->ctx, D.stream)) != XR_OK) return rc;
            }
        }
        S.last_step_set = S.step_turn & 1u;
        S.iter = it + 1;
    }
    if (iter_events) XR_HIP(hipEventRecord((hipEvent_t)iter_events[k], stream));
    return XR_OK;
}
