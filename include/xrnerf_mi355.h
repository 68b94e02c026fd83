/*
 * libxrnerf_mi355.so -- C-ABI of the MI355X-native (gfx950) Instant-NGP hot path that drops in
 * behind openxrlab/xrnerf's `raymarch_cuda` extension module and its `tinycudann` dependency.
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain pointers and sizes, no torch types.  Every pointer is a DEVICE pointer
 *     unless its name ends in `_host`.  Tensors are contiguous fp32 / int32 / uint8 in exactly the
 *     layouts the reference's pybind entry points take (file:line cited per function; paths are
 *     relative to /root/reference/).
 *   - returns 0 on success, a negative XR_E* code on failure (message: xr_last_error()); never
 *     throws, never allocates, never synchronises: work is enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream).  Scratch memory is caller supplied
 *     (`workspace`, size from the matching *_workspace_bytes()).
 *   - the reference keeps hidden global RNG state (`static pcg32 rng{9121}`,
 *     extensions/ngp_raymarch/include/raymarch_shared.h:38, advanced by 2^32 after each launch);
 *     here the generator state is an explicit argument (rng_state, rng_inc), obtained from
 *     xr_pcg32_host_state(seed, n_previous_calls).
 *   - sample order: the reference reserves output ranges with atomicAdd, so its sample order is
 *     schedule dependent; this library assigns bases as the exclusive prefix sum in RAY ORDER
 *     (one valid schedule of the reference, and the one its serial CPU build produces), which
 *     makes every output bit-reproducible.
 */
#ifndef XRNERF_MI355_H
#define XRNERF_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XR_OK 0
#define XR_EINVAL (-22)   /* bad argument (null pointer, size, alignment)      */
#define XR_ENOMEM (-12)   /* workspace too small                                */
#define XR_EHIP (-5)      /* a HIP runtime call / kernel launch failed         */

#define XR_NERF_STEPS 1024u      /* raymarch_shared.h:42 */
#define XR_NERF_CASCADES 8u      /* :43 */
#define XR_NERF_GRIDSIZE 128u    /* :48 */
#define XR_GRID_CELLS (128u * 128u * 128u)

/* ENerfActivation, raymarch_shared.h:619-625 */
enum { XR_ACT_NONE = 0, XR_ACT_RELU = 1, XR_ACT_LOGISTIC = 2, XR_ACT_EXPONENTIAL = 3 };

const char* xr_last_error(void);
int xr_version(void);

/* host helper: state of `pcg32 rng{seed}` after `ncalls` launches (each launch ends with
 * rng.advance() = 2^32; ray_sampler.cu:198, generate_grid_samples_nerf_nonuniform.cu:84) */
void xr_pcg32_host_state(uint64_t seed, uint64_t ncalls, uint64_t* state_host, uint64_t* inc_host);

/* ------------------------------------------------------------------------------------------
 * K1  rays_sampler_api   (extensions/ngp_raymarch/src/ray_sampler.cu:119-200, kernel :5-116)
 * in : rays_o, rays_d [n_rays,3] f32; bitfield [8*128^3/8] u8
 *      (metadata / img_ids / xforms of the reference are loaded-but-dead there, :34-38)
 * out: coords_out [max_samples,7] f32 rows {pos3, warped dt, warped dir3}; rays_index [n_rays]
 *      i32; rays_numsteps [n_rays,2] i32 = (n, base); counter2 [2] u32 = (rays, samples).
 * counter2 is zeroed by this call.  workspace: xr_rays_sampler_workspace_bytes(n_rays, 1). */
size_t xr_rays_sampler_workspace_bytes(uint32_t n_rays, uint32_t n_series /* 1: a single launch */);
/* xyz_planes (nullable): the three position columns of coords_out once more as planes of plane_stride (>= max_samples) floats each --
 * what xr_hashgrid_fwd reads with coalesced loads.
 * rng_chunk (0 = off): ray i draws its jitter as ray i % rng_chunk of launch number i / rng_chunk of a series of launches over
 * rng_chunk rays each, starting at (rng_state, rng_inc) -- a frame that the reference marches in `chunk`-sized launches
 * (networks/nerf.py:50-69; the hidden generator advances by 2^32 per launch, ray_sampler.cu:198) in ONE launch, same samples.
 * rng_ray0 (used with rng_chunk > 0): the launch's ray i is ray rng_ray0 + i of the frame that series of launches covers -- a
 * rank that renders a band of image rows draws exactly the jitter the whole-frame loop would give those rays (image-space sharding
 * of validation frames: pixels independent of the world size).
 * (xr_version 120 retired the two earlier generations of this entry point: the form without planes and the `2` form; this is the former xr_rays_sampler3.) */
int xr_rays_sampler(const float* rays_o, const float* rays_d, const uint8_t* bitfield, uint32_t n_rays,
                    float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples,
                    uint64_t rng_state, uint64_t rng_inc, float* coords_out, int32_t* rays_index,
                    int32_t* rays_numsteps, uint32_t* counter2, float* xyz_planes, uint32_t plane_stride,
                    uint32_t rng_chunk, uint32_t rng_ray0, uint32_t flags, void* workspace, size_t workspace_bytes, void* stream);
/* flags: XR_K1_WIDE = nothing else is running on the device (the march in place at a grid refresh): launches of up to 32 768 rays
 * use 8 lanes per ray (same samples bit for bit, a shorter critical path, 8x the waves) */
#define XR_K1_WIDE 1u
/* n_series launches of K1 over n_rays rays each as ONE launch per pass (blockIdx.y = launch): launch c reads rays_o / rays_d at row
 * c * ray_stride, draws the jitter of the hidden generator's call (first + c) -- (rng_state, rng_inc) = the state of call `first`; the
 * generator moves on by 2^32 per launch, ray_sampler.cu:198 -- and writes coords_out + 7 * c * coords_stride, rays_index / rays_numsteps at
 * row c * ray_stride, counter2 + 2 * c and the three planes at xyz_planes + 3 * c * plane_stride: bit for bit what n_series calls of
 * xr_rays_sampler write.  Launches of more than 65 536 rays are enqueued one after the other.
 * workspace: xr_rays_sampler_workspace_bytes(n_rays, n_series). */
int xr_rays_sampler_series(const float* rays_o, const float* rays_d, uint32_t ray_stride, const uint8_t* bitfield, uint32_t n_rays,
                           uint32_t n_series, float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples,
                           uint64_t rng_state, uint64_t rng_inc, float* coords_out, size_t coords_stride, int32_t* rays_index,
                           int32_t* rays_numsteps, uint32_t* counter2, float* xyz_planes, uint32_t plane_stride, void* workspace,
                           size_t workspace_bytes, void* stream);

/* K2  compacted_coord_api (src/compacted_coord.cu:79-143, kernel :6-77).  The reference's
 * transmittance loop cannot influence any output (its `break` is commented out, :41-44), so
 * network_output is not an argument.  counters are zeroed by this call.
 * out: coords_out [max_compacted,7]; numsteps_out [n_rays,2] = (n_clipped, base_c);
 *      rays_counter [1], numstep_counter [1] u32 (numstep_counter = UNCLIPPED total).
 * workspace: xr_rays_sampler_workspace_bytes(n_rays, 1). */
int xr_compacted_coord(const float* coords_in, const int32_t* numsteps_in, uint32_t n_rays,
                       uint32_t max_compacted, float* coords_out, int32_t* numsteps_out,
                       uint32_t* rays_counter, uint32_t* numstep_counter, void* workspace,
                       size_t workspace_bytes, void* stream);

/* K2 when K1's ray-ordered output is kept in place (no overflow): the compacted coordinates ARE the
 * first min(total, max_compacted) rows of K1's buffer, so only the clipped per-ray counts are
 * needed: numsteps_out[i] = (min(max_compacted - min(max_compacted, base), n), base)
 * (compacted_coord.cu:63-66), and the device-side count of valid rows n_valid_dev[0] = n_valid_dev[1] = min(counter2[1], max_compacted).
 * For the n_series launches of xr_rays_sampler_series in one launch: numsteps arrays with ray_stride rows per launch, counter2 and
 * n_valid_dev [n_series][2]; a single launch: n_series = 1, ray_stride = n_rays. */
int xr_clip_numsteps(const int32_t* numsteps_in, const uint32_t* counter2, uint32_t n_rays, uint32_t n_series, uint32_t ray_stride,
                     uint32_t max_compacted, int32_t* numsteps_out, uint32_t* n_valid_dev, void* stream);

/* K3  calc_rgb_forward_api (src/calc_rgb.cu:208-264, kernel :6-67) */
int xr_calc_rgb_forward(const float* network_output /*[S,4]*/, const float* coords /*[S,7]*/,
                        const int32_t* rays_numsteps, const int32_t* rays_numsteps_compacted,
                        const float* bg_color /*[n_rays,3]*/, uint32_t n_rays, int rgb_activation,
                        int density_activation, float* rgb_output /*[n_rays,3]*/, void* stream);
/* K4  calc_rgb_backward_api (src/calc_rgb.cu:267-327, kernel :71-140); rows of dloss_doutput
 * not covered by a ray are left untouched (caller zero-fills, as renders/hashnerf_render.py:121) */
int xr_calc_rgb_backward(const float* network_output, const int32_t* rays_numsteps_compacted,
                         const float* coords, const float* grad_rgb /*[n_rays,3]*/,
                         const float* rgb_output /*[n_rays,3] saved forward result*/,
                         const float* density_grid_mean /*device, [0] read*/, uint32_t n_rays,
                         int rgb_activation, int density_activation, float* dloss_doutput /*[S,4]*/,
                         void* stream);

/* K3 + scale * HuberLoss(delta, sum) with its gradient + K4 in ONE launch (the training step's compositor sequence,
 * networks/hashnerf.py:32-44 around renders/hashnerf_render.py:60-135): rgb_output [n_rays,3] and dloss_doutput [S,4] are what
 * xr_calc_rgb_forward / xr_calc_rgb_backward produce.  Rows of dloss_doutput behind the last sample are not written.
 * live_seg_count (nullable; XR_LIVE_ROWS_SEGMENTS(S) words the caller zero-fills) gets, per segment of 1024 rows of dloss_doutput,
 * the number of rows that are not exactly zero ADDED -- the counting pass of xr_live_rows (follow with seg_counts_ready = 1).
 * loss_mse_out (nullable): without it the launch is the wave-per-ray kernel and the two loss scalars are left to
 * xr_train_loss_scalars; with it (caller zero-fills) the 16-lanes-per-ray kernel, which also ADDS [0] += scale * sum huber,
 * [1] += sum ((rgb - target) * alpha)^2 -- the two kernels' rgb_output / dloss_doutput differ like two fp32 summation orders.
 * (xr_version 120: this is the former `...train2`; the form without live_seg_count is retired.) */
int xr_composite_train(const float* network_output, const float* coords, const int32_t* rays_numsteps,
                       const int32_t* rays_numsteps_compacted, const float* bg_color, const float* target,
                       const float* alpha_mask, const float* density_grid_mean, uint32_t n_rays, int rgb_activation,
                       int density_activation, float delta, float scale, float* rgb_output, float* loss_mse_out,
                       float* dloss_doutput, uint32_t* live_seg_count, void* stream);
/* loss_mse_out[0] = scale * sum HuberLoss(rgb - target), [1] = sum ((rgb - target) * alpha)^2 (utils/metrics.py:8-16,
 * networks/hashnerf.py:36-44), WRITTEN, one workgroup's fixed-order sum: bit-reproducible run to run */
int xr_train_loss_scalars(const float* rgb, const float* target, const float* alpha_mask, uint32_t n_rays, float delta,
                          float scale, float* loss_mse_out, void* stream);
/* K5  calc_rgb_influence_api (src/calc_rgb.cu:330-389, kernel :144-206); bg is by value like the
 * reference's host tensor */
int xr_calc_rgb_inference(const float* network_output, const float* coords, const int32_t* rays_numsteps,
                          float bg_r, float bg_g, float bg_b, uint32_t n_rays, int rgb_activation,
                          int density_activation, float* rgb_output, float* alpha_output, void* stream);

/* Optional early-terminated rendering (the reference's K5 walks every sample; its EPSILON is unused,
 * calc_rgb.cu:178).  A frame is evaluated in depth slices [s0,s1) of per-ray sample indices:
 *   xr_render_slice_select   : rays with T > eps and more than s0 samples contribute rows
 *                              base+s0 .. base+min(n,s1)-1 to rows_out (compacted; count_out[0] = rows;
 *                              ray_offset_out[i] = first slot of ray i or -1)
 *   (encode + MLP on those rows through the `rows` arguments of xr_hashgrid_fwd / xr_nerf_mlp_fwd)
 *   xr_render_slice_composite: continues the front-to-back integration of K5 for the selected rays;
 *                              T [n_rays] and rgb_acc [n_rays,3] carry the state (init 1 and 0).
 * Final pixel = rgb_acc + T*bg, alpha = 1 - T; differs from K5 by < eps. */
int xr_render_slice_select(const int32_t* rays_numsteps, const float* T, uint32_t n_rays, uint32_t s0, uint32_t s1,
                           float eps, uint32_t* rows_out, int32_t* ray_offset_out, uint32_t* count_out, void* stream);
int xr_render_slice_composite(const float* raw_slice /*[count,4]*/, const float* coords, const int32_t* rays_numsteps,
                              const int32_t* ray_offset, uint32_t n_rays, uint32_t s0, uint32_t s1, int rgb_activation,
                              int density_activation, float* T, float* rgb_acc, void* stream);

/* K6  generate_grid_samples_nerf_nonuniform_api (src/generate_grid_samples_nerf_nonuniform.cu:44-87).  Coordinate d of point i goes
 * to positions[i * pos_row_stride + d * pos_comp_stride]: (3, 1) = the reference's [n,3] rows; (1, plane size) = three planes, the
 * layout xr_hashgrid_fwd reads with coalesced loads -- the sampler writes both K6 calls of a grid refresh into one plane buffer and
 * queries the density without a concatenation.  (xr_version 120: the former xr_generate_grid_samples2.) */
int xr_generate_grid_samples(const float* density_grid, uint32_t ema_step, uint32_t n_elements, uint32_t n_cascades /* = max_cascade+1 */,
                             float thresh, float aabb0, float aabb1, uint64_t rng_state, uint64_t rng_inc,
                             float* positions, uint32_t pos_row_stride, uint32_t pos_comp_stride, int32_t* indices /*[n]*/, void* stream);
/* K7  mark_untrained_density_grid_api (src/mark_untrained_density_grid.cu:54-82): writes 0 where
 * the cell is visible from any training camera, -1 elsewhere (the reference leaves visible cells of
 * its UNINITIALISED buffer untouched when they happen to be >= 0) */
int xr_mark_untrained_density_grid(const float* focal_lengths /*[n_img,2]*/, const float* xforms /*[n_img,4,3]*/,
                                   uint32_t n_elements, uint32_t n_images, int resolution0, int resolution1,
                                   float* density_grid, void* stream);
/* K8  splat_grid_samples_nerf_max_nearest_neighbor_api (src/splat_...cu:30-57) */
int xr_splat_grid_samples(const float* mlp_out, const int32_t* indices, uint32_t padded_output_width,
                          uint32_t n_samples, float* density_grid_tmp, void* stream);
/* K9  ema_grid_samples_nerf_api (src/ema_grid_samples_nerf.cu:29-50) */
int xr_ema_grid_samples(const float* density_grid_tmp, uint32_t n_elements, float decay, float* density_grid,
                        void* stream);
/* K10+K11  update_bitfield_api (src/update_bitfield.cu:74-116): mean of max(v,0) over level 0
 * (fixed-order tree => bit-reproducible), threshold min(0.01, mean), bitfield + 7 max-pools.
 * density_grid_mean: device, [0] written.  workspace: xr_update_bitfield_workspace_bytes(). */
size_t xr_update_bitfield_workspace_bytes(void);
int xr_update_bitfield(const float* density_grid, float* density_grid_mean, uint8_t* bitfield, void* workspace,
                       size_t workspace_bytes, void* stream);
/* K9 + K10 + K11 of one grid refresh as THREE launches (the refresh tail of NGPGridSampler.update_density_grid,
 * ngp_grid_sampler.py:150-174): xr_ema_grid_samples over the first n_elements cells followed by xr_update_bitfield -- the same density
 * grid, mean and bitfield bit for bit (K10's partial sums are taken where K9 walks cascade 0, every workgroup of the bits kernel folds
 * the partials itself, and the seven dependent max-pool launches become one: csrc/xr_grid.hip).  Same workspace as
 * xr_update_bitfield (which runs the last two of these launches behind K10's partial sums). */
int xr_ema_update_bitfield(const float* density_grid_tmp, uint32_t n_elements, float decay, float* density_grid,
                           float* density_grid_mean, uint8_t* bitfield, void* workspace, size_t workspace_bytes, void* stream);
/* K11 alone with the mean supplied (device pointer) -- for bit-exact tests against the reference */
int xr_bitfield_from_mean(const float* density_grid, const float* density_grid_mean, uint8_t* bitfield,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * tiny-cuda-nn surface used by xrnerf/models/mlps/hashnerf_mlp.py:34-45,55-111
 *
 * Hash grid (tcnn.Encoding otype=HashGrid).  Level geometry is computed ONCE on the host
 * (xr_hashgrid_meta) and passed to every call, so that host oracle and device agree on indices.
 * Table layout: per level `offset[l]` entries of F=2 floats, levels consecutive (tcnn `params`).
 * Positions are read with an element stride (`x_stride` floats between samples) so the [S,7]
 * coordinate rows of K1 can be consumed in place.
 * Encoded features are FEATURE-MAJOR: enc_t [2*n_levels][ld] with ld >= n (what both the encoder's
 * stores and the MLP's MFMA operand loads want for coalescing). */
void xr_hashgrid_meta(int n_levels, int log2_hashmap_size, int base_resolution, double per_level_scale,
                      float* scale_host, uint32_t* resolution_host, uint32_t* offset_host /*[L+1]*/);
/* `n_dev` (nullable, device): when given, only min(n, *n_dev) rows are processed -- the row count
 * then never has to be read back to the host (n is the launch-sizing upper bound). */
/* `rows` (nullable, device): sample i reads its position from row rows[i] of x (render depth slices) */
/* Coordinate d of sample i is x[i * x_stride + d * x_comp_stride]: (7, 1) consumes K1's [S,7] coordinate rows in place, (3, 1) plain
 * [n,3] rows; x_stride = 1 with x_comp_stride = plane size reads positions stored as three planes (structure of arrays): three
 * coalesced dword loads per sample instead of three strided ones out of 28-byte rows.  (xr_version 120: the former `..._fwd2`.) */
int xr_hashgrid_fwd(const float* table, const float* x, uint32_t x_stride, uint32_t x_comp_stride, uint32_t n, const uint32_t* n_dev,
                    const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                    const uint32_t* offset_host, float* enc_t, uint32_t ld, void* stream);
/* Backward of the encoding above (tcnn's kernel_grid_backward as reached from hashnerf_mlp.py:59-61 under autograd):
 * grad_table[idx,f] += w * denc_t[2l+f][i]; the caller zero-fills grad_table (with XR_SCATTER_OVERWRITE: no).
 * workspace (nullable; xr_hashgrid_bwd_workspace_bytes(n, n_levels, resolution_host, offset_host), 16-byte
 * aligned): with it and n >= 16384 NO level touches the table with an atomic -- hashed levels and the larger dense levels
 * are binned by table partition (16-byte items in workgroup-private sub-bins, overflow lists for clustered inputs) and
 * one workgroup per partition sums its bins in LDS in 64-bit fixed point (scaled per level from the launch's largest gradient: integer
 * sums, so the result is the same bits whatever the schedule); dense levels up to 2^16 entries are run-length reduced per
 * thread into workgroup-private LDS partitions whose per-chunk partials are folded in fixed order (csrc/xr_scatter.hip).
 * Three launches on `stream`, no event.  A non-finite gradient makes every entry of its level NaN.
 * Without a workspace, for small n and for table shapes outside those rules a level takes the atomic scatter.
 * Positions must lie in the unit cube (the sampler's aabb): a hashed level's x-neighbour pair is kept in one partition
 * by x < 2^13. */
size_t xr_hashgrid_bwd_workspace_bytes(uint32_t n, int n_levels, const uint32_t* resolution_host,
                                       const uint32_t* offset_host);
/* rows (nullable, with n_dev): launch sample j is row rows[j] of x / denc_t and *n_dev the list's length -- the live-row
 * list of xr_live_rows, so that the scatter never touches the rows whose gradient is exactly zero.
 * flags: 0 = grad_table is ADDED to (caller zero-fills); XR_SCATTER_OVERWRITE: the table slices of the call's levels are WRITTEN
 * (grad = scatter result) -- the caller needs no zero-fill (the training step saves a 48.8-MB one); with n == 0 the slices are
 * zero-filled.  (xr_version 120: the former `..._bwd2`.) */
#define XR_SCATTER_OVERWRITE 1
int xr_hashgrid_bwd(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n,
                    const uint32_t* n_dev, const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                    const uint32_t* offset_host, float* grad_table, void* workspace, size_t workspace_bytes, int flags,
                    void* stream);
/* A HELPER stream for xr_hashgrid_bwd's fallback: geometries where some levels take the atomic kernel and others the binned path run the
 * two beside each other on it.  The library creates no stream or event: the caller hands over one stream and two events (fork / join;
 * plain hipEvent_t, timing not needed) once per host thread; null stream = none, everything then runs in order on the caller's stream
 * (same results).  (Rounds 3-5 also ran the binned scatter's small dense levels and xr_ngp_train_step's small sums and updates on it;
 * since round 6 those ride inside the scatter's own launches on the caller's stream -- a training iteration records and waits for no event.) */
int xr_set_helper_stream(void* stream, void* fork_event, void* join_event);
/* tcnn.Encoding otype=SphericalHarmonics degree 4; dirs in [0,1] (the sampler's warp_direction);
 * out row-major [n,16] */
int xr_sh4(const float* dirs, uint32_t dir_stride, uint32_t n, float* out, void* stream);

/* HashNerfMLP.run_mlp fused (hashnerf_mlp.py:55-79): density_net (32 -> nhd x 64 -> 16) on the encoded
 * features, SH-4 of the view direction, color_net (15+16 (+1 pad = pad_value) -> nhc x 64 -> 16),
 * raw [n,4] = [r,g,b,sigma].
 * Weights: tcnn `params` layout = row-major [out,in] matrices in layer order, out padded to 16.
 * dirs may be NULL (run_density, hashnerf_mlp.py:107-111: only raw[:,3] is meaningful then).
 * Depths: (1, 2) -- what the config's `num_layers` says -- keeps every activation in registers and both weight sets in LDS (so do
 * (1, 1), (2, 2), (2, 3) under XR_MLP_F32); ANY other 1..8 + 1..8 hidden layers, tcnn's own default 5 + 5 first of all (tcnn reads
 * `n_hidden_layers` and ignores the `num_layers` key of configs/instant_ngp/nerf_blender_local01.py:106-124), runs on the streamed
 * kernels: a workgroup takes its sample tiles through one layer at a time, the layers' weights pass through LDS (arithmetic of
 * XR_MLP_F16X2; XR_MLP_F16 is built for (1, 2) only).
 * `arithmetic` -- ONE entry point, four kernel families; the first, third and fourth are the fp32 "parity mode" (1e-4 on raw):
 *   XR_MLP_F32     v_mfma_f32_32x32x2_f32 == an fmaf chain, exact fp32 products.
 *   XR_MLP_F16     the reference's precision: tiny-cuda-nn computes FullyFusedMLP in fp16 with fp32 accumulation (the reference casts
 *                  its half outputs to fp32, hashnerf_mlp.py:76-77).  Parameters and gradients stay fp32 in memory; weights /
 *                  activations are rounded to fp16 inside the kernel (v_mfma_f32_32x32x16_f16), gradients carry tcnn's loss scale 128
 *                  through the fp16 stages.
 *   XR_MLP_BF16X3  every fp32 operand split EXACTLY into three bf16 numbers (8 + 8 + 8 significand bits), a product carried by the six
 *                  bf16 x bf16 MFMA terms above 2^-23 of it, fp32 accumulate: fp32-rounding accuracy at 0.375 of the fp32-MFMA
 *                  matrix-core time; results differ from XR_MLP_F32 the way two fp32 summation orders differ.  (1, 2) only.
 *   XR_MLP_F16X2   (round 5; what the host side selects by default) x = xh + xl with xh = fp16(x), xl = fp16(x - xh) keeps 22 significand
 *                  bits, and the three products wh.xh + wh.xl + wl.xh (each exact in the fp32 accumulator) carry w.x to ~2^-21 of
 *                  |w||x| -- measured 3.4e-7 .. 4.6e-7 of max|raw| against a float64 statement where the fp32 MFMA is at 1.7e-7 ..
 *                  3.0e-7 (profiles/r05_mlp_fwd_f16x2_split_probe.txt) -- with HALF the matrix instructions and ~60 % of the
 *                  conversions of the 3-way bf16 split: 27.5 us against 42.6 us at 2^18 rows.  Range: hidden activations up to 65504
 *                  and hash-grid features up to 4e3 in magnitude (the features enter the first layer scaled by 2^4, exactly; the
 *                  reference's own fp16 tcnn overflows at 65504 too); low parts below 2^-14 are fp16 subnormals (absolute precision
 *                  2^-25).  Any depth.  Out of range is SATURATED, not inf: every operand is clamped to +-65504 before it is split
 *                  (hidden activations in the ReLU's own instruction), forward and backward alike, and the forward counts the waves
 *                  that met such an operand in the caller's range word (xr_set_mlp_range_word below).
 * (xr_version 120 -> 121: the four arithmetics used to be four entry points, xr_nerf_mlp_fwd / _f16 / _bf16x3 / _f16x2.) */
#define XR_MLP_F32 0
#define XR_MLP_F16 1
#define XR_MLP_BF16X3 2
#define XR_MLP_F16X2 3
/* XR_MLP_F16X2's range word: one uint32 in device memory, the caller's (per host thread, like the helper stream; NULL = none).  Every
 * XR_MLP_F16X2 forward launched by this thread afterwards -- xr_nerf_mlp_fwd, xr_nerf_density_splat, and inside xr_ngp_train_step /
 * xr_ngp_loop_run / the frame entry points -- adds one per wave that split an operand (feature x 2^4, activation, colour input or weight)
 * above 65504 in magnitude.  Nothing reads it on the device; the host reads it when it synchronises anyway (the trainer: at the grid
 * refresh) and XR_MLP_F32 is the escape when it is not zero.  With NULL the forwards run WITHOUT the count (the saturation stays): the
 * maximum behind it costs 0.5 VALU instruction per operand on a kernel bound by those (27.5 -> 31.9 us at 2^18 rows), so the host side
 * hands the word over for the refresh iterations, frames and direct calls and takes it back for the iterations in between. */
int xr_set_mlp_range_word(uint32_t* word);
int xr_nerf_mlp_fwd(int arithmetic, const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                    const uint32_t* n_dev, const uint32_t* rows /* nullable: dirs row of sample i */, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                    float pad_value, float* raw /*[n,4]*/, void* stream);
/* backward of the above given dL/draw [n,4]: writes denc_t [32][ld] (for xr_hashgrid_bwd) and
 * ACCUMULATES weight gradients into grad_w_density / grad_w_color (caller zero-fills).
 * Activations are recomputed in-kernel (nothing saved by the forward).
 * workspace: xr_nerf_mlp_bwd_workspace_bytes(n, n_hidden_density, n_hidden_color) -- list of live rows first (its position does
 * not depend on the depths), then the per-workgroup weight-gradient partials, then (streamed depths) the activation scratch area.
 * Streamed depths (anything but (1, 2)): the forward is recomputed with the forward's own arithmetic (2-way fp16 split: the same ReLU
 * decisions bit for bit), the gradient chain and the weight-gradient products on 2-way split bf16 operands like the default below. */
size_t xr_nerf_mlp_bwd_workspace_bytes(uint32_t n, int n_hidden_density, int n_hidden_color);
/* live_rows / n_live (both or neither): the list xr_live_rows built from `draw`.  The backward then computes exactly the
 * listed rows and leaves the other rows of denc_t UNTOUCHED (pass the same list to xr_hashgrid_bwd).  Without a list
 * the call builds its own in the workspace and writes exact zeros to the dead rows of denc_t (same results as the
 * backward over every row, which XR_MLP_LIVE=0 still runs for measurement).
 * `arithmetic`: XR_MLP_F16 = the reference-precision backward (topology (1, 2)); the three fp32 forwards share ONE backward, whose
 * products are chosen by XR_MLP_BWD_DW (environment, read per call; topology (1, 2)): the weight-gradient products and the gradient chain
 * run on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into two bf16 parts (x = xh + xl, three products kept: 2^-16 relative per
 * product, gradients within 2e-5 of their scale of the all-fp32 kernel), and the activations are recomputed
 *   "h2f" (default) with XR_MLP_F16X2's arithmetic, product for product -- the ReLU decisions of the default forward, bit for bit;
 *   "b2x" on the fp32 MFMA (the ReLU decisions of XR_MLP_F32);   "b2f" on 2-way split bf16 operands (a hidden unit within ~1e-5 of zero can
 *   land on the other side of its ReLU than in the forward);   "b2": only the weight-gradient products split;   "f32": every product on the
 *   fp32 MFMA. */
int xr_nerf_mlp_bwd(int arithmetic, const float* enc_t, uint32_t ld, const float* dirs, uint32_t dir_stride, uint32_t n,
                    const uint32_t* n_dev, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                    float pad_value, const float* draw /*[n,4]*/, float* denc_t, float* grad_w_density,
                    float* grad_w_color, void* workspace, size_t workspace_bytes, const uint32_t* live_rows,
                    const uint32_t* n_live, void* stream);
/* Rows of dL/d(raw) [n,4] that are not exactly (0,0,0,0), in order: live_rows[0 .. *n_live).  A sample behind an opaque
 * surface has transmittance exactly 0 in fp32 (calc_rgb.cu:108-135: every term of the row carries the weight alpha*T, or T times a suffix colour that is exactly 0 there), so its
 * row is exactly zero and contributes exactly nothing to dW or to the table gradient -- in steady-state training more
 * than half of the marched samples.  n_live: FOUR words -- [0] the list's length, [1] / [2] running totals of live / valid rows
 * over the calls since the caller last cleared them (statistics: bench.py reports the live fraction from them), [3] spare.
 * seg_count: scratch of XR_LIVE_ROWS_SEGMENTS(n) words.  zero_denc_t (nullable,
 * [32][ld]): the dead rows of it are set to zero.  Stable order and a fixed partition: reproducible run to run. */
#define XR_LIVE_SEGMENT_ROWS 1024u
#define XR_LIVE_ROWS_SEGMENTS(n) (((n) + XR_LIVE_SEGMENT_ROWS - 1u) / XR_LIVE_SEGMENT_ROWS)
/* the list area inside an xr_nerf_mlp_bwd workspace of n rows (unused by a backward that is handed a list) */
int xr_nerf_mlp_bwd_list_slots(void* workspace, size_t workspace_bytes, uint32_t n, uint32_t** live_rows, uint32_t** seg_count,
                               uint32_t** n_live);
/* seg_counts_ready != 0: seg_count already holds the per-segment counts (xr_composite_train counted them): the ranking pass only */
int xr_live_rows(const float* dloss_doutput, uint32_t n, const uint32_t* n_dev, uint32_t* seg_count, uint32_t* live_rows,
                 uint32_t* n_live, float* zero_denc_t, uint32_t ld, int seg_counts_ready, void* stream);

/* The table scatter with the optimiser's update applied in place of the gradient write: the scatter owns every entry of its
 * levels exactly once per launch, so it can run Adam (+ L2 weight decay, + the EMA copy) on (param, m, v, ema) where an entry's
 * gradient is complete -- same update, bit for bit, as xr_hashgrid_bwd(XR_SCATTER_OVERWRITE) followed by xr_adam_step_multi
 * on the table, without the 48.8-MB gradient write + read and without the separate HBM-bound launch.  No gradient is produced.
 * Single-GPU training only (a data-parallel step must reduce the gradient first).  Needs a non-atomic path for every level of this
 * geometry / row capacity, else the call fails before launching anything; adam == NULL is the dry run of exactly that check (XR_OK =
 * supported; nothing else is read, nothing is launched). */
typedef struct xr_adam_fuse {
    float* param; float* m; float* v; float* ema;   /* whole tensors, 16-byte aligned; ema nullable */
    int step;                                       /* 1, 2, ... (this update's bias correction) */
    float lr, beta1, beta2, eps, weight_decay, ema_momentum, grad_scale;
    uint64_t n;                                     /* floats in the tensor (read where the caller cannot know it: xr_ngp_train_step's mlp_adam) */
} xr_adam_fuse;
int xr_hashgrid_bwd_adam(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                         const uint32_t* rows, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                         const uint32_t* offset_host, void* workspace, size_t workspace_bytes, const xr_adam_fuse* adam,
                         void* stream);

/* K9's density query and K8 in one launch (grid refresh: ngp_grid_sampler.py:103-137 -> hashnerf_mlp.py:107-111 +
 * splat_grid_samples_nerf_max_nearest_neighbor.cu:7-28): the density network over n encoded points (enc_t as xr_hashgrid_fwd writes it),
 * each result's optical thickness exp(density) * min_step merged into density_grid_tmp[indices[i]] by an order-free maximum from the
 * forward kernel's epilogue -- the same values xr_splat_grid_samples would merge, without the [n,4] network output in HBM and without
 * the 2^20-thread launch.  mlp_mode = the `arithmetic` of xr_nerf_mlp_fwd. */
int xr_nerf_density_splat(int mlp_mode, const float* enc_t, uint32_t ld, uint32_t n, const float* w_density, int n_hidden_density,
                          int n_hidden_color, const int32_t* indices, float* density_grid_tmp, void* stream);

/* One training step's device work of HashNerfNetwork.train_step (networks/hashnerf.py:32-52, optimiser excluded) as ONE call:
 * xr_hashgrid_fwd -> xr_nerf_mlp_fwd[_f16] -> xr_composite_train (which counts the live rows per segment) -> xr_live_rows -> xr_nerf_mlp_bwd[_f16] ->
 * xr_hashgrid_bwd(XR_SCATTER_OVERWRITE), on `stream`:
 * grad_table's slices of the scattered levels are WRITTEN (no zero-fill; whatever they held is gone).
 * table_adam (nullable): the scatter becomes xr_hashgrid_bwd_adam -- the table is UPDATED by this call and grad_table (then
 * nullable) is not written; single GPU, scatter_level0 == 0.
 * w_density_adam / w_color_adam (both or neither, same step and constants): xr_adam_step_multi's update of the two MLP tensors right
 * behind the fixed-order sum of their gradient partials -- both inside the scatter's binning launch (same bits as the two launches).
 * live_seg_count (nullable): XR_LIVE_ROWS_SEGMENTS(n_rows) words for the per-segment counts, ZERO on entry; the call leaves them
 * zero again (cleared inside the scatter's accumulate launch).  Null: the slot in ws_mlp_bwd, cleared by a fill on `stream`.
 * zero_block / zero_floats: kept for the layout (grad_w_density, grad_w_color, loss_mse live in it); nothing in it is zero-filled
 * any more -- the two gradient buffers and loss_mse[0..1] are WRITTEN.
 * coords: K1's [n_rows,7] rows (positions / directions consumed in place); n_dev: device count of valid rows; every buffer
 * is caller-owned (enc_t / denc_t [32][ld], raw / draw [n_rows,4], rgb_out [n_rays,3]); zero_draw != 0 also clears draw
 * (needed only without n_dev).  mlp_mode: the `arithmetic` of xr_nerf_mlp_fwd / _bwd (XR_MLP_F16X2 is the default of the host
 * package).  scatter_level0: the step scatters hash levels [scatter_level0, n_levels) only (0 = all) -- a
 * data-parallel caller hands that slice of grad_table to its gradient collective and then scatters the coarser levels with
 * xr_hashgrid_bwd(XR_SCATTER_OVERWRITE) on the same row list (xr_nerf_mlp_bwd_list_slots), so the exchange runs under the rest
 * of the backward.
 * Same kernels and results as the separate calls -- this exists because issuing them one by
 * one from an interpreter costs as much host time as the kernels take on the device. */
int xr_ngp_train_step(const float* table, const float* w_density, const float* w_color, int n_hidden_density, int n_hidden_color,
                      float pad_value, int mlp_mode, int n_levels, const float* scale_host, const uint32_t* resolution_host,
                      const uint32_t* offset_host, const float* coords, uint32_t n_rows, const uint32_t* n_dev,
                      const int32_t* rays_numsteps, const int32_t* rays_numsteps_compacted, uint32_t n_rays, const float* bg_color,
                      const float* target, const float* alpha_mask, const float* density_grid_mean, int rgb_activation,
                      int density_activation, float huber_delta, float loss_scale, float* enc_t, uint32_t ld, float* raw,
                      float* draw, float* denc_t, float* rgb_out, float* zero_block, size_t zero_floats, float* grad_w_density,
                      float* grad_w_color, float* loss_mse, uint32_t* live_seg_count, float* grad_table, size_t table_floats, int zero_draw,
                      void* ws_mlp_bwd, size_t ws_mlp_bwd_bytes, void* ws_scatter, size_t ws_scatter_bytes, int scatter_level0,
                      const float* xyz_planes, uint32_t plane_stride, const xr_adam_fuse* table_adam, const xr_adam_fuse* w_density_adam,
                      const xr_adam_fuse* w_color_adam, const char* timed_entry, void* timing_begin, void* timing_end, void* stream);
/* xyz_planes (nullable): the positions of `coords` once more as three planes of plane_stride floats (xr_rays_sampler /
 * xr_ngp_window_march write them); the encoder then reads them with coalesced loads (xr_hashgrid_fwd). */
/* timed_entry (nullable): the name of ONE of the entry points the step runs ("xr_hashgrid_fwd", "xr_nerf_mlp_fwd",
 * "xr_composite_train", "xr_live_rows", "xr_nerf_mlp_bwd", "xr_hashgrid_bwd"): its launches are bracketed on `stream` by the two
 * events -- bench.py's live duration of the dominant kernel inside the timed region.  A caller with a HIP binding of its own passes its own
 * hipEvent_t (timing enabled); for callers without one (ctypes) the four calls below make, record, read and destroy such events.
 */
void* xr_timing_event_create(void);
int xr_event_record(void* event, void* stream);
int xr_timing_event_destroy(void* event);
int xr_timing_event_elapsed_ms(void* begin, void* end, float* ms);

/* ------------------------------------------------------------------------------------------
 * The marches of a refresh WINDOW as one series of launches, and the training LOOP between two grid refreshes as one call.
 * The reference's loop is mmcv's IterBasedRunner calling HashNerfNetwork.train_step once per iteration
 * (xrnerf/core/apis/train.py:58-66, networks/hashnerf.py:32-52) with three hooks around it (core/hooks/hash_hook.py:12-42); its
 * sampler refreshes the occupancy bitfield at iterations = 0 (mod update_grid_freq = 16) only (samplers/ngp_grid_sampler.py:194-197) and
 * changes the batch size at iterations = 15 (mod 16) only (:268-281), and K1 reads no weights (ray_sampler.cu:5-116).  So right behind a
 * refresh every batch of the window can be drawn and marched: xr_ngp_window_march does that for up to XR_NGP_WINDOW iterations with ONE
 * launch per kernel (batch assembly, K1 count, K1 write, K2 clip) and one copy of the (rays, samples) counters to pinned host memory --
 * bit for bit the batches, samples and RNG call indices of one xr_make_batch_series(1) / xr_rays_sampler / xr_clip_numsteps sequence per iteration.
 * The window's buffers are caller-owned: chunk c (iteration it lives in chunk it % XR_NGP_WINDOW) sits at fixed strides. */
#define XR_NGP_WINDOW 16
typedef struct xr_ngp_window {
    float *rays_o, *rays_d, *target, *alpha, *bg; int32_t* img_ids;   /* xr_make_batch_series outputs; chunk c at row c * ray_stride */
    int32_t *rays_index, *rays_numsteps, *numsteps_clipped;           /* K1 / K2 per-ray outputs ([.,1], [.,2], [.,2]); chunk c at row c * ray_stride */
    uint32_t ray_stride;                                              /* rows per chunk of the per-ray arrays (>= n_rays) */
    float* coords; size_t coords_stride;                              /* chunk c: coords + 7 * c * coords_stride, coords_stride >= max_samples rows */
    float* xyz_planes; uint32_t plane_stride;                         /* nullable; chunk c's plane k: xyz_planes + (3 c + k) * plane_stride */
    uint32_t *counter2, *n_valid;                                     /* [XR_NGP_WINDOW][2] each: K1's (rays, samples), K2's valid row count (twice) */
} xr_ngp_window;
/* chunks [first_chunk, first_chunk + n_chunks): the first `batches_ready` of them already hold their batch (the caller drew it with
 * xr_make_batch_series into the chunk's rows), the others are drawn here from the device-resident [n_table_rays, 11] table with the cursor
 * *cur_ray (in / out; a batch that would run over the end starts at row 0) and the batch generator's call indices batch_call_index ...;
 * K1 runs for all n_chunks with call indices k1_call_index ...; counter_host_pinned (nullable): [XR_NGP_WINDOW][2] pinned words,
 * the chunks' pairs are copied to their slots.  workspace: xr_rays_sampler_workspace_bytes(n_rays, n_chunks). */
int xr_ngp_window_march(const xr_ngp_window* window, uint32_t first_chunk, uint32_t n_chunks, uint32_t batches_ready, uint32_t n_rays,
                        const float* rays_rgb_rows, uint64_t n_table_rays, uint64_t* cur_ray, uint64_t batch_seed, uint64_t batch_call_index,
                        const uint8_t* bitfield, float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples,
                        uint64_t k1_call_index, uint32_t max_compacted, void* workspace, size_t workspace_bytes,
                        uint32_t* counter_host_pinned, void* stream);
/* n_series <= XR_NGP_WINDOW batches in one launch (the batch assembly of xr_ngp_window_march on its own) */
int xr_make_batch_series(const float* rays_rgb_rows, const uint64_t* row0_host, uint32_t n, uint32_t n_series, uint32_t ray_stride,
                         uint64_t rng_state, uint64_t rng_inc, float* rays_o, float* rays_d, float* target, float* alpha, float* bg,
                         int32_t* img_ids, void* stream);

/* Driven from an interpreter the enqueue of one iteration costs 0.36 ms of host time against 0.4 ms of device work; xr_ngp_loop_run
 * enqueues k iterations -- each one xr_ngp_train_step on its chunk of the marched window, with the optimiser's updates inside (three
 * xr_adam_fuse) -- from native code on ONE in-order stream: same entry points, same order, hence the same parameters bit for bit as k
 * single calls.  The caller orders `stream` behind the window's marches once; nothing is allocated, recorded or waited for here. */
typedef struct xr_ngp_step_set {       /* one of the TWO alternating sets of step buffers (see xr_ngp_train_step) */
    float *enc_t, *raw, *draw, *denc_t, *rgb_out, *zero_block; size_t zero_floats;
    float *grad_w_density, *grad_w_color, *loss_mse; uint32_t* live_seg_count;
    float* grad_table;                 /* data parallel only: the table gradient of this set (zero1: padded to world * shard floats) */
} xr_ngp_step_set;
/* The gradient exchange of a data-parallel loop (N > 1 ranks; the reference: MMDistributedDataParallel's all-reduce behind the
 * backward pass, xrnerf/core/apis/train.py:28-38).  Each call enqueues ONE collective ordered behind everything enqueued on `stream`
 * so far and returns at once (the collective runs on the implementation's own stream, under what the caller enqueues next); `finish`
 * orders `stream` behind every collective begun since the last finish.  Implementations: xr_rccl_exchange (RCCL driven from native
 * code) or caller-provided function pointers (e.g. a ctypes callback issuing torch.distributed collectives: the gloo tests). */
typedef struct xr_grad_exchange {
    int (*all_reduce)(void* ctx, float* buf, size_t n, void* stream);                                   /* in-place sum */
    int (*reduce_scatter)(void* ctx, const float* send, float* recv, size_t n_recv, void* stream);      /* zero1 only */
    int (*all_gather)(void* ctx, const float* send, float* recv, size_t n_send, void* stream);          /* zero1 only */
    int (*finish)(void* ctx, void* stream);
    void* ctx; int world_size, rank;
} xr_grad_exchange;
typedef struct xr_ngp_loop_desc {
    float *table, *w_density, *w_color; int n_hidden_density, n_hidden_color; float pad_value; int mlp_mode;
    int n_levels; const float* scale_host; const uint32_t *resolution_host, *offset_host;
    xr_adam_fuse adam_table, adam_w_density, adam_w_color;  /* tensors and constants; step / lr / ema_momentum are set per iteration */
    const float* density_grid_mean; int rgb_activation, density_activation; float huber_delta, loss_scale;
    uint32_t n_rows, ld;                                   /* rows of the step (= K2's clip), leading dimension of the feature planes */
    xr_ngp_window window; xr_ngp_step_set step[2];
    void* ws_mlp_bwd; size_t ws_mlp_bwd_bytes; void* ws_scatter; size_t ws_scatter_bytes;
    void* stream;
    /* data parallel (null exchange = one GPU: the three updates inside the step).  Per iteration: the step writes its gradients
     * (hash levels >= split_level first), the buckets go to the exchange as they complete (MLP gradients, fine levels, then the coarse
     * levels scattered underneath the fine bucket's collective), `finish`, then ONE xr_adam_step_multi with grad_scale = 1 / world.
     * dp_mode 1 (zero1): reduce-scatter of the padded table gradient into shard_grad, the optimiser on this rank's
     * shard (adam_table then describes the SHARD: param = table_padded + rank * shard_floats), all-gather of the updated shards. */
    const xr_grad_exchange* exchange; int dp_mode; int split_level;
    float *shard_grad, *table_padded; uint64_t shard_floats;
} xr_ngp_loop_desc;
typedef struct xr_ngp_loop_state {     /* the counters the loop shares with its caller (read AND written) */
    uint64_t iter;                     /* next iteration */
    uint32_t step_turn;                /* the next step takes set step_turn ^ 1 */
    int32_t adam_step;                 /* updates applied so far */
    uint32_t last_step_set;            /* (out) the set iteration iter - 1 used */
} xr_ngp_loop_state;
/* k <= XR_NGP_WINDOW iterations iter .. iter + k - 1, all of them marched (chunks iter % XR_NGP_WINDOW ...); n_rays = the window's batch
 * size; lr / ema_momentum: k values each (the schedules are the caller's).
 * timed_entry + timing_events [2k] (nullable): one entry point of every step bracketed by a pair of timing events;
 * iter_events [k + 1] (nullable): timing events recorded on `stream` in front of every iteration and behind the last. */
int xr_ngp_loop_run(const xr_ngp_loop_desc* desc, xr_ngp_loop_state* state, uint32_t k, uint32_t n_rays, const float* lr,
                    const float* ema_momentum, const char* timed_entry, void* const* timing_events, void* const* iter_events);
/* RCCL driven from native code (csrc/xr_dist.hip): librccl is dlopen'ed on first use (librccl_path nullable: a copy the process
 * already mapped, else the system library).  xr_rccl_unique_id on ONE rank -> 128 bytes the caller hands to every rank ->
 * xr_rccl_create on every rank (collective; the current device) -> xr_rccl_exchange fills the hooks.  This is the one handle the
 * library creates (a communicator cannot live in caller-provided memory). */
int xr_rccl_unique_id(const char* librccl_path, void* id128);
void* xr_rccl_create(const char* librccl_path, const void* id128, int world_size, int rank);
int xr_rccl_destroy(void* handle);
int xr_rccl_exchange(void* handle, xr_grad_exchange* out);
/* measured exposure of the exchange: with timing on, every `finish` that waits for something brackets the wait with two events on the
 * caller's stream.  The call reports (three non-null results; after a device synchronisation) mean / max of the last <= 64 of them and
 * the total count, THEN sets the recording state: timing_on != 0 starts a new record, 0 stops recording. */
int xr_rccl_exposed_ms(void* handle, int timing_on, float* mean_ms /*nullable*/, float* max_ms, int* count);

/* tcnn.Network(FullyFusedMLP) on its own (compatibility surface; the hot path uses the fused kernels above):
 * x [n, n_in] with arbitrary row / column strides (in floats), n_in <= 32, missing input columns = pad_value;
 * weights in the tcnn layout; y / dy [n,16] row-major (columns >= n_output_dims are padding); dx [n, n_in]
 * row-major (nullable).  grad_w ACCUMULATES.  n_hidden: forward 1..3, backward 1..2. */
int xr_mlp_fwd(const float* x, long row_stride, long col_stride, int n_in, float pad_value, uint32_t n, const float* w,
               int n_hidden, float* y, void* stream);
size_t xr_mlp_bwd_workspace_bytes(int n_hidden);
int xr_mlp_bwd(const float* x, long row_stride, long col_stride, int n_in, float pad_value, uint32_t n, const float* w,
               int n_hidden, const float* dy, float* dx, float* grad_w, void* workspace, size_t workspace_bytes,
               void* stream);

/* ------------------------------------------------------------------------------------------
 * callers either side of the path
 * ray generation, get_rays_np_hash (xrnerf/datasets/load_data/get_rays.py:35-69) in fp32:
 * rows [row0, row0+nrows) of an H x W image; pose_host = the python [4,3] matrix. */
int xr_gen_rays(const float* pose43_host, int H, int W, float fx, float fy, float cx, float cy, int row0,
                int nrows, float* rays_o, float* rays_d, void* stream);
/* loss = scale * sum huber_delta(rgb - target) (networks/hashnerf.py:37-44, utils/metrics.py); writes dL/drgb and ADDS the loss into
 * loss_out[0] (caller zero-fills).  alpha (nullable, [n_elems / 3]): also the alpha-masked squared error sum the reference turns into its
 * logged PSNR (networks/hashnerf.py:40-42): loss_out[1] += sum ((rgb - target) * alpha)^2 -- loss_out then has two floats. */
int xr_huber_loss_grad(const float* rgb, const float* target, const float* alpha, uint32_t n_elems, float delta, float scale,
                       float* grad, float* loss_out, void* stream);
/* HashBatchSample + RandomBGColor (datasets/pipelines/create.py:153-191, augment.py:290-317): xr_make_batch_series below, one launch for
 * 1..XR_NGP_WINDOW batches -- rows [n,11] = (o3, d3, rgba4, img_id) of the device-resident ray table -> batch tensors; bg ~ U[0,1) (PCG32) */
/* gradients[k] *= (*scale_dev) * host_factor for up to 4 tensors in one launch (scale_dev nullable = 1): the
 * incoming-gradient scaling of the fused train step (networks/hashnerf.py:24-43 leaves it to autograd) and the
 * 1/world_size of data-parallel averaging; a factor of exactly 1 costs no memory traffic. */
int xr_scale_multi(int n_tensors, float* const* tensors, const size_t* n, const float* scale_dev, float host_factor,
                   void* stream);
/* torch.optim.Adam step with L2 weight decay (configs/instant_ngp/nerf_blender_local01.py:14-18), fused with the optional EMA copy of
 * mmcv's EMAHook (:24): ema = (1-mom)*ema + mom*p -- for 1..4 parameter tensors in ONE launch; arrays are HOST arrays of device pointers /
 * element counts (ema may be NULL, or hold NULL entries).  grad_scale: factor on the gradients as they are read (1 = none;
 * 1/world_size after a summing all-reduce -- bit for bit the update of `g *= grad_scale` followed by this call, without
 * that pass over the gradients; the gradient buffers themselves are not modified). */
int xr_adam_step_multi(int n_tensors, float* const* p_host, const float* const* g_host, float* const* m_host,
                       float* const* v_host, float* const* ema_host, const size_t* n_host, int step, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float ema_momentum, float grad_scale,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Mip-NeRF (BASELINE config #3; configs/mipnerf/mipnerf_multiscale.py): the sampling / encoding / rendering
 * stages either side of the 8x256 MLP, one launch each.  Tensors are contiguous fp32; n_z = number of interval
 * EDGES per ray (config: 129), so a ray has n_z-1 samples.
 *
 * GetZvals (xrnerf/datasets/pipelines/create.py:486-531): near/far [n_rays]; z_rand [n_rays,n_z] holds the
 * uniform draws the reference takes from torch.rand (NULL = not randomized); z_out [n_rays,n_z]. */
int xr_mip_zvals(const float* near, const float* far, uint32_t n_rays, uint32_t n_z, int lindisp,
                 const float* z_rand, float* z_out, void* stream);
/* row width of the encoding: 6*(max_deg-min_deg) + 6*(max_deg_view-min_deg_view) (+3 with append_identity)
 * (MipNerfEmbedder.get_embed_ch, xrnerf/models/embedders/mipnerf_embedder.py:76-83) */
uint32_t xr_mip_encode_channels(int min_deg, int max_deg, int min_deg_view, int max_deg_view, int append_identity);
/* sample_along_rays / cast_rays (xrnerf/models/networks/utils/mip.py:96-148, diag covariance, stable cone
 * formula) + MipNerfEmbedder.forward (mipnerf_embedder.py:34-99) fused: conical-frustum (ray_shape 0) or cylinder
 * (1) gaussians -> integrated positional encoding, concatenated with the positional encoding of the ray's view
 * direction.  out [n_rays*(n_z-1), ld] row-major, ld >= channels (= data['embedded']). */
int xr_mip_encode(const float* rays_o, const float* rays_d, const float* viewdirs, const float* radii /*[n_rays]*/,
                  const float* z_vals /*[n_rays,n_z]*/, uint32_t n_rays, uint32_t n_z, int min_deg, int max_deg,
                  int min_deg_view, int max_deg_view, int append_identity, int ray_shape, float* out, uint32_t ld,
                  void* stream);
/* the same from gaussians that already exist (data['samples'] = (means, covs) [n_rays,n_samples,3]) */
int xr_mip_encode_gaussians(const float* means, const float* covs, const float* viewdirs, uint32_t n_rays,
                            uint32_t n_samples, int min_deg, int max_deg, int min_deg_view, int max_deg_view,
                            int append_identity, float* out, uint32_t ld, void* stream);
/* NerfRender.forward with MipNerfRender's get_weights / get_disp_map (xrnerf/models/renders/nerf_render.py:45-98,
 * mipnerf_render.py:12-33; raw_noise_std = 0): raw [n_rays,n_z-1,4] -> rgb [n_rays,3], distance ('disp')
 * [n_rays], acc [n_rays], weights [n_rays,n_z-1].  density_activation: 0 = softplus, 1 = relu. */
int xr_mip_render_forward(const float* raw, const float* z_vals, const float* rays_d, uint32_t n_rays, uint32_t n_z,
                          float density_bias, float rgb_padding, int white_bkgd, int density_activation, float* rgb,
                          float* distance, float* acc, float* weights, void* stream);
/* dL/draw [n_rays,n_z-1,4] given dL/drgb [n_rays,3] (the reference's losses use the colours only,
 * networks/mipnerf.py:52-60; weights feed the detached resampling); everything else is recomputed from raw */
int xr_mip_render_backward(const float* raw, const float* z_vals, const float* rays_d, const float* grad_rgb,
                           uint32_t n_rays, uint32_t n_z, float density_bias, float rgb_padding, int white_bkgd,
                           int density_activation, float* grad_raw, void* stream);
/* resample_along_rays' new z_vals (mip.py:151-176 with sorted_piecewise_constant_pdf :7-62): max-blurred weights
 * + resample_padding -> pdf -> cdf (fp64 prefix sum like torch's CPU cumsum) -> inverse-cdf sampling at n_z
 * points.  rand [n_rays,n_z] = the torch.rand draws of the randomized branch (NULL = deterministic linspace).
 * z_vals must be sorted per ray (they are edges).  n_z <= 2048. */
int xr_mip_resample(const float* z_vals, const float* weights /*[n_rays,n_z-1]*/, const float* rand,
                    float resample_padding, uint32_t n_rays, uint32_t n_z, float* z_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * KiloNeRF rendering (BASELINE config #5; configs/kilonerf/kilonerf_finetune_*.py, test / validation path)
 *
 * KiloNerfMLP.forward (xrnerf/models/mlps/kilonerf_mlp.py:138-190) in one call: sample -> network assignment with the
 * occupancy / domain filters and the grouping by network (reorder_points_and_dirs, networks/utils/transforms.py:57-151),
 * local coordinates (kilonerf_cuda.global_to_local == transforms.py:35-45), Fourier features
 * (kilonerf_cuda.compute_fourier_features == embedders/kilonerf_fourier_embedder.py:33-52), the per-network MLP
 * (6 x kilonerf_cuda.multimatmul_magma_grouped_static == MultiNetwork.forward, mlps/multi_modules.py:590-668 with
 * late_feed_direction, relu, hidden = direction width = 32, no position re-feed) and the scatter back into
 * raw [n_rays*n_samples, 4] (zeros where no network is evaluated).
 *   samples: either pts [n_rays*n_samples,3] (data['pts']) or, with pts = NULL, o + d * z from rays_o / rays_d [n_rays,3]
 *            and z_vals [n_rays,n_samples] (GetPts, datasets/pipelines/create.py:577-601, never materialised);
 *   viewdirs [n_rays,3]; gmin/gmax/fixed_res/occ_res: HOST arrays of 3 (data['global_domain_min'/'max'], resolution//16,
 *   resolution); occupancy: device bool grid (1 byte per cell, row-major) or NULL; domain_mins/maxs [N,3] device;
 *   params [N, param_stride] device: per network the packed block described in xr_kilo_param_floats' source
 *   (input-major weights = the reference's `multimatmul` layout, kilonerf_mlp.py:104-121);
 *   counts_out (nullable, device [N]) = batch_size_per_network.
 * workspace: xr_kilo_workspace_bytes(n_rays*n_samples, N), 256-byte aligned.  No host synchronisation. */
uint32_t xr_kilo_param_floats(int pos_freqs, int dir_freqs, int n_hidden);
size_t xr_kilo_workspace_bytes(uint64_t n_samples_total, uint32_t num_networks);
int xr_kilo_mlp_forward(const float* pts, const float* rays_o, const float* rays_d, const float* z_vals,
                        const float* viewdirs, uint32_t n_rays, uint32_t n_samples, const float* gmin_host,
                        const float* gmax_host, const int32_t* fixed_res_host, const int32_t* occ_res_host,
                        const uint8_t* occupancy, const float* domain_mins, const float* domain_maxs, const float* params,
                        uint32_t param_stride, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                        float* raw, uint32_t* counts_out, void* workspace, size_t workspace_bytes, void* stream);
/* gradients of the tiny MLPs' parameters for fine-tuning (AddMultiMatMul.backward, mlps/multi_modules.py:215-236, chained
 * through MultiNetwork.forward): same sample arguments as xr_kilo_mlp_forward; draw [n_rays*n_samples,4] = dL/draw (rows no
 * network evaluated are ignored); ACCUMULATES into grad_params [N, param_stride] (block layout of params; caller
 * zero-fills).  n_hidden <= 2, pos_freqs <= 10, dir_freqs <= 4.  workspace: xr_kilo_workspace_bytes. */
int xr_kilo_mlp_backward(const float* pts, const float* rays_o, const float* rays_d, const float* z_vals,
                         const float* viewdirs, uint32_t n_rays, uint32_t n_samples, const float* gmin_host,
                         const float* gmax_host, const int32_t* fixed_res_host, const int32_t* occ_res_host,
                         const uint8_t* occupancy, const float* domain_mins, const float* domain_maxs, const float* params,
                         uint32_t param_stride, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                         const float* draw, float* grad_params, int reuse_assignment, void* workspace, size_t workspace_bytes, void* stream);
/* reuse_assignment != 0: `workspace` still holds what the xr_kilo_mlp_forward call on these very samples left there (network of every
 * sample, the order and the segments): the backward skips the assignment / offsets / scatter launches over all n_rays * n_samples
 * samples -- in a fine-tuning step (8192 x 384 samples of which ~1.5 % meet an occupied cell) that is most of its time.  The caller
 * vouches that nothing else used the workspace in between (the host side keeps a generation count: xrnerf_amd/ops.py).
 *
 * The reference keeps ONE TENSOR PER LAYER of all networks (multi_modules.py:238-340: weight [N, in, out], bias [N, out]); the kernels
 * take one packed block per network.  One launch each way instead of a dozen concatenations / strided copies per step:
 *   xr_kilo_pack_params: tensors_host = 2 (n_hidden + 4) device pointers in block order -- (weight, bias) of pts_linears.*, alpha_linear,
 *     feature_linear, direction_layer, rgb_linear, contiguous fp32 -- -> blocks [N, param_stride];
 *   xr_kilo_unpack_grads: the gradient blocks -> one contiguous tensor per parameter (same order and shapes); clear_blocks != 0 leaves
 *     the blocks zero-filled for the next xr_kilo_mlp_backward (which accumulates). */
int xr_kilo_pack_params(const float* const* tensors_host, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden, float* blocks,
                        uint32_t param_stride, void* stream);
int xr_kilo_unpack_grads(float* blocks, uint32_t param_stride, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                         float* const* grads_host, int clear_blocks, void* stream);
/* One frame (or chunk) of the reference's KiloNeRF test path in one call, for the real-time bench: GetZvals (not
 * randomized; datasets/pipelines/create.py:486-531) + GetPts + KiloNerfMLP.forward + NerfRender.forward.  Same values
 * as xr_mip_zvals -> xr_kilo_mlp_forward -> xr_nerf_render_forward, but no [n_rays, n_samples] tensor other than the
 * per-sample network id is written or read: z is evaluated where it is needed, rows without a network are neither
 * zero-filled nor read back (they contribute exactly nothing to NerfRender's sums).  near / far [n_rays] device.
 * The lattice passes only visit each ray's span of samples that can lie inside the global domain (slab test).
 * workspace: xr_kilo_render_workspace_bytes(n_rays, n_samples, N) (its raw area is only touched where a network runs). */
size_t xr_kilo_render_workspace_bytes(uint32_t n_rays, uint32_t n_samples, uint32_t num_networks);
int xr_kilo_render_rays(const float* rays_o, const float* rays_d, const float* viewdirs, const float* near,
                        const float* far, uint32_t n_rays, uint32_t n_samples, int lindisp, const float* gmin_host,
                        const float* gmax_host, const int32_t* fixed_res_host, const int32_t* occ_res_host,
                        const uint8_t* occupancy, const float* domain_mins, const float* domain_maxs, const float* params,
                        uint32_t param_stride, uint32_t num_networks, int pos_freqs, int dir_freqs, int n_hidden,
                        int white_bkgd, float* rgb, float* disp, float* acc, void* workspace, size_t workspace_bytes,
                        void* stream);
/* NerfRender.forward (xrnerf/models/renders/nerf_render.py:45-98; raw_noise_std = 0, relu density, sigmoid colours,
 * cumprod weights, last interval 1e10): raw [n_rays,n_samples,4], z_vals [n_rays,n_samples] sample positions ->
 * rgb [n_rays,3], disp [n_rays], acc [n_rays], weights [n_rays,n_samples] (nullable).  Inference only. */
int xr_nerf_render_forward(const float* raw, const float* z_vals, const float* rays_d, uint32_t n_rays,
                           uint32_t n_samples, int white_bkgd, float* rgb, float* disp, float* acc, float* weights,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 linear layers of the 8x256 NeRF MLP (xrnerf/models/mlps/nerf_mlp.py:27-94: nn.Linear + F.relu) on the fp32
 * MFMA; row-major tensors, w = nn.Linear.weight [N,K].  K and N must be multiples of 4, pointers 16-byte aligned.
 *   forward:          y [M,N] = act(x [M,K] . w^T + bias)            (bias nullable, relu 0/1)
 *   backward, input:  dx [M,K] = (dy [M,N] where mask_src > 0) . w    (mask_src nullable: the layer's relu output)
 *   backward, weight: dw_partials [splits,N,K] = per-M-range partial sums of (dy masked)^T . x; the caller adds them in
 *                     order (bit-reproducible); splits = xr_linear_backward_splits(M, N, K)
 *   backward, bias:   db_partials [splits,N] = per-M-range column sums of (dy masked); splits = xr_linear_backward_splits(M, 0, 0)
 * Row strides (floats, multiples of 4; 0 = dense): a layer may read its input from, and write its output into, a column range of a wider
 * buffer, and take its output gradient from a column range of the next layer's input gradient -- the skip connection's [x | h]
 * (nerf_mlp.py:70-72) and the view branch's [feature | dir] (:80-85) are then never concatenated or split (xrnerf_amd/vanilla.py). */
int xr_linear_forward(const float* x, uint32_t ldx, const float* w, const float* bias, uint32_t M, uint32_t N, uint32_t K, int relu,
                      float* y, uint32_t ldy, void* stream);
/* w_transposed != 0: the weight is handed over transposed by the caller (w_t [K,N] row-major): served by the forward's split-operand kernel.
 * lddy: row stride of dy and of mask_src */
int xr_linear_backward_input(const float* dy, uint32_t lddy, const float* mask_src, const float* w, int w_transposed, uint32_t M, uint32_t N,
                             uint32_t K, float* dx, void* stream);
/* M ranges of the weight gradient of an N x K layer; N == K == 0: of the bias gradient alone (xr_linear_backward_bias) */
uint32_t xr_linear_backward_splits(uint32_t M, uint32_t N, uint32_t K);
/* db_partials (nullable): weight and bias gradient in one launch: db_partials [splits,N] over the SAME M ranges as dw_partials (taken from
 * the operand panels the product stages anyway; no pass of its own over dy and the mask).  lddy (dy, mask_src) / ldx: row strides.
 * part_stride: floats between two splits' partials (0 = N * K); db_partials == dw_partials + N * K with part_stride = N * K + N puts both
 * sets into one [splits, N * K + N] buffer (one reduction over the splits finishes both gradients). */
int xr_linear_backward_weight(const float* dy, uint32_t lddy, const float* mask_src, const float* x, uint32_t ldx, uint32_t M, uint32_t N,
                              uint32_t K, uint32_t splits, float* dw_partials, float* db_partials, size_t part_stride, void* stream);
int xr_linear_backward_bias(const float* dy, const float* mask_src, uint32_t M, uint32_t N, uint32_t splits,
                            float* db_partials, void* stream);
/* out[j] = sum over b of partials[b * stride + j], j < n, in a FIXED order (every 4th partial per row group, 16 loads in flight, the four
 * group sums as (0 + 1) + (2 + 3)): what finishes the split weight / bias gradients above -- and the fused MLP's gradient partials -- with
 * one launch at memory speed (torch's `partials.sum(0)` took 17 us per 67-MB set of a 256 x 256 layer, this 11).  out is WRITTEN. */
int xr_sum_partials(const float* partials, uint32_t n_partials, size_t stride, uint32_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
