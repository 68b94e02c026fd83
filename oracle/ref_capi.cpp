// TEST INFRASTRUCTURE ONLY (oracle/). Never imported by the product path.
//
// C-ABI wrappers around the reference's OWN `*_api` entry points
// (/root/reference/extensions/ngp_raymarch/include/pybind_api.h:4-95), compiled for the
// CPU through oracle/shim. Each reference .cu file is one
// translation unit here (selected with -DTU_<name>) because every TU owns a private
// `static pcg32 rng{9121}` (raymarch_shared.h:38) exactly like the CUDA build; the
// wrapper includes the .cu so that it can reset that TU-local generator for tests.
// No reference source is copied: the .cu files are #included from where they lie.
#include <cstdint>
#include <vector>

#if defined(TU_K1)
#include "ray_sampler.cu"
#elif defined(TU_K2)
#include "compacted_coord.cu"
#elif defined(TU_K345)
#include "calc_rgb.cu"
#elif defined(TU_K6)
#include "generate_grid_samples_nerf_nonuniform.cu"
#elif defined(TU_K7)
#include "mark_untrained_density_grid.cu"
#elif defined(TU_K8)
#include "splat_grid_samples_nerf_max_nearest_neighbor.cu"
#elif defined(TU_K9)
#include "ema_grid_samples_nerf.cu"
#elif defined(TU_K1011)
#include "update_bitfield.cu"   // the shadow copy (serial mean), see build.py
#elif defined(TU_GLOBALS)
#include "cuda_runtime.h"
thread_local xr_shim_dim3 threadIdx, blockIdx, blockDim, gridDim;
#endif

#ifndef TU_GLOBALS
using torch::Tensor;
typedef at::ScalarType ST;
static inline Tensor T_(const void* p, std::vector<int64_t> s, ST t = ST::Float) {
    return Tensor((void*)p, s, t);
}
// put the TU-local generator into the state it has after `ncalls` launches
static inline void rng_set(uint64_t ncalls) {
    rng = pcg32{9121};
    for (uint64_t c = 0; c < ncalls; ++c) rng.advance();
}
#endif

extern "C" {

#if defined(TU_K1)
void ref_rays_sampler(const float* rays_o, const float* rays_d, const uint8_t* bitfield,
                      const float* metadata, const int32_t* img_ids, const float* xforms,
                      int n_rays, int n_img, float aabb0, float aabb1, float near_distance,
                      float cone_angle, int max_samples, uint64_t rng_calls,
                      float* coords_out, int32_t* rays_index, int32_t* rays_numsteps,
                      int32_t* counter2) {
    rng_set(rng_calls);
    Tensor ro = T_(rays_o, {n_rays, 3}), rd = T_(rays_d, {n_rays, 3});
    Tensor bf = T_(bitfield, {128 * 128 * 128}, ST::Byte), md = T_(metadata, {n_img, 11});
    Tensor ii = T_(img_ids, {n_rays, 1}, ST::Int), xf = T_(xforms, {n_img, 4, 3});
    Tensor co = T_(coords_out, {max_samples, 7}), ri = T_(rays_index, {n_rays, 1}, ST::Int);
    Tensor rn = T_(rays_numsteps, {n_rays, 2}, ST::Int), cn = T_(counter2, {2}, ST::Int);
    rays_sampler_api(ro, rd, bf, md, ii, xf, aabb0, aabb1, near_distance, cone_angle, co, ri, rn, cn);
}
// PCG32 known-answer helpers (pcg32.h:39-201), for the oracle's own generator check
void ref_pcg32_probe(uint64_t seed, uint64_t advance_by, uint64_t* state, uint64_t* inc,
                     uint32_t* u5, float* f3) {
    pcg32 r{seed};
    r.advance((int64_t)advance_by);
    *state = r.state; *inc = r.inc;
    pcg32 a = r; for (int i = 0; i < 5; ++i) u5[i] = a.next_uint();
    pcg32 b = r; for (int i = 0; i < 3; ++i) f3[i] = b.next_float();
}
#elif defined(TU_K2)
void ref_compacted_coord(const float* network_output, const float* coords_in,
                         const int32_t* rays_numsteps, int n_rays, int n_samples,
                         int compacted_elements, int rgb_act, int density_act, float aabb0,
                         float aabb1, float* coords_out, int32_t* numsteps_compacted,
                         int32_t* rays_counter, int32_t* numstep_counter) {
    float bg[3] = {1, 1, 1};
    Tensor no = T_(network_output, {n_samples, 4}), ci = T_(coords_in, {n_samples, 7});
    Tensor rn = T_(rays_numsteps, {n_rays, 2}, ST::Int), b = T_(bg, {3});
    Tensor co = T_(coords_out, {compacted_elements, 7});
    Tensor nc = T_(numsteps_compacted, {n_rays, 2}, ST::Int);
    Tensor rc = T_(rays_counter, {1}, ST::Int), sc = T_(numstep_counter, {1}, ST::Int);
    compacted_coord_api(no, ci, rn, b, rgb_act, density_act, aabb0, aabb1, co, nc, rc, sc);
}
#elif defined(TU_K345)
void ref_calc_rgb_forward(const float* network_output, const float* coords, const int32_t* numsteps,
                          const int32_t* numsteps_compacted, const float* bg, int n_rays,
                          int n_samples, int rgb_act, int density_act, float aabb0, float aabb1,
                          float* rgb_out) {
    Tensor no = T_(network_output, {n_samples, 4}), ci = T_(coords, {n_samples, 7});
    Tensor rn = T_(numsteps, {n_rays, 2}, ST::Int), rc = T_(numsteps_compacted, {n_rays, 2}, ST::Int);
    Tensor b = T_(bg, {n_rays, 3}), o = T_(rgb_out, {n_rays, 3});
    calc_rgb_forward_api(no, ci, rn, rc, b, rgb_act, density_act, aabb0, aabb1, o);
}
void ref_calc_rgb_backward(const float* network_output, const int32_t* numsteps_compacted,
                           const float* coords, const float* grad_rgb, const float* rgb_out,
                           const float* density_grid_mean, int n_rays, int n_samples, int rgb_act,
                           int density_act, float aabb0, float aabb1, float* dloss_doutput) {
    Tensor no = T_(network_output, {n_samples, 4}), ci = T_(coords, {n_samples, 7});
    Tensor rc = T_(numsteps_compacted, {n_rays, 2}, ST::Int);
    Tensor g = T_(grad_rgb, {n_rays, 3}), o = T_(rgb_out, {n_rays, 3});
    Tensor m = T_(density_grid_mean, {1}), d = T_(dloss_doutput, {n_samples, 4});
    calc_rgb_backward_api(no, rc, ci, g, o, m, rgb_act, density_act, aabb0, aabb1, d);
}
void ref_calc_rgb_inference(const float* network_output, const float* coords, const int32_t* numsteps,
                            const float* bg3, int n_rays, int n_samples, int rgb_act,
                            int density_act, float aabb0, float aabb1, float* rgb_out,
                            float* alpha_out) {
    Tensor no = T_(network_output, {n_samples, 4}), ci = T_(coords, {n_samples, 7});
    Tensor rn = T_(numsteps, {n_rays, 2}, ST::Int), b = T_(bg3, {3});
    Tensor o = T_(rgb_out, {n_rays, 3}), a = T_(alpha_out, {n_rays, 1});
    calc_rgb_influence_api(no, ci, rn, b, rgb_act, density_act, aabb0, aabb1, o, a);
}
#elif defined(TU_K6)
void ref_generate_grid_samples(const float* density_grid, int ema_step, int n_elements,
                               int max_cascade, float thresh, float aabb0, float aabb1,
                               uint64_t rng_calls, float* positions, int32_t* indices) {
    rng_set(rng_calls);
    Tensor g = T_(density_grid, {128 * 128 * 128 * 8});
    Tensor p = T_(positions, {n_elements, 3}), i = T_(indices, {n_elements}, ST::Int);
    generate_grid_samples_nerf_nonuniform_api(g, ema_step, n_elements, max_cascade, thresh, aabb0,
                                              aabb1, p, i);
}
#elif defined(TU_K7)
void ref_mark_untrained(const float* focal, const float* xforms, int n_elements, int n_img,
                        int res0, int res1, float* density_grid) {
    Tensor f = T_(focal, {n_img, 2}), x = T_(xforms, {n_img, 4, 3}), g = T_(density_grid, {n_elements});
    mark_untrained_density_grid_api(f, x, n_elements, n_img, res0, res1, g);
}
#elif defined(TU_K8)
void ref_splat(const float* mlp_out, const int32_t* indices, int padded_width, int n_samples,
               float* density_grid_tmp) {
    Tensor m = T_(mlp_out, {n_samples, padded_width}), i = T_(indices, {n_samples}, ST::Int);
    Tensor g = T_(density_grid_tmp, {128 * 128 * 128 * 8});
    splat_grid_samples_nerf_max_nearest_neighbor_api(m, i, padded_width, n_samples, g);
}
#elif defined(TU_K9)
void ref_ema(const float* density_grid_tmp, int n_elements, float decay, float* density_grid) {
    Tensor t = T_(density_grid_tmp, {n_elements}), g = T_(density_grid, {n_elements});
    ema_grid_samples_nerf_api(t, n_elements, decay, g);
}
#elif defined(TU_K1011)
void ref_update_bitfield(const float* density_grid, float* density_grid_mean, uint8_t* bitfield) {
    Tensor g = T_(density_grid, {128 * 128 * 128 * 8}), m = T_(density_grid_mean, {16384});
    Tensor b = T_(bitfield, {128 * 128 * 128}, ST::Byte);
    update_bitfield_api(g, m, b);
}
// K11 alone (grid_to_bitfield + 7 max-pools, update_bitfield.cu:24-71,103-112) with the mean
// SUPPLIED by the caller, so bit-exactness of the bitfield can be tested independently of the
// float summation order of K10.
void ref_bitfield_given_mean(const float* density_grid, const float* mean1, uint8_t* bitfield) {
    const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE();
    linear_kernel(grid_to_bitfield, 0, (cudaStream_t)0, n_elements / 8 * NERF_CASCADES(),
                  density_grid, bitfield, mean1);
    for (uint32_t level = 1; level < NERF_CASCADES(); ++level)
        linear_kernel(bitfield_max_pool, 0, (cudaStream_t)0, n_elements / 64,
                      bitfield + grid_mip_offset(level - 1) / 8, bitfield + grid_mip_offset(level) / 8);
}
#endif

}  // extern "C"
