"""Drop-in for the reference's pybind extension module `raymarch_cuda`
(/root/reference/extensions/ngp_raymarch/src/pybind_api.cu:6-17, prototypes include/pybind_api.h:4-95).

Same ten function names, same argument order, same caller-allocates convention, same blocking
behaviour (the reference ends every entry point with cudaDeviceSynchronize()), same hidden per
translation-unit RNG (`static pcg32 rng{9121}`, raymarch_shared.h:38: one generator for
rays_sampler_api, one for generate_grid_samples_nerf_nonuniform_api, each advanced by 2^32 per call) --
so that the reference's own wrappers (`xrnerf/models/samplers/utils/*.py`,
`xrnerf/models/renders/hashnerf_render.py`) run unchanged on top of libxrnerf_mi355.so:

    import sys, xrnerf_amd.raymarch_cuda
    sys.modules['raymarch_cuda'] = xrnerf_amd.raymarch_cuda      # before importing xrnerf.models

The registry classes of this package do NOT go through this module (they call `ops` without the
per-call synchronisation); it exists for the extension-module boundary of SURVEY.md section 8b.
"""
import ctypes as C

import torch

from . import _lib, ops

_calls = {'rays_sampler': 0, 'generate_grid_samples': 0}


def reset_rng():
    """back to the state of a freshly imported reference module"""
    _calls['rays_sampler'] = 0
    _calls['generate_grid_samples'] = 0


def _sync():
    torch.cuda.synchronize()


def _p(t):
    return ops._ptr(t)


def rays_sampler_api(rays_o, rays_d, density_grid_bitfield, metadata, imgs_id, xforms, aabb0, aabb1, near_distance,
                     cone_angle_constant, coords_out, rays_index, rays_numsteps, ray_numstep_counter):
    L = _lib.load()
    n = rays_o.shape[0]
    ws = ops._ws(rays_o.device, L.xr_rays_sampler_workspace_bytes(n, 1), 'k1')
    st, inc = ops.pcg32_host_state(_calls['rays_sampler'])
    _calls['rays_sampler'] += 1
    _lib.check(L.xr_rays_sampler(_p(rays_o), _p(rays_d), _p(density_grid_bitfield), n, float(aabb0), float(aabb1),
                                 float(near_distance), float(cone_angle_constant), coords_out.shape[0], st, inc,
                                 _p(coords_out), _p(rays_index), _p(rays_numsteps), _p(ray_numstep_counter), None, 0, 0, 0, 0, _p(ws),
                                 ws.numel(), ops._stream()), 'rays_sampler_api')
    _sync()


def compacted_coord_api(network_output, coords_in, rays_numsteps, bg_color_in, rgb_activation_i, density_activation_i,
                        aabb0, aabb1, coords_out, rays_numsteps_compacted, compacted_rays_counter,
                        compacted_numstep_counter):
    L = _lib.load()
    n = rays_numsteps.shape[0]
    ws = ops._ws(coords_in.device, L.xr_rays_sampler_workspace_bytes(n, 1), 'k1')
    _lib.check(L.xr_compacted_coord(_p(coords_in), _p(rays_numsteps), n, coords_out.shape[0], _p(coords_out),
                                    _p(rays_numsteps_compacted), _p(compacted_rays_counter),
                                    _p(compacted_numstep_counter), _p(ws), ws.numel(), ops._stream()),
               'compacted_coord_api')
    _sync()


def calc_rgb_forward_api(network_output, coords_in, rays_numsteps, rays_numsteps_compacted, training_background_color,
                         rgb_activation_i, density_activation_i, aabb0, aabb1, rgb_output):
    ops.calc_rgb_forward(network_output, coords_in, rays_numsteps, rays_numsteps_compacted, training_background_color,
                         rgb_activation_i, density_activation_i, out=rgb_output)
    _sync()


def calc_rgb_backward_api(network_output, rays_numsteps_compacted, coords_in, grad_x, rgb_output, density_grid_mean,
                          rgb_activation_i, density_activation_i, aabb0, aabb1, dloss_doutput):
    ops.calc_rgb_backward(network_output, rays_numsteps_compacted, coords_in, grad_x.contiguous(), rgb_output,
                          density_grid_mean, rgb_activation_i, density_activation_i, out=dloss_doutput)
    _sync()


def calc_rgb_influence_api(network_output, coords_in, rays_numsteps, bg_color_cpu, rgb_activation_i,
                           density_activation_i, aabb0, aabb1, rgb_output, alpha_output):
    bg = [float(v) for v in bg_color_cpu.reshape(-1)[:3]]          # a HOST tensor, read on the host (calc_rgb.cu:366,378)
    n = rays_numsteps.shape[0]
    _lib.check(_lib.load().xr_calc_rgb_inference(_p(network_output), _p(coords_in), _p(rays_numsteps), bg[0], bg[1],
                                                 bg[2], n, int(rgb_activation_i), int(density_activation_i),
                                                 _p(rgb_output), _p(alpha_output), ops._stream()),
               'calc_rgb_influence_api')
    _sync()


def generate_grid_samples_nerf_nonuniform_api(density_grid, density_grid_ema_step, n_elements, max_cascade, thresh,
                                              aabb0, aabb1, density_grid_positions_uniform,
                                              density_grid_indices_uniform):
    st, inc = ops.pcg32_host_state(_calls['generate_grid_samples'])
    _calls['generate_grid_samples'] += 1                    # advances on every call, also for n_elements == 0 (:84)
    if n_elements > 0:
        _lib.check(_lib.load().xr_generate_grid_samples(_p(density_grid), int(density_grid_ema_step), int(n_elements),
                                                        int(max_cascade) + 1, float(thresh), float(aabb0),
                                                        float(aabb1), st, inc, _p(density_grid_positions_uniform), 3, 1,
                                                        _p(density_grid_indices_uniform), ops._stream()),
                   'generate_grid_samples_nerf_nonuniform_api')
    _sync()


def mark_untrained_density_grid_api(focal_lengths, transforms, n_elements, n_images, img_resolution0, img_resolution1,
                                    density_grid):
    ops.mark_untrained_density_grid(focal_lengths, transforms, int(n_elements), (img_resolution0, img_resolution1),
                                    grid=density_grid)
    _sync()


def splat_grid_samples_nerf_max_nearest_neighbor_api(mlp_out, density_grid_indices, padded_output_width,
                                                     n_density_grid_samples, density_grid_tmp):
    ops.splat_grid_samples(mlp_out, density_grid_indices, int(padded_output_width), int(n_density_grid_samples),
                           density_grid_tmp)
    _sync()


def ema_grid_samples_nerf_api(density_grid_tmp, n_elements, decay, density_grid):
    ops.ema_grid_samples(density_grid_tmp, int(n_elements), float(decay), density_grid)
    _sync()


def update_bitfield_api(density_grid, density_grid_mean, density_grid_bitfield):
    ops.update_bitfield(density_grid, density_grid_mean, density_grid_bitfield)
    _sync()


__all__ = ['generate_grid_samples_nerf_nonuniform_api', 'mark_untrained_density_grid_api',
           'splat_grid_samples_nerf_max_nearest_neighbor_api', 'ema_grid_samples_nerf_api', 'update_bitfield_api',
           'rays_sampler_api', 'compacted_coord_api', 'calc_rgb_forward_api', 'calc_rgb_backward_api',
           'calc_rgb_influence_api', 'reset_rng']
