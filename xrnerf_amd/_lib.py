"""ctypes binding of libxrnerf_mi355.so (the C-ABI of include/xrnerf_mi355.h).

There is NO fallback: if the library is missing or a call fails this raises. The product path never
imports anything from oracle/.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# XRNERF_LIB: another build of the same library (kernel A/Bs that need a compile-time constant changed: tools/build_variant.sh)
LIB_PATH = os.environ.get('XRNERF_LIB') or os.path.join(HERE, 'libxrnerf_mi355.so')

_vp, _u32, _i32, _f, _u64, _sz, _d = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_uint64, C.c_size_t, C.c_double

# name -> (restype, argtypes); mirrors include/xrnerf_mi355.h one to one
class AdamFuse(C.Structure):
    """xr_adam_fuse (include/xrnerf_mi355.h)"""
    _fields_ = [('param', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('ema', C.c_void_p), ('step', C.c_int),
                ('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('weight_decay', C.c_float),
                ('ema_momentum', C.c_float), ('grad_scale', C.c_float), ('n', C.c_uint64)]


class Window(C.Structure):
    """xr_ngp_window"""
    _fields_ = [(k, C.c_void_p) for k in ('rays_o', 'rays_d', 'target', 'alpha', 'bg', 'img_ids', 'rays_index', 'rays_numsteps',
                                          'numsteps_clipped')] + \
               [('ray_stride', C.c_uint32), ('coords', C.c_void_p), ('coords_stride', C.c_size_t), ('xyz_planes', C.c_void_p),
                ('plane_stride', C.c_uint32), ('counter2', C.c_void_p), ('n_valid', C.c_void_p)]


class StepSet(C.Structure):
    """xr_ngp_step_set"""
    _fields_ = [(k, C.c_void_p) for k in ('enc_t', 'raw', 'draw', 'denc_t', 'rgb_out', 'zero_block')] + [('zero_floats', C.c_size_t)] + \
               [(k, C.c_void_p) for k in ('grad_w_density', 'grad_w_color', 'loss_mse', 'live_seg_count', 'grad_table')]


WINDOW = 16                      # XR_NGP_WINDOW
LIVE_SEGMENT_ROWS = 1024         # XR_LIVE_SEGMENT_ROWS

EX_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
EX_SCATTER_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
EX_FINISH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class GradExchange(C.Structure):
    """xr_grad_exchange"""
    _fields_ = [('all_reduce', EX_ALL_REDUCE), ('reduce_scatter', EX_SCATTER_GATHER), ('all_gather', EX_SCATTER_GATHER),
                ('finish', EX_FINISH), ('ctx', C.c_void_p), ('world_size', C.c_int), ('rank', C.c_int)]


class LoopDesc(C.Structure):
    """xr_ngp_loop_desc"""
    _fields_ = [('table', C.c_void_p), ('w_density', C.c_void_p), ('w_color', C.c_void_p), ('n_hidden_density', C.c_int),
                ('n_hidden_color', C.c_int), ('pad_value', C.c_float), ('mlp_mode', C.c_int), ('n_levels', C.c_int),
                ('scale_host', C.c_void_p), ('resolution_host', C.c_void_p), ('offset_host', C.c_void_p),
                ('adam_table', AdamFuse), ('adam_w_density', AdamFuse), ('adam_w_color', AdamFuse),
                ('density_grid_mean', C.c_void_p), ('rgb_activation', C.c_int), ('density_activation', C.c_int),
                ('huber_delta', C.c_float), ('loss_scale', C.c_float), ('n_rows', C.c_uint32), ('ld', C.c_uint32),
                ('window', Window), ('step', StepSet * 2),
                ('ws_mlp_bwd', C.c_void_p), ('ws_mlp_bwd_bytes', C.c_size_t), ('ws_scatter', C.c_void_p), ('ws_scatter_bytes', C.c_size_t),
                ('stream', C.c_void_p),
                ('exchange', C.POINTER(GradExchange)), ('dp_mode', C.c_int), ('split_level', C.c_int),
                ('shard_grad', C.c_void_p), ('table_padded', C.c_void_p), ('shard_floats', C.c_uint64)]


class LoopState(C.Structure):
    """xr_ngp_loop_state"""
    _fields_ = [('iter', C.c_uint64), ('step_turn', C.c_uint32), ('adam_step', C.c_int32), ('last_step_set', C.c_uint32)]


SIGNATURES = {
    'xr_last_error': (C.c_char_p, []),
    'xr_version': (_i32, []),
    'xr_event_record': (_i32, [_vp, _vp]),
    'xr_timing_event_create': (_vp, []),
    'xr_timing_event_destroy': (_i32, [_vp]),
    'xr_timing_event_elapsed_ms': (_i32, [_vp, _vp, _vp]),
    'xr_pcg32_host_state': (None, [_u64, _u64, _vp, _vp]),
    'xr_rays_sampler': (_i32, [_vp, _vp, _vp, _u32, _f, _f, _f, _f, _u32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _sz, _vp]),
    'xr_compacted_coord': (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xr_rays_sampler_workspace_bytes': (_sz, [_u32, _u32]),
    'xr_rays_sampler_series': (_i32, [_vp, _vp, _u32, _vp, _u32, _u32, _f, _f, _f, _f, _u32, _u64, _u64, _vp, _sz, _vp, _vp, _vp, _vp, _u32, _vp,
                                      _sz, _vp]),
    'xr_clip_numsteps': (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    'xr_make_batch_series': (_i32, [_vp, _vp, _u32, _u32, _u32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xr_ngp_window_march': (_i32, [_vp, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _u64, _u64, _vp, _f, _f, _f, _f, _u32, _u64, _u32, _vp, _sz,
                                   _vp, _vp]),
    'xr_render_slice_select': (_i32, [_vp, _vp, _u32, _u32, _u32, _f, _vp, _vp, _vp, _vp]),
    'xr_render_slice_composite': (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _i32, _i32, _vp, _vp, _vp]),
    'xr_calc_rgb_forward': (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _i32, _i32, _vp, _vp]),
    'xr_calc_rgb_backward': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _i32, _i32, _vp, _vp]),
    'xr_hashgrid_bwd_adam': (_i32, [_vp, _u32, _vp, _u32, _u32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'xr_train_loss_scalars': (_i32, [_vp, _vp, _vp, _u32, _f, _f, _vp, _vp]),
    'xr_composite_train': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _i32, _i32, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'xr_calc_rgb_inference': (_i32, [_vp, _vp, _vp, _f, _f, _f, _u32, _i32, _i32, _vp, _vp, _vp]),
    'xr_generate_grid_samples': (_i32, [_vp, _u32, _u32, _u32, _f, _f, _f, _u64, _u64, _vp, _u32, _u32, _vp, _vp]),
    'xr_mark_untrained_density_grid': (_i32, [_vp, _vp, _u32, _u32, _i32, _i32, _vp, _vp]),
    'xr_splat_grid_samples': (_i32, [_vp, _vp, _u32, _u32, _vp, _vp]),
    'xr_ema_grid_samples': (_i32, [_vp, _u32, _f, _vp, _vp]),
    'xr_update_bitfield_workspace_bytes': (_sz, []),
    'xr_update_bitfield': (_i32, [_vp, _vp, _vp, _vp, _sz, _vp]),
    'xr_ema_update_bitfield': (_i32, [_vp, _u32, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xr_bitfield_from_mean': (_i32, [_vp, _vp, _vp, _vp]),
    'xr_hashgrid_meta': (None, [_i32, _i32, _i32, _d, _vp, _vp, _vp]),
    'xr_hashgrid_fwd': (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _u32, _vp]),
    'xr_hashgrid_bwd_workspace_bytes': (_sz, [_u32, _i32, _vp, _vp]),
    'xr_hashgrid_bwd': (_i32, [_vp, _u32, _vp, _u32, _u32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    'xr_set_helper_stream': (_i32, [_vp, _vp, _vp]),
    'xr_set_mlp_range_word': (_i32, [_vp]),
    'xr_sh4': (_i32, [_vp, _u32, _u32, _vp, _vp]),
    'xr_nerf_mlp_fwd': (_i32, [_i32, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _i32, _i32, _f, _vp, _vp]),
    'xr_nerf_mlp_bwd_workspace_bytes': (_sz, [_u32, _i32, _i32]),
    'xr_ngp_train_step': (_i32, [_vp, _vp, _vp, _i32, _i32, _f, _i32, _i32, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp,
                                 _vp, _i32, _i32, _f, _f, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp, _sz,
                                 _vp, _sz, _i32, _vp, _u32, _vp, _vp, _vp, C.c_char_p, _vp, _vp, _vp]),
    'xr_ngp_loop_run': (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, C.c_char_p, _vp, _vp]),
    'xr_rccl_unique_id': (_i32, [C.c_char_p, _vp]),
    'xr_rccl_create': (_vp, [C.c_char_p, _vp, _i32, _i32]),
    'xr_rccl_destroy': (_i32, [_vp]),
    'xr_rccl_exchange': (_i32, [_vp, _vp]),
    'xr_rccl_exposed_ms': (_i32, [_vp, _i32, _vp, _vp, _vp]),
    'xr_nerf_mlp_bwd': (_i32, [_i32, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _i32, _i32, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    'xr_live_rows': (_i32, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _i32, _vp]),
    'xr_nerf_mlp_bwd_list_slots': (_i32, [_vp, _sz, _u32, _vp, _vp, _vp]),
    'xr_mlp_fwd': (_i32, [_vp, C.c_long, C.c_long, _i32, _f, _u32, _vp, _i32, _vp, _vp]),
    'xr_mlp_bwd_workspace_bytes': (_sz, [_i32]),
    'xr_mlp_bwd': (_i32, [_vp, C.c_long, C.c_long, _i32, _f, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xr_gen_rays': (_i32, [_vp, _i32, _i32, _f, _f, _f, _f, _i32, _i32, _vp, _vp, _vp]),
    'xr_huber_loss_grad': (_i32, [_vp, _vp, _vp, _u32, _f, _f, _vp, _vp, _vp]),
    'xr_adam_step_multi': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f, _f, _f, _f, _f, _f, _f, _vp]),
    'xr_scale_multi': (_i32, [_i32, _vp, _vp, _vp, _f, _vp]),
    'xr_mip_zvals': (_i32, [_vp, _vp, _u32, _u32, _i32, _vp, _vp, _vp]),
    'xr_mip_encode_channels': (_u32, [_i32, _i32, _i32, _i32, _i32]),
    'xr_mip_encode': (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _u32, _vp]),
    'xr_mip_encode_gaussians': (_i32, [_vp, _vp, _vp, _u32, _u32, _i32, _i32, _i32, _i32, _i32, _vp, _u32, _vp]),
    'xr_mip_render_forward': (_i32, [_vp, _vp, _vp, _u32, _u32, _f, _f, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'xr_mip_render_backward': (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _f, _f, _i32, _i32, _vp, _vp]),
    'xr_mip_resample': (_i32, [_vp, _vp, _vp, _f, _u32, _u32, _vp, _vp]),
    'xr_kilo_param_floats': (_u32, [_i32, _i32, _i32]),
    'xr_kilo_workspace_bytes': (_sz, [_u64, _u32]),
    'xr_kilo_mlp_forward': (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32,
                                   _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    'xr_kilo_mlp_backward': (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32,
                                    _i32, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    'xr_kilo_pack_params': (_i32, [_vp, _u32, _i32, _i32, _i32, _vp, _u32, _vp]),
    'xr_kilo_unpack_grads': (_i32, [_vp, _u32, _u32, _i32, _i32, _i32, _vp, _i32, _vp]),
    'xr_nerf_render_forward': (_i32, [_vp, _vp, _vp, _u32, _u32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'xr_nerf_density_splat': (_i32, [_i32, _vp, _u32, _u32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'xr_linear_forward': (_i32, [_vp, _u32, _vp, _vp, _u32, _u32, _u32, _i32, _vp, _u32, _vp]),
    'xr_linear_backward_input': (_i32, [_vp, _u32, _vp, _vp, _i32, _u32, _u32, _u32, _vp, _vp]),
    'xr_linear_backward_splits': (_u32, [_u32, _u32, _u32]),
    'xr_linear_backward_bias': (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    'xr_sum_partials': (_i32, [_vp, _u32, _sz, _u32, _vp, _vp]),
    'xr_linear_backward_weight': (_i32, [_vp, _u32, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _sz, _vp]),
    'xr_kilo_render_workspace_bytes': (_sz, [_u32, _u32, _u32]),
    'xr_kilo_render_rays': (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32,
                                   _u32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


class XrError(RuntimeError):
    pass


def load():
    """dlopen the library (building it first when the sources are newer and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build

    def is_stale():
        if not os.environ.get('XRNERF_LIB') and os.path.exists(LIB_PATH) and os.path.isdir(_build.CSRC):
            # the binary beside the sources must be the build OF those sources (build.py stamps every build with both hashes)
            i = _build.info()
            return i.get('stamp') == 'missing' or not i.get('binary_is_the_stamped_build') or i.get('sources_match_the_stamped_build') is False
        return False

    stale = is_stale()
    if not os.path.exists(LIB_PATH) or stale:
        # one builder per tree: the ranks of a multi-GPU job start together (torch.distributed.run), and N compilers writing the same
        # objects would hand every rank a broken library.  The others wait for the lock and find the finished build.
        import fcntl
        import warnings

        def rebuild():
            try:
                _build.build(force=stale)
            except Exception as e:  # noqa: BLE001
                if not os.path.exists(LIB_PATH):
                    raise XrError('libxrnerf_mi355.so is missing and could not be built: %s. '
                                  'Run `python -m xrnerf_amd.build` (needs hipcc).' % e)
                # a usable binary whose stamp does not match (copied without its .stamp, header or recipe edited, no hipcc here): load
                # it and say so (bench.py's `library_build` reports the mismatch either way; a header / library mismatch fails below)
                warnings.warn('libxrnerf_mi355.so is not the stamped build of the sources beside it and could not be rebuilt (%s): '
                              'loading it as it is' % e)
        try:
            lock = open(LIB_PATH + '.lock', 'w')
        except OSError:                    # read-only install: nobody can build here, so nobody races either
            lock = None
        if lock is None:
            rebuild()
        else:
            with lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    stale = is_stale()
                    if not os.path.exists(LIB_PATH) or stale:
                        rebuild()
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        raise XrError('%s failed (%d): %s' % (what, rc, load().xr_last_error().decode()))
