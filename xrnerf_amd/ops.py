"""Tensor-level front-end of the C-ABI: torch is used only for device memory and streams.

Every function takes contiguous CUDA(ROCm) tensors, enqueues on torch's current stream and does
NOT synchronise.  There is no CPU path: a non-CUDA tensor is an error.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

G3 = 128 ** 3
N_GRID = 8 * G3

_workspaces = {}


class KernelTimer:
    """Optional per-entry-point device timing with events recorded on the launch stream (torch's
    current stream IS the stream every kernel of this library is enqueued on).  bench.py uses it to
    get the dominant kernel's average duration live inside the timed region."""

    def __init__(self, only=None, train_only=False):
        self.events = {}
        self.only = only            # record only these entry points (None = all)
        self.train_only = train_only   # record only the training step's launches (row count on the device), not the
                                       # occupancy-grid density queries / render launches of the same entry points

    def span(self, name, units=0, train=True):
        if self.only is not None and name not in self.only:
            return _NOSPAN
        if self.train_only and not train:
            return _NOSPAN
        return _Span(self, name, units)

    # the entry points xr_ngp_train_step runs (it can bracket ONE of them with a pair of events): a timer that asks for
    # nothing else inside the step leaves the native path usable
    STEP_STAGES = ('xr_hashgrid_fwd', 'xr_nerf_mlp_fwd', 'xr_composite_train', 'xr_live_rows', 'xr_nerf_mlp_bwd', 'xr_hashgrid_bwd')

    def native_stage(self):
        """-> (ok, stage): ok if the native step can serve this timer; stage = the one step stage to bracket (or None)"""
        if self.only is None:
            return False, None
        inside = [k for k in self.only if k in self.STEP_STAGES]
        return len(inside) <= 1, (inside[0] if inside else None)

    def summary(self):
        """name -> (launches, total_ms, total_units); call after torch.cuda.synchronize()"""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v), sum(u for _, _, u in v))
                for k, v in self.events.items()}


class _Span:
    def __init__(self, timer, name, units):
        self.timer, self.name, self.units = timer, name, units

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.timer.events.setdefault(self.name, []).append((self.a, self.b, self.units))


class _CEvent:
    """a timing event of the library (xr_timing_event_*): what xr_ngp_train_step / xr_ngp_loop_run record around a stage or in front of an
    iteration"""

    def __init__(self):
        self.h = _lib.load().xr_timing_event_create()
        if not self.h:
            raise _lib.XrError('cannot create a timing event')

    def elapsed_time(self, other):
        ms = C.c_float()
        _lib.check(_lib.load().xr_timing_event_elapsed_ms(self.h, other.h, C.byref(ms)), 'xr_timing_event_elapsed_ms')
        return float(ms.value)

    def __del__(self):
        try:
            _lib.load().xr_timing_event_destroy(self.h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER = None
_NOSPAN = _NoSpan()


def _span(name, units=0, train=True):
    return TIMER.span(name, units, train) if TIMER is not None else _NOSPAN


_PRECISION = os.environ.get('XRNERF_MLP_PRECISION', 'f32')
if _PRECISION not in ('f32', 'f16'):
    raise ValueError("XRNERF_MLP_PRECISION must be 'f32' or 'f16' (got %r)" % _PRECISION)


# forward of the 'f32' mode (the `arithmetic` argument of xr_nerf_mlp_fwd): 'f16x2' = XR_MLP_F16X2 (the default: fp32 operands as two fp16 parts, three fp16 MFMAs per product
# block, fp32 accumulate: within a few ulps of fp32 at half the matrix instructions of the 3-way split), 'bf16x3' =
# XR_MLP_BF16X3 (three bf16 parts, six MFMAs: fp32-rounding accuracy), 'mfma' = XR_MLP_F32 (v_mfma_f32_32x32x2_f32).
# Depths other than (1, 2) run the streamed kernel (f16x2 arithmetic) whatever this says, except (1,1), (2,2), (2,3) under 'mfma'.
_F32_FORWARD = os.environ.get('XRNERF_F32_FORWARD', 'f16x2')          # (set_f32_forward; switches.py)
if _F32_FORWARD not in ('f16x2', 'bf16x3', 'mfma'):
    raise ValueError("XRNERF_F32_FORWARD must be 'f16x2', 'bf16x3' or 'mfma' (got %r)" % _F32_FORWARD)


def f32_forward():
    return _F32_FORWARD


def set_f32_forward(kind):
    global _F32_FORWARD
    if kind not in ('f16x2', 'bf16x3', 'mfma'):
        raise ValueError("the fp32 forward is 'f16x2', 'bf16x3' or 'mfma'")
    _F32_FORWARD = kind


def _mlp_mode(nhd=1, nhc=2):
    """xr_ngp_train_step's mlp_mode for the current settings"""
    if nhd == 1 and nhc == 2:
        if _PRECISION == 'f16':
            return 1
        if _F32_FORWARD == 'bf16x3':
            return 2
    return 3 if _F32_FORWARD == 'f16x2' else 0


def precision():
    """arithmetic mode of the fused MLP: 'f32' (parity mode, default: v_mfma_f32_32x32x2_f32, exact fp32) or 'f16'
    (the reference's own precision -- tiny-cuda-nn computes FullyFusedMLP in fp16 with fp32 accumulation:
    v_mfma_f32_32x32x16_f16, fp32 parameters / gradients in memory, fp32 outputs)"""
    return _PRECISION


def set_precision(p):
    global _PRECISION
    if p not in ('f32', 'f16'):
        raise ValueError("precision must be 'f32' or 'f16'")
    _PRECISION = p


def _stream():
    # raw hipStream_t of torch's current stream; torch.cuda.current_stream() costs ~7 us of Python per call and
    # a training step makes ~20 of them
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _on_device(t):
    """the one place that decides whether a tensor may be handed to the library (tests/hip_emu swaps it to run the
    kernels' host build on CPU tensors)"""
    return t.is_cuda


def _ptr(t):
    if t is None:
        return None
    if not _on_device(t):
        raise _lib.XrError('xrnerf_amd ops need ROCm device tensors (got a %s tensor): there is no CPU fallback' % t.device)
    if not t.is_contiguous():
        raise _lib.XrError('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())


def _ws(device, nbytes, tag):
    key = (str(device), tag)
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if tag == 'mlpbwd':
            # holds the live-row count block, whose words 1-2 are RUNNING totals (k_live_fill adds to them): they start at zero.
            # (uint32: a reader -- bench.py prices the backward kernels from them -- clears them per measurement window)
            w.zero_()
        _workspaces[key] = w
    return w


def _buf(device, shape, tag):
    """grow-only fp32 scratch of the given shape (a view of a persistent buffer per tag): frame-sized intermediates whose size
    changes from frame to frame (13-15 M samples: 1.7 GB of encoded features) would otherwise send the caching allocator to
    hipMalloc / hipFree -- device-synchronising calls of ~10 ms -- on most frames"""
    n = 1
    for v in shape:
        n *= int(v)
    key = (str(device), tag)
    have = _workspaces.get(key)
    need = 4 * max(n, 1) + 256
    # grow with 25 % headroom: frames of one sequence differ by a few per cent in their sample counts
    w = _ws(device, need if (have is not None and have.numel() >= need) else need + need // 4, tag)
    off = (-w.data_ptr()) % 16
    return w[off:off + 4 * n].view(torch.float32).view(*shape)


_helpers = {}
_helper_current = {}            # host thread -> the device whose helper the library's (per-thread) slot holds right now


def _ensure_helper(device):
    """hand the library this thread's helper stream (xr_set_helper_stream): one stream + fork / join events per device and host thread,
    created HERE -- the library itself creates nothing.  Without it the scatter's side work runs in order on the compute stream.
    The library's slot is one per host thread: a thread that moves to another device hands over that device's stream again."""
    import threading
    if device.type != 'cuda':
        return
    tid = threading.get_ident()
    key = (str(device), tid)
    if _helper_current.get(tid) == key:
        return
    h = _helpers.get(key)
    if h is None:
        with torch.cuda.device(device):
            st = torch.cuda.Stream(device=device)
            evs = (torch.cuda.Event(), torch.cuda.Event())
            for e in evs:
                e.record(st)                           # (torch creates the underlying event at its first record, on the current device)
        h = _helpers[key] = (st, evs)
    st, evs = h
    _lib.check(_lib.load().xr_set_helper_stream(C.c_void_p(st.cuda_stream), C.c_void_p(evs[0].cuda_event), C.c_void_p(evs[1].cuda_event)),
               'xr_set_helper_stream')
    _helper_current[tid] = key


# ---- range of the default MLP arithmetic (XR_MLP_F16X2 saturates its operands at +-65504 and counts the waves that saw one out of
# range in the caller's word: include/xrnerf_mi355.h).  One uint32 word per device; the library's slot is per host thread.
_range_words = {}
_range_current = {}
_range_off = set()              # host threads whose launches run untracked right now (mlp_range_tracking)


def mlp_range_word(device):
    """this device's range word (a [1] int32 tensor, allocated once), handed to the library for the calling thread -- unless the thread
    has tracking switched off (mlp_range_tracking), in which case the library is told `none`: the forwards then run without the max
    chain behind the count (4 us of a 32-us launch), the saturation itself stays"""
    import threading
    if device.type != 'cuda':
        return None
    w = _range_words.get(str(device))
    if w is None:
        w = _range_words[str(device)] = torch.zeros((1,), dtype=torch.int32, device=device)
    tid = threading.get_ident()
    want = None if tid in _range_off else str(device)
    if _range_current.get(tid, 0) != want:
        _lib.check(_lib.load().xr_set_mlp_range_word(C.c_void_p(w.data_ptr()) if want is not None else None), 'xr_set_mlp_range_word')
        _range_current[tid] = want
    return w


def mlp_range_tracking(device, on):
    """Switch the range count of the calling thread's XR_MLP_F16X2 forwards on (default) or off.  The trainer runs the iterations
    between two grid refreshes untracked (train._NativeLoop) and everything else -- the refresh iterations with their 2^20-point
    density queries, frames, direct calls -- tracked: one iteration in 16 looks, which is where growing weights or features show up"""
    import threading
    tid = threading.get_ident()
    if on:
        _range_off.discard(tid)
    else:
        _range_off.add(tid)
    mlp_range_word(device)


def mlp_range_events(device, reset=False):
    """waves of the default forward that met an operand above fp16's range since the last reset (a host read-back: synchronises)"""
    w = mlp_range_word(device)
    if w is None:
        return 0
    v = int(w.item())
    if reset and v:
        w.zero_()
    return v


def pcg32_host_state(ncalls, seed=9121):
    s, i = C.c_uint64(), C.c_uint64()
    _lib.load().xr_pcg32_host_state(seed, ncalls, C.byref(s), C.byref(i))
    return s.value, i.value


# ---------------------------------------------------------------- K1 / K2
def rays_sampler(rays_o, rays_d, bitfield, aabb, near_distance, cone_angle, max_samples, rng_calls,
                 coords_out=None, ws_tag='k1', small_out=None, xyz_out=None, rng_chunk=0, rng_ray0=0, wide=False):
    """returns coords_out [max_samples,7], rays_index [n,1], rays_numsteps [n,2], counter [2] (device).
    `small_out` = (rays_index, numsteps, counter) caller-owned buffers (persistent double buffers of the trainer).
    `xyz_out` = [3, >= max_samples] float32: the sample positions once more as three planes (hashgrid_fwd's fast input).
    `rng_chunk` > 0: the jitter of ceil(n / rng_chunk) consecutive launches over rng_chunk rays each (call indices rng_calls,
    rng_calls + 1, ...) in this one launch; `rng_ray0`: the launch's rays are rays rng_ray0.. of that series' frame.
    `wide`: nothing else runs on the device (XR_K1_WIDE): up to 32 768 rays march with 8 lanes per ray, same samples."""
    L = _lib.load()
    n = rays_o.shape[0]
    dev = rays_o.device
    if coords_out is None:
        coords_out = torch.empty((max_samples, 7), dtype=torch.float32, device=dev)
    if small_out is not None:
        rays_index, numsteps, counter = small_out
    else:
        rays_index = torch.empty((n, 1), dtype=torch.int32, device=dev)
        numsteps = torch.empty((n, 2), dtype=torch.int32, device=dev)
        counter = torch.empty((2,), dtype=torch.int32, device=dev)
    nb = L.xr_rays_sampler_workspace_bytes(n, 1)
    ws = _ws(dev, nb, ws_tag)
    st, inc = pcg32_host_state(rng_calls)
    with _span('xr_rays_sampler', n):
        _lib.check(L.xr_rays_sampler(_ptr(rays_o), _ptr(rays_d), _ptr(bitfield), n, aabb[0], aabb[1], near_distance,
                                      cone_angle, max_samples, st, inc, _ptr(coords_out), _ptr(rays_index),
                                      _ptr(numsteps), _ptr(counter), _ptr(xyz_out), xyz_out.shape[1] if xyz_out is not None else 0,
                                      int(rng_chunk), int(rng_ray0), 1 if wide else 0, _ptr(ws), ws.numel(), _stream()), 'xr_rays_sampler')
    return coords_out, rays_index, numsteps, counter


def compacted_coord(coords_in, numsteps, max_compacted, coords_out=None):
    L = _lib.load()
    n = numsteps.shape[0]
    dev = coords_in.device
    if coords_out is None:
        coords_out = torch.zeros((max_compacted, 7), dtype=torch.float32, device=dev)
    nc = torch.empty((n, 2), dtype=torch.int32, device=dev)
    rc = torch.empty((1,), dtype=torch.int32, device=dev)
    sc = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _ws(dev, L.xr_rays_sampler_workspace_bytes(n, 1), 'k1')
    _lib.check(L.xr_compacted_coord(_ptr(coords_in), _ptr(numsteps), n, max_compacted, _ptr(coords_out), _ptr(nc),
                                    _ptr(rc), _ptr(sc), _ptr(ws), ws.numel(), _stream()), 'xr_compacted_coord')
    return coords_out, nc, rc, sc


# ---------------------------------------------------------------- K3 / K4 / K5
def calc_rgb_forward(raw, coords, numsteps, numsteps_c, bg, rgb_act, density_act, out=None):
    L = _lib.load()
    n = numsteps.shape[0]
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=raw.device)
    with _span('xr_calc_rgb_forward', 0):
        _lib.check(L.xr_calc_rgb_forward(_ptr(raw), _ptr(coords), _ptr(numsteps), _ptr(numsteps_c), _ptr(bg), n,
                                         int(rgb_act), int(density_act), _ptr(out), _stream()), 'xr_calc_rgb_forward')
    return out


def calc_rgb_backward(raw, numsteps_c, coords, grad_rgb, rgb_out, density_grid_mean, rgb_act, density_act, out=None):
    L = _lib.load()
    n = numsteps_c.shape[0]
    if out is None:
        out = torch.zeros_like(raw)
    with _span('xr_calc_rgb_backward', 0):
        _lib.check(L.xr_calc_rgb_backward(_ptr(raw), _ptr(numsteps_c), _ptr(coords), _ptr(grad_rgb), _ptr(rgb_out),
                                          _ptr(density_grid_mean), n, int(rgb_act), int(density_act), _ptr(out),
                                          _stream()), 'xr_calc_rgb_backward')
    return out


def composite_train(raw, coords, numsteps, numsteps_c, bg, target, alpha, density_grid_mean, rgb_act, density_act, loss_mse, draw,
                    delta=0.1, scale=5.0, rgb=None, live_seg=None):
    """K3 + scale * Huber (+ masked MSE) + K4 in one launch -> rgb [n,3]; `loss_mse` [2] and `draw` [S,4] must be zero-filled
    (loss terms are added, rows behind the last sample are not written).
    live_seg (int32 [live_segments(S)], zero-filled by the caller): the launch also counts the non-zero rows of `draw` per
    1024-row segment into it; follow with live_rows(..., seg_counts=live_seg).
    loss_mse=None: the wave-per-ray kernel (rgb / draw differ from the 16-lane kernels' like two fp32 summation orders); the loss
    scalars then come from train_loss_scalars(rgb, ...)."""
    n = numsteps.shape[0]
    if rgb is None:
        rgb = torch.empty((n, 3), dtype=torch.float32, device=raw.device)
    p_seg = _ptr(live_seg)
    with _span('xr_composite_train', 0):
        _lib.check(_lib.load().xr_composite_train(_ptr(raw), _ptr(coords), _ptr(numsteps), _ptr(numsteps_c), _ptr(bg), _ptr(target),
                                                   _ptr(alpha), _ptr(density_grid_mean), n, int(rgb_act), int(density_act),
                                                   float(delta), float(scale), _ptr(rgb), _ptr(loss_mse), _ptr(draw), p_seg, _stream()),
                   'xr_composite_train')
    return rgb


class MarchWindow:
    """Caller-owned buffers of a marched refresh window (xr_ngp_window, include/xrnerf_mi355.h): XR_NGP_WINDOW chunks at fixed strides.
    Iteration `it` lives in chunk it % WINDOW.  One allocation per array; `chunk(c)` hands out the views a single march / step takes."""

    N = _lib.WINDOW

    def __init__(self, device, ray_stride, coords_stride, planes=True):
        W = self.N
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=device)
        self.device, self.ray_stride, self.coords_stride = device, int(ray_stride), int(coords_stride)
        self.rays_o, self.rays_d, self.target, self.bg = f(W, ray_stride, 3), f(W, ray_stride, 3), f(W, ray_stride, 3), f(W, ray_stride, 3)
        self.alpha, self.img_ids = f(W, ray_stride, 1), i(W, ray_stride, 1)
        self.rays_index, self.numsteps, self.clipped = i(W, ray_stride, 1), i(W, ray_stride, 2), i(W, ray_stride, 2)
        self.coords = f(W, coords_stride, 7)
        self.xyz = f(W, 3, coords_stride) if planes else None
        self.counter2, self.n_valid = i(W, 2), i(W, 2)
        self.pinned = torch.zeros((W, 2), dtype=torch.int32).pin_memory() if device.type == 'cuda' else None
        c = _lib.Window()
        vp = lambda t: t.data_ptr() if t is not None else None
        c.rays_o, c.rays_d, c.target, c.alpha, c.bg, c.img_ids = vp(self.rays_o), vp(self.rays_d), vp(self.target), vp(self.alpha), vp(self.bg), vp(self.img_ids)
        c.rays_index, c.rays_numsteps, c.numsteps_clipped = vp(self.rays_index), vp(self.numsteps), vp(self.clipped)
        c.ray_stride, c.coords, c.coords_stride = self.ray_stride, vp(self.coords), self.coords_stride
        c.xyz_planes, c.plane_stride = vp(self.xyz), (self.coords_stride if planes else 0)
        c.counter2, c.n_valid = vp(self.counter2), vp(self.n_valid)
        self.c = c

    def batch(self, c, n):
        """the batch dict of chunk c (views; what HashBatchSample + RandomBGColor hand to train_step)"""
        return {'rays_o': self.rays_o[c, :n], 'rays_d': self.rays_d[c, :n], 'target_s': self.target[c, :n], 'alpha': self.alpha[c, :n],
                'img_ids': self.img_ids[c, :n], 'bg_color': self.bg[c, :n]}

    def batch_out(self, c):
        """chunk c's buffers in the form make_batch(out=...) takes"""
        return {'rays_o': self.rays_o[c], 'rays_d': self.rays_d[c], 'target_s': self.target[c], 'alpha': self.alpha[c],
                'img_ids': self.img_ids[c], 'bg_color': self.bg[c]}


def ngp_window_march(win, first_chunk, n_chunks, batches_ready, n, rows_table, cur_ray, batch_call_index, bitfield, aabb, near_distance,
                     cone_angle, max_samples, k1_call_index, max_compacted, batch_seed=20220901, ws_tag='k1_window'):
    """the batches and marches of window chunks [first_chunk, first_chunk + n_chunks) as one series of launches on the current stream
    (xr_ngp_window_march): batch assembly for all but the first `batches_ready` chunks, K1, K2's clip, counters to win.pinned.
    -> the table cursor behind the last batch drawn"""
    L = _lib.load()
    ws = _ws(win.device, L.xr_rays_sampler_workspace_bytes(n, n_chunks), ws_tag)
    try:
        native = L.xr_ngp_window_march
    except AttributeError:
        native = None            # the kernels' host build (tests/hip_emu) has no native executors: the same series calls from here
    if native is not None:
        cur = C.c_uint64(int(cur_ray))
        with _span('xr_rays_sampler', n * n_chunks):
            _lib.check(native(C.byref(win.c), first_chunk, n_chunks, batches_ready, n, _ptr(rows_table),
                              rows_table.shape[0] if rows_table is not None else 0, C.byref(cur), batch_seed, batch_call_index,
                              _ptr(bitfield), aabb[0], aabb[1], near_distance, cone_angle, max_samples, k1_call_index, max_compacted,
                              _ptr(ws), ws.numel(), C.c_void_p(win.pinned.data_ptr()) if win.pinned is not None else None, _stream()),
                       'xr_ngp_window_march')
        return int(cur.value)
    c0, c1, cur = first_chunk, first_chunk + n_chunks, int(cur_ray)
    if batches_ready < n_chunks:
        row0 = []
        for _ in range(batches_ready, n_chunks):
            if cur + n > rows_table.shape[0]:
                cur = 0
            row0.append(cur)
            cur += n
        b0 = c0 + batches_ready
        st, inc = pcg32_host_state(batch_call_index, batch_seed)
        _lib.check(L.xr_make_batch_series(_ptr(rows_table), (C.c_uint64 * len(row0))(*row0), n, len(row0), win.ray_stride, st, inc, _ptr(win.rays_o[b0:c1]),
                                          _ptr(win.rays_d[b0:c1]), _ptr(win.target[b0:c1]), _ptr(win.alpha[b0:c1]), _ptr(win.bg[b0:c1]),
                                          _ptr(win.img_ids[b0:c1]), _stream()), 'xr_make_batch_series')
    st, inc = pcg32_host_state(k1_call_index)
    _lib.check(L.xr_rays_sampler_series(_ptr(win.rays_o[c0:c1]), _ptr(win.rays_d[c0:c1]), win.ray_stride, _ptr(bitfield), n, n_chunks, aabb[0], aabb[1],
                                        near_distance, cone_angle, max_samples, st, inc, _ptr(win.coords[c0:c1]), win.coords_stride,
                                        _ptr(win.rays_index[c0:c1]), _ptr(win.numsteps[c0:c1]), _ptr(win.counter2[c0:c1]),
                                        _ptr(win.xyz[c0:c1]) if win.xyz is not None else None, win.coords_stride if win.xyz is not None else 0,
                                        _ptr(ws), ws.numel(), _stream()), 'xr_rays_sampler_series')
    _lib.check(L.xr_clip_numsteps(_ptr(win.numsteps[c0:c1]), _ptr(win.counter2[c0:c1]), n, n_chunks, win.ray_stride, max_compacted,
                                         _ptr(win.clipped[c0:c1]), _ptr(win.n_valid[c0:c1]), _stream()), 'xr_clip_numsteps')
    return cur


class TrainStepBuffers:
    """caller-owned buffers of xr_ngp_train_step for `n_rows` sample rows and up to `ray_cap` rays"""

    def __init__(self, device, n_rows, ray_cap, table_floats, wd_floats, wc_floats, meta, table_grad_storage=None):
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        self.n_rows, self.ray_cap, self.ld = n_rows, ray_cap, (n_rows + 63) // 64 * 64
        self.enc_t, self.denc_t = f(meta.n_output_dims, self.ld), f(meta.n_output_dims, self.ld)
        self.raw, self.draw, self.rgb = f(n_rows, 4), f(n_rows, 4), f(ray_cap, 3)
        # MLP gradients, the (loss, mse) scalars and the compositor's live-row counts in one block; the step writes the first two and
        # hands the counts back zeroed (no fill on the step's stream)
        n_seg = live_segments(n_rows)
        self.zero_block = torch.zeros((wd_floats + wc_floats + 4 + n_seg,), dtype=torch.float32, device=device)     # (the counts start at zero)
        self.live_seg = self.zero_block[wd_floats + wc_floats + 4:]
        self.g_wd, self.g_wc = self.zero_block[:wd_floats], self.zero_block[wd_floats:wd_floats + wc_floats]
        self.g_mlp = self.zero_block[:wd_floats + wc_floats]
        self.loss_mse = self.zero_block[wd_floats + wc_floats:wd_floats + wc_floats + 2]
        # `table_grad_storage`: -> (padded buffer, leading [table_floats] view) when the gradient collective wants padded storage
        # (dist.Zero1GradSync.pad_grad); the step writes the view
        self.g_table = table_grad_storage(device)[1] if table_grad_storage is not None else f(table_floats)


def ngp_train_step(table, wd, wc, nhd, nhc, pad_value, meta, coords, n_dev, numsteps, numsteps_c, bg, target, alpha,
                   density_grid_mean, rgb_act, density_act, bufs, huber_delta=0.1, loss_scale=5.0, scatter_level0=0, xyz=None,
                   adam=None, mlp_adam=None):
    """the device work of one HashNerfNetwork training step as one native call (xr_ngp_train_step): encode -> MLP -> K3 +
    Huber + K4 -> MLP backward -> table scatter into `bufs` (TrainStepBuffers).  Returns rgb [n_rays,3] (a view of bufs.rgb).
    scatter_level0 > 0 (data parallel): only hash levels [scatter_level0, n_levels) are scattered; the caller finishes with
    hashgrid_bwd(..., live=bufs.live, levels=(0, scatter_level0)) after handing the finer slice to its collective.
    adam (ops.adam_fuse of the table, single GPU): the scatter applies the optimiser's update to the table itself; bufs.g_table
    is not written.  mlp_adam = (adam_fuse of wd, adam_fuse of wc): their update runs behind the reduction of their gradients.

    The ~45 pointer / size arguments are validated and converted once per set of buffers (keyed by the data pointers of the
    tensors that alternate between iterations): at 0.45 ms per iteration the interpreter time of this call -- 30 pointer checks,
    four workspace queries -- was what the GPU waited for (tools/hosttime2.py)."""
    global LIVE_STATS
    L = _lib.load()
    n_rays = numsteps.shape[0]
    mode = _mlp_mode(nhd, nhc)
    mlp_range_word(coords.device)
    key = (table.data_ptr(), wd.data_ptr(), wc.data_ptr(), coords.data_ptr(), coords.shape[0], n_dev.data_ptr() if n_dev is not None else 0,
           numsteps.data_ptr(), numsteps_c.data_ptr(), bg.data_ptr(), target.data_ptr(), alpha.data_ptr(), density_grid_mean.data_ptr(),
           xyz.data_ptr() if xyz is not None else 0, xyz.shape[1] if xyz is not None else 0, nhd, nhc, pad_value, mode, id(meta),
           int(rgb_act), int(density_act), huber_delta, loss_scale, int(scatter_level0))
    cache = bufs.__dict__.setdefault('_step_args', {})
    ent = cache.get(key)
    if ent is None:
        n_rows = bufs.n_rows
        if coords.shape[0] < n_rows or coords.shape[1] != 7 or not coords.is_contiguous():
            raise _lib.XrError('coords must be contiguous [>= n_rows, 7] rows')
        s, r, o = meta._args()
        _ensure_helper(coords.device)
        _ws(coords.device, L.xr_nerf_mlp_bwd_workspace_bytes(n_rows, nhd, nhc), 'mlpbwd')
        ws_sc = _ws(coords.device, L.xr_hashgrid_bwd_workspace_bytes(n_rows, meta.n_levels, r, o), 'hgb')
        # the backward's (count, running live total, running valid total) block sits in the MLP workspace on this path
        ws_mlp, live_list, _, live_stats = _list_slots(coords.device, n_rows, nhd, nhc)
        live = (live_list, live_stats) if os.environ.get('XR_MLP_LIVE') != '0' else None     # (the native step reads the same switch)
        head = (_ptr(table), _ptr(wd), _ptr(wc), nhd, nhc, pad_value, mode, meta.n_levels, s, r, o,
                _ptr(coords), n_rows, _ptr(n_dev), _ptr(numsteps), _ptr(numsteps_c))
        mid = (_ptr(bg), _ptr(target), _ptr(alpha),
               _ptr(density_grid_mean), int(rgb_act), int(density_act), float(huber_delta), float(loss_scale),
               _ptr(bufs.enc_t), bufs.ld, _ptr(bufs.raw), _ptr(bufs.draw), _ptr(bufs.denc_t), _ptr(bufs.rgb),
               _ptr(bufs.zero_block), bufs.zero_block.numel(), _ptr(bufs.g_wd), _ptr(bufs.g_wc), _ptr(bufs.loss_mse), _ptr(bufs.live_seg),
               _ptr(bufs.g_table), bufs.g_table.numel(), 0 if n_dev is not None else 1,
               _ptr(ws_mlp), ws_mlp.numel(), _ptr(ws_sc), ws_sc.numel(), int(scatter_level0),
               _ptr(xyz), xyz.shape[1] if xyz is not None else 0)
        if len(cache) > 16:
            cache.clear()
        ent = cache[key] = (head, mid, live, live_stats, (ws_mlp, ws_sc, meta, table, wd, wc))     # (keeps what the pointers name alive)
    head, mid, live, LIVE_STATS = ent[0], ent[1], ent[2], ent[3]
    bufs.live = live
    stage, ev = None, (None, None)
    if TIMER is not None:
        ok, stage = TIMER.native_stage()
        if not ok:
            raise _lib.XrError('this KernelTimer needs the per-entry-point launch sequence (XRNERF_STEP=py)')
        if stage is not None:
            ev = (_CEvent(), _CEvent())
            TIMER.events.setdefault(stage, []).append((ev[0], ev[1], 0))
    rc = L.xr_ngp_train_step(*head, n_rays, *mid, C.byref(adam) if adam is not None else None,
                             C.byref(mlp_adam[0]) if mlp_adam else None, C.byref(mlp_adam[1]) if mlp_adam else None,
                             stage.encode() if stage else None, ev[0].h if stage else None, ev[1].h if stage else None, _stream())
    if rc != 0:
        _lib.check(rc, 'xr_ngp_train_step')
    return bufs.rgb[:n_rays]


def record_event(cevent):
    """record a timing event (_CEvent) on the current stream"""
    _lib.check(_lib.load().xr_event_record(cevent.h, _stream()), 'xr_event_record')


def calc_rgb_inference(raw, coords, numsteps, bg3, rgb_act, density_act):
    L = _lib.load()
    n = numsteps.shape[0]
    rgb = torch.empty((n, 3), dtype=torch.float32, device=raw.device)
    alpha = torch.empty((n, 1), dtype=torch.float32, device=raw.device)
    if raw.shape[0] == 0 or coords.shape[0] == 0:
        # a chunk of rays that all miss the occupied cells (the sky rows of a frame marched in chunk = 4096 pieces): every
        # per-ray count is 0 and nothing is read, but the entry point wants non-null buffers
        raw = torch.zeros((1, 4), dtype=torch.float32, device=numsteps.device)
        coords = torch.zeros((1, 7), dtype=torch.float32, device=numsteps.device)
    _lib.check(L.xr_calc_rgb_inference(_ptr(raw), _ptr(coords), _ptr(numsteps), float(bg3[0]), float(bg3[1]),
                                       float(bg3[2]), n, int(rgb_act), int(density_act), _ptr(rgb), _ptr(alpha),
                                       _stream()), 'xr_calc_rgb_inference')
    return rgb, alpha


def render_slice_select(numsteps, T, s0, s1, eps, rows, ray_off, count):
    _lib.check(_lib.load().xr_render_slice_select(_ptr(numsteps), _ptr(T), numsteps.shape[0], s0, s1, eps, _ptr(rows),
                                                  _ptr(ray_off), _ptr(count), _stream()), 'xr_render_slice_select')


def render_slice_composite(raw_s, coords, numsteps, ray_off, s0, s1, rgb_act, density_act, T, rgb_acc):
    _lib.check(_lib.load().xr_render_slice_composite(_ptr(raw_s), _ptr(coords), _ptr(numsteps), _ptr(ray_off),
                                                     numsteps.shape[0], s0, s1, int(rgb_act), int(density_act), _ptr(T),
                                                     _ptr(rgb_acc), _stream()), 'xr_render_slice_composite')


# ---------------------------------------------------------------- K6 .. K11
def generate_grid_samples(grid, ema_step, n_elements, n_cascades, thresh, aabb, rng_calls, planes_out=None, idx_out=None, offset=0):
    """K6.  Default: -> (positions [n,3], indices [n]).  `planes_out` [3, m] + `idx_out` [m]: the n points go to columns
    [offset, offset + n) of the planes (and of idx_out) instead -- both calls of a grid refresh fill one buffer, no concatenation,
    and the density query reads the planes with coalesced loads."""
    L = _lib.load()
    dev = grid.device
    st, inc = pcg32_host_state(rng_calls)
    if planes_out is not None:
        if n_elements:
            _ptr(planes_out); _ptr(idx_out)
            _lib.check(L.xr_generate_grid_samples(_ptr(grid), ema_step, n_elements, n_cascades, thresh, aabb[0], aabb[1], st, inc,
                                                   C.c_void_p(planes_out.data_ptr() + 4 * offset), 1, planes_out.stride(0),
                                                   C.c_void_p(idx_out.data_ptr() + 4 * offset), _stream()), 'xr_generate_grid_samples')
        return planes_out[:, offset:offset + n_elements], idx_out[offset:offset + n_elements]
    pos = torch.empty((n_elements, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((n_elements,), dtype=torch.int32, device=dev)
    _lib.check(L.xr_generate_grid_samples(_ptr(grid), ema_step, n_elements, n_cascades, thresh, aabb[0], aabb[1],
                                          st, inc, _ptr(pos), 3, 1, _ptr(idx), _stream()), 'xr_generate_grid_samples')
    return pos, idx


def mark_untrained_density_grid(focal, xforms, n_elements, resolutions, grid=None):
    L = _lib.load()
    if grid is None:
        grid = torch.empty((n_elements,), dtype=torch.float32, device=focal.device)
    _lib.check(L.xr_mark_untrained_density_grid(_ptr(focal), _ptr(xforms), n_elements, xforms.shape[0],
                                                int(resolutions[0]), int(resolutions[1]), _ptr(grid), _stream()),
               'xr_mark_untrained_density_grid')
    return grid


def splat_grid_samples(mlp_out, indices, padded_width, n_samples, grid_tmp):
    """mlp_out may be a strided [n,1] view: `padded_width` is its row stride in floats."""
    if not _on_device(mlp_out):
        raise _lib.XrError('xrnerf_amd ops need ROCm device tensors: there is no CPU fallback')
    _lib.check(_lib.load().xr_splat_grid_samples(C.c_void_p(mlp_out.data_ptr()), _ptr(indices), padded_width, n_samples,
                                                 _ptr(grid_tmp), _stream()), 'xr_splat_grid_samples')
    return grid_tmp


def ema_grid_samples(grid_tmp, n_elements, decay, grid):
    _lib.check(_lib.load().xr_ema_grid_samples(_ptr(grid_tmp), n_elements, decay, _ptr(grid), _stream()),
               'xr_ema_grid_samples')
    return grid


def update_bitfield(grid, mean, bitfield):
    L = _lib.load()
    ws = _ws(grid.device, L.xr_update_bitfield_workspace_bytes(), 'k10')
    _lib.check(L.xr_update_bitfield(_ptr(grid), _ptr(mean), _ptr(bitfield), _ptr(ws), ws.numel(), _stream()),
               'xr_update_bitfield')
    return bitfield, mean


def ema_update_bitfield(grid_tmp, n_elements, decay, grid, mean, bitfield):
    """K9 + K10 + K11 of one refresh in three launches (xr_ema_update_bitfield) == ema_grid_samples then update_bitfield, bit for bit"""
    L = _lib.load()
    ws = _ws(grid.device, L.xr_update_bitfield_workspace_bytes(), 'k10')
    _lib.check(L.xr_ema_update_bitfield(_ptr(grid_tmp), n_elements, decay, _ptr(grid), _ptr(mean), _ptr(bitfield), _ptr(ws), ws.numel(), _stream()),
               'xr_ema_update_bitfield')
    return grid


def bitfield_from_mean(grid, mean, bitfield):
    _lib.check(_lib.load().xr_bitfield_from_mean(_ptr(grid), _ptr(mean), _ptr(bitfield), _stream()),
               'xr_bitfield_from_mean')
    return bitfield


# ---------------------------------------------------------------- hash grid / SH / MLP
class GridMeta:
    """Host-side level geometry (shared verbatim with the oracle)."""

    def __init__(self, n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=None):
        if per_level_scale is None:
            per_level_scale = float(np.exp2(np.log2(2048 * 1 / 16) / (16 - 1)))  # hashnerf_mlp.py:17-20
        self.n_levels = int(n_levels)
        self.log2_hashmap_size = int(log2_hashmap_size)
        self.n_features = 2
        self.scale = np.zeros(n_levels, np.float32)
        self.resolution = np.zeros(n_levels, np.uint32)
        self.offset = np.zeros(n_levels + 1, np.uint32)
        _lib.load().xr_hashgrid_meta(n_levels, log2_hashmap_size, base_resolution, per_level_scale,
                                     self.scale.ctypes.data, self.resolution.ctypes.data, self.offset.ctypes.data)
        self.n_params = int(self.offset[-1]) * 2
        self.n_output_dims = 2 * self.n_levels

    def _args(self):
        return self.scale.ctypes.data, self.resolution.ctypes.data, self.offset.ctypes.data


def _pos_view(x):
    """(tensor, element stride) for an [n,>=3] fp32 position source: a contiguous [n,3] tensor, or
    the leading 3 columns of contiguous [n,7] coordinate rows (consumed in place)."""
    if x.dim() != 2 or x.shape[1] < 3 or x.dtype != torch.float32:
        raise _lib.XrError('positions must be a 2-D float32 tensor with >= 3 columns')
    if x.stride(1) != 1:
        raise _lib.XrError('positions must have unit column stride')
    if not _on_device(x):
        raise _lib.XrError('xrnerf_amd ops need ROCm device tensors (got a %s tensor): there is no CPU fallback' % x.device)
    return x, int(x.stride(0))


def clip_numsteps(numsteps, counter, max_compacted, out=None):
    """K2 without the copy (K1's output kept in place): clipped per-ray counts + the device-side count of valid
    rows n_valid [2] = (total, total)  (the C entry point can also split the count over row chunks).
    `out` = caller-owned (clipped [n,2] int32, n_valid [2] int32) buffers."""
    n = numsteps.shape[0]
    if out is not None:
        out, n_valid = out
    else:
        out = torch.empty_like(numsteps)
        n_valid = torch.empty((2,), dtype=torch.int32, device=numsteps.device)
    _lib.check(_lib.load().xr_clip_numsteps(_ptr(numsteps), _ptr(counter), n, 1, n, max_compacted, _ptr(out), _ptr(n_valid),
                                            _stream()), 'xr_clip_numsteps')
    return out, n_valid


def hashgrid_fwd(table, x, meta, enc_t=None, ld=None, n_dev=None, rows=None, row0=0, count=None, levels=None):
    """x: [n,3] (or a column slice of [n,7] rows), or positions as three planes [3, m] (m >= n: structure of arrays, three
    coalesced loads per sample) -> enc_t [2L, ld] feature-major.
    n_dev: optional device int32[1]; only min(n, n_dev) rows are touched (no host read-back needed).
    levels=(l0, l1): only that level range is evaluated (rows 2*l0 .. 2*l1 of enc_t; measurement / partial updates)."""
    L = _lib.load()
    if x.dim() == 2 and x.shape[0] == 3 and x.shape[1] != 3 and x.stride(1) == 1:      # planes [3, m]
        if x.dtype != torch.float32 or not _on_device(x):
            raise _lib.XrError('positions must be float32 device tensors')
        xs, xcs, n_x = 1, int(x.stride(0)), x.shape[1]
    else:
        x, xs = _pos_view(x)
        xcs, n_x = 1, x.shape[0]
    n = n_x if rows is None else rows.shape[0]
    if ld is None:
        ld = (n + 63) // 64 * 64
    if enc_t is None:
        enc_t = torch.empty((meta.n_output_dims, ld), dtype=torch.float32, device=x.device)
    s, r, o = meta._args()
    l0, l1 = (0, meta.n_levels) if levels is None else levels
    if not 0 <= l0 < l1 <= meta.n_levels:
        raise _lib.XrError('bad level range %r' % (levels,))
    if count is not None:          # a row chunk [row0, row0+count) of the sample buffer, written to the same columns
        n = count
    xp = x.data_ptr() + 4 * xs * row0
    ep = enc_t.data_ptr() + 4 * (row0 + 2 * l0 * ld)
    with _span('xr_hashgrid_fwd', 0 if n_dev is not None else n, train=n_dev is not None):
        _ptr(enc_t)
        _lib.check(L.xr_hashgrid_fwd(_ptr(table), C.c_void_p(xp), xs, xcs, n, _ptr(n_dev), _ptr(rows), l1 - l0, s + 4 * l0, r + 4 * l0,
                                      o + 4 * l0, C.c_void_p(ep), ld, _stream()), 'xr_hashgrid_fwd')
    return enc_t


def hashgrid_bwd(x, denc_t, meta, grad_table, n_dev=None, row0=0, count=None, levels=None, use_workspace=True, live=None,
                 overwrite=False):
    """scatter dL/denc into dL/dtable.  `levels=(l0, l1)` restricts the launch to that level range (the level
    metadata arrays are passed from l0 on; table offsets are absolute, so `grad_table` stays the full table):
    the data-parallel trainer scatters the fine half first and reduces it across ranks under the coarse half.
    `live=(rows, n_live)` (ops.live_rows): only the listed rows are scattered.
    `overwrite=True` (XR_SCATTER_OVERWRITE): the levels' slices of grad_table are written instead of added to."""
    L = _lib.load()
    _ensure_helper(denc_t.device)
    x, xs = _pos_view(x)
    n = x.shape[0] if count is None else count
    s, r, o = meta._args()
    l0, l1 = (0, meta.n_levels) if levels is None else levels
    if not 0 <= l0 < l1 <= meta.n_levels:
        raise _lib.XrError('bad level range %r' % (levels,))
    ld = denc_t.shape[1]
    _ptr(denc_t)
    ws = _ws(x.device, L.xr_hashgrid_bwd_workspace_bytes(n, l1 - l0, r + 4 * l0, o + 4 * l0), 'hgb') if use_workspace else None
    rows = None
    if live is not None:
        if row0:
            raise _lib.XrError('a live-row list addresses rows from 0')
        rows, n_dev = live
    with _span('xr_hashgrid_bwd', 0 if n_dev is not None else n, train=n_dev is not None):
        _lib.check(L.xr_hashgrid_bwd(C.c_void_p(x.data_ptr() + 4 * xs * row0), xs,
                                      C.c_void_p(denc_t.data_ptr() + 4 * (row0 + 2 * l0 * ld)), ld, n, _ptr(n_dev), _ptr(rows),
                                      l1 - l0, s + 4 * l0, r + 4 * l0, o + 4 * l0, _ptr(grad_table),
                                      _ptr(ws), ws.numel() if ws is not None else 0, 1 if overwrite else 0, _stream()),
                   'xr_hashgrid_bwd')
    return grad_table


def adam_fuse(param, m, v, ema, step, lr, beta1, beta2, eps, weight_decay, ema_momentum=0.0, grad_scale=1.0):
    """-> xr_adam_fuse for hashgrid_bwd_adam / ngp_train_step(adam=): whole tensors + this update's constants (keeps the tensors alive)"""
    a = _lib.AdamFuse(param.data_ptr(), m.data_ptr(), v.data_ptr(), ema.data_ptr() if ema is not None else None, int(step), float(lr),
                      float(beta1), float(beta2), float(eps), float(weight_decay), float(ema_momentum), float(grad_scale), param.numel())
    a._keep = (param, m, v, ema)
    for t in a._keep:
        if t is not None:
            _ptr(t)                       # device / contiguity check
    return a


def hashgrid_bwd_adam_supported(n, meta):
    """every level of `meta` has a non-atomic scatter path at a capacity of n rows (what the fused update needs)"""
    s, r, o = meta._args()
    return _lib.load().xr_hashgrid_bwd_adam(None, 0, None, 0, int(n), None, None, meta.n_levels, s, r, o, None, 0, None, None) == 0       # (adam = NULL: the dry run)


def hashgrid_bwd_adam(x, denc_t, meta, adam, n_dev=None, live=None, count=None):
    """the table scatter with the optimiser's update in place of the gradient write (xr_hashgrid_bwd_adam): the tensors named
    by `adam` (ops.adam_fuse) are updated exactly as hashgrid_bwd(overwrite=True) + adam_step_multi would; no gradient is produced"""
    L = _lib.load()
    _ensure_helper(denc_t.device)
    x, xs = _pos_view(x)
    n = x.shape[0] if count is None else count
    s, r, o = meta._args()
    ws = _ws(x.device, L.xr_hashgrid_bwd_workspace_bytes(n, meta.n_levels, r, o), 'hgb')
    rows = None
    if live is not None:
        rows, n_dev = live
    with _span('xr_hashgrid_bwd', 0 if n_dev is not None else n, train=n_dev is not None):
        _lib.check(L.xr_hashgrid_bwd_adam(C.c_void_p(x.data_ptr()), xs, _ptr(denc_t), denc_t.shape[1], n, _ptr(n_dev), _ptr(rows),
                                          meta.n_levels, s, r, o, _ptr(ws), ws.numel(), C.byref(adam), _stream()), 'xr_hashgrid_bwd_adam')


def sh4(dirs):
    dirs, ds = _pos_view(dirs)
    n = dirs.shape[0]
    out = torch.empty((n, 16), dtype=torch.float32, device=dirs.device)
    _lib.check(_lib.load().xr_sh4(C.c_void_p(dirs.data_ptr()), ds, n, _ptr(out), _stream()), 'xr_sh4')
    return out


# ---- deeper tiny MLPs than the fused kernels are built for --------------------------------------------------------
# tiny-cuda-nn's key for the depth is `n_hidden_layers` (default 5); the reference config writes `num_layers`
# (configs/instant_ngp/nerf_blender_local01.py:106-124, passed unchanged by xrnerf/models/mlps/hashnerf_mlp.py:39-45), so a
# checkpoint of the real reference may hold 5-hidden-layer nets (19 456 floats each; SURVEY.md section 2c, XRNERF_TCNN_STRICT_DEFAULTS).
# The (1, 2) kernels keep every activation of both nets in registers and both weight sets in LDS.  Any other depth up to 8 + 8
# hidden layers -- tcnn's default 5 + 5 first of all -- runs on the STREAMED fused kernels (k_nerf_mlp_fwd_deep / _bwd_deep: a
# workgroup takes its sample tiles through one layer at a time, the layers' weights pass through LDS), round 5.  The layer-by-layer
# path on the fp32 linear kernels of csrc/xr_gemm.hip (activations through HBM; rounds 3-4) stays as the independent statement the
# tests hold the fused kernels against (empty these two tuples to take it).
MAX_HIDDEN = 8          # XR_MLP_MAX_HIDDEN (csrc/xr_mlp.hip)
_FUSED_FWD = tuple((a, b) for a in range(1, MAX_HIDDEN + 1) for b in range(1, MAX_HIDDEN + 1))
_FUSED_BWD = _FUSED_FWD


def _net_layers(w_flat, n_hidden, n_in=32, width=64, n_out=16):
    """views [out, in] of a flat FullyFusedMLP parameter vector (row-major matrices in layer order)"""
    dims = [n_in] + [width] * n_hidden + [n_out]
    out, o = [], 0
    for a, b in zip(dims[:-1], dims[1:]):
        out.append(w_flat[o:o + a * b].view(b, a))
        o += a * b
    if o != w_flat.numel():
        raise _lib.XrError('parameter vector of %d floats does not hold %d hidden layers' % (w_flat.numel(), n_hidden))
    return out


def _layered_net(x, w_flat, n_hidden):
    from .linear import linear_act
    ws = _net_layers(w_flat, n_hidden)
    h = x
    for w in ws[:-1]:
        h = linear_act(h, w, None, True)
    return linear_act(h, ws[-1], None, False)


def _valid_rows(x, n_dev):
    """contiguous copy of the row-major [m, c] tensor `x` with the rows behind the device-side count n_dev[0] set to zero"""
    if n_dev is None:
        return x.contiguous()
    keep = torch.arange(x.shape[0], device=x.device)[:, None] < n_dev.reshape(-1)[:1].to(torch.int64)
    return torch.where(keep, x, torch.zeros((), dtype=x.dtype, device=x.device)).contiguous()


def _layered_nerf_mlp(enc, dirs, w_density, w_color, nhd, nhc, pad_value):
    """enc [n,32] (, dirs [n,3]) -> raw [n,4] = [rgb raw, sigma raw]; differentiable w.r.t. enc, w_density, w_color"""
    dout = _layered_net(enc, w_density, nhd)
    if dirs is None:
        z = torch.zeros((enc.shape[0], 3), dtype=torch.float32, device=enc.device)
        return torch.cat([z, dout[:, :1]], 1)
    cin = torch.cat([dout[:, 1:16], sh4(dirs), torch.full((enc.shape[0], 1), float(pad_value), dtype=torch.float32, device=enc.device)], 1)
    cout = _layered_net(cin, w_color, nhc)
    return torch.cat([cout[:, :3], dout[:, :1]], 1)


def nerf_mlp_fwd(enc_t, dirs, n, w_density, w_color, nhd, nhc, pad_value=1.0, raw=None, n_dev=None, rows=None, row0=0,
                 count=None):
    L = _lib.load()
    if (nhd, nhc) not in _FUSED_FWD:
        if rows is not None:
            raise _lib.XrError('row lists are served by the fused kernels only (hidden layers %d, %d)' % (nhd, nhc))
        m = n if count is None else count
        with torch.no_grad(), _span('xr_nerf_mlp_fwd', 0 if n_dev is not None else m, train=n_dev is not None):
            enc = _valid_rows(enc_t[:, row0:row0 + m].t(), n_dev)
            d = _pos_view(dirs)[0][row0:row0 + m] if dirs is not None else None
            out = _layered_nerf_mlp(enc, d.contiguous() if d is not None and d.stride(0) != 3 else d, w_density.detach(),
                                    w_color.detach() if w_color is not None else None, nhd, nhc, pad_value)
        if raw is None:
            raw = torch.empty((n, 4), dtype=torch.float32, device=enc_t.device)
        raw[row0:row0 + m] = out
        return raw
    if raw is None:
        raw = torch.empty((n, 4), dtype=torch.float32, device=enc_t.device)
    if dirs is not None:
        dirs, ds = _pos_view(dirs)
        dp = C.c_void_p(dirs.data_ptr() + 4 * ds * row0)
    else:
        ds, dp = 0, None
    _ptr(enc_t); _ptr(raw)
    if count is not None:
        n = count
    mlp_range_word(enc_t.device)
    with _span('xr_nerf_mlp_fwd', 0 if n_dev is not None else n, train=n_dev is not None):
        _lib.check(L.xr_nerf_mlp_fwd(_mlp_mode(nhd, nhc), C.c_void_p(enc_t.data_ptr() + 4 * row0), enc_t.shape[1], dp, ds, n, _ptr(n_dev),
                                     _ptr(rows), _ptr(w_density), _ptr(w_color) if w_color is not None else None, nhd,
                                     nhc, pad_value, C.c_void_p(raw.data_ptr() + 16 * row0), _stream()),
                   'xr_nerf_mlp_fwd')
    return raw


def nerf_density_splat(enc_t, n, w_density, nhd, nhc, indices, grid_tmp):
    """K9's density query + K8 in one launch: the density network over the n encoded points of enc_t, exp(density) * min_step merged into
    grid_tmp[indices[i]] by maximum from the forward kernel's epilogue (fused topologies only: see density_splat_supported)"""
    if indices.dtype != torch.int32 or grid_tmp.dtype != torch.float32:
        raise _lib.XrError('nerf_density_splat: int32 indices, float32 grid')
    mlp_range_word(enc_t.device)
    with _span('xr_nerf_mlp_fwd', n, train=False):
        _lib.check(_lib.load().xr_nerf_density_splat(_mlp_mode(nhd, nhc), _ptr(enc_t), enc_t.shape[1], n, _ptr(w_density), nhd, nhc, _ptr(indices),
                                                     _ptr(grid_tmp), _stream()), 'xr_nerf_density_splat')


def density_splat_supported(nhd, nhc):
    return (nhd, nhc) in _FUSED_FWD and (_PRECISION != 'f16' or (nhd, nhc) == (1, 2))


LIVE_STATS = None      # the backward's 4-word count block (words 1, 2: running live / valid row totals; clear to restart)


def _list_slots(dev, n, nhd=1, nhc=2):
    """the list area of the MLP backward's workspace for n rows (at its start, whatever the topology the workspace is sized for)
    -> (workspace, rows view, seg pointer, count-block view)"""
    L = _lib.load()
    ws = _ws(dev, L.xr_nerf_mlp_bwd_workspace_bytes(n, nhd, nhc), 'mlpbwd')
    p_rows, p_seg, p_cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.xr_nerf_mlp_bwd_list_slots(_ptr(ws), ws.numel(), n, C.byref(p_rows), C.byref(p_seg), C.byref(p_cnt)),
               'xr_nerf_mlp_bwd_list_slots')
    base = ws.data_ptr()
    rows = ws[p_rows.value - base:p_rows.value - base + 4 * n].view(torch.int32)
    cnt = ws[p_cnt.value - base:p_cnt.value - base + 16].view(torch.int32)   # [count, running live total, running valid total, spare]
    return ws, rows, p_seg, cnt


def train_loss_scalars(rgb, target, alpha, delta=0.1, scale=5.0, out=None):
    """-> out[2] = (scale * sum HuberLoss(rgb - target), sum ((rgb - target) * alpha)^2), written by one fixed-order sum"""
    if out is None:
        out = torch.empty((2,), dtype=torch.float32, device=rgb.device)
    _lib.check(_lib.load().xr_train_loss_scalars(_ptr(rgb), _ptr(target), _ptr(alpha), rgb.shape[0], float(delta), float(scale),
                                                 _ptr(out), _stream()), 'xr_train_loss_scalars')
    return out


def live_segments(n):
    """number of 1024-row segments of n rows (length of the compositor's live-row count array)"""
    return (int(n) + _lib.LIVE_SEGMENT_ROWS - 1) // _lib.LIVE_SEGMENT_ROWS


def live_rows(draw, n, n_dev=None, zero_denc_t=None, seg_counts=None):
    """-> (rows int32 [n], n_live int32 [4]) on the device: the rows of dL/d(raw) [n,4] that are not exactly zero, in
    order (xr_live_rows); n_live[0] is the count.  Handed to nerf_mlp_bwd and hashgrid_bwd as `live=`.  The list sits in
    the MLP backward's workspace (where xr_ngp_train_step keeps it too) and is overwritten by the next call.
    seg_counts: the per-segment counts composite_train(..., live_seg=) left (one launch less: the ranking pass only)."""
    global LIVE_STATS
    _, rows, p_seg, n_live = _list_slots(draw.device, n)
    LIVE_STATS = n_live
    with _span('xr_live_rows', 0 if n_dev is not None else n, train=n_dev is not None):
        _lib.check(_lib.load().xr_live_rows(_ptr(draw), n, _ptr(n_dev), p_seg if seg_counts is None else _ptr(seg_counts), _ptr(rows),
                                             _ptr(n_live), _ptr(zero_denc_t), zero_denc_t.shape[1] if zero_denc_t is not None else 0,
                                             0 if seg_counts is None else 1, _stream()), 'xr_live_rows')
    return rows, n_live


def nerf_mlp_bwd(enc_t, dirs, n, w_density, w_color, nhd, nhc, draw, grad_wd, grad_wc, pad_value=1.0, denc_t=None,
                 n_dev=None, row0=0, count=None, live=None):
    """`live=(rows, n_live)` (ops.live_rows(draw, ...)): the backward computes the listed rows only and leaves the other
    rows of denc_t untouched -- pass the same list to hashgrid_bwd.  Without it the call builds its own list and writes
    exact zeros to the dead rows (identical results to the backward over every row)."""
    L = _lib.load()
    if (nhd, nhc) not in _FUSED_BWD:
        # layer by layer: recompute the forward under autograd, back-propagate dL/d(raw); rows outside a live list have an
        # exactly-zero dL/d(raw) and get an exactly-zero dL/d(encoding) (a superset of the fused contract, which leaves them alone)
        m = n if count is None else count
        if denc_t is None:
            denc_t = torch.empty_like(enc_t)
        with _span('xr_nerf_mlp_bwd', 0 if n_dev is not None else m, train=n_dev is not None):
            d = _pos_view(dirs)[0][row0:row0 + m]
            d = d.contiguous() if d.stride(0) != 3 else d
            with torch.enable_grad():
                # rows behind the device-side count are padding of a fixed-size buffer that nothing wrote this step (the encode and the
                # compositor stop at the count): whatever they hold -- stale rows, NaN bit patterns of fresh memory -- is replaced by
                # zeros, not multiplied by zero (0 * NaN), in the features AND in the incoming gradient
                enc = _valid_rows(enc_t[:, row0:row0 + m].t(), n_dev).requires_grad_(True)
                wd, wc = w_density.detach().requires_grad_(True), w_color.detach().requires_grad_(True)
                out = _layered_nerf_mlp(enc, d, wd, wc, nhd, nhc, pad_value)
                g = _valid_rows(draw[row0:row0 + m], n_dev)
                ge, gwd, gwc = torch.autograd.grad(out, [enc, wd, wc], grad_outputs=g)
            denc_t[:, row0:row0 + m] = ge.t()
            grad_wd.add_(gwd)
            grad_wc.add_(gwc)
        return denc_t
    dirs, ds = _pos_view(dirs)
    if denc_t is None:
        denc_t = torch.empty_like(enc_t)
    ws = _ws(enc_t.device, L.xr_nerf_mlp_bwd_workspace_bytes(n, nhd, nhc), 'mlpbwd')
    _ptr(enc_t); _ptr(draw); _ptr(denc_t)
    if count is not None:
        n = count
    with _span('xr_nerf_mlp_bwd', 0 if n_dev is not None else n, train=n_dev is not None):
        _lib.check(L.xr_nerf_mlp_bwd(_mlp_mode(nhd, nhc), C.c_void_p(enc_t.data_ptr() + 4 * row0), enc_t.shape[1], C.c_void_p(dirs.data_ptr() + 4 * ds * row0), ds, n, _ptr(n_dev), _ptr(w_density),
                                     _ptr(w_color), nhd, nhc, pad_value, C.c_void_p(draw.data_ptr() + 16 * row0), C.c_void_p(denc_t.data_ptr() + 4 * row0), _ptr(grad_wd),
                                     _ptr(grad_wc), _ptr(ws), ws.numel(), _ptr(live[0]) if live is not None else None,
                                     _ptr(live[1]) if live is not None else None, _stream()), 'xr_nerf_mlp_bwd')
    return denc_t


# ---------------------------------------------------------------- callers either side
def gen_rays(pose43, H, W, fx, fy, cx, cy, row0=0, nrows=None, device='cuda'):
    pose = np.ascontiguousarray(np.asarray(pose43, dtype=np.float32).reshape(4, 3))
    nrows = H - row0 if nrows is None else nrows
    o = torch.empty((nrows * W, 3), dtype=torch.float32, device=device)
    d = torch.empty((nrows * W, 3), dtype=torch.float32, device=device)
    _lib.check(_lib.load().xr_gen_rays(pose.ctypes.data, H, W, fx, fy, cx, cy, row0, nrows, _ptr(o), _ptr(d),
                                       _stream()), 'xr_gen_rays')
    return o, d


def huber_loss_grad(rgb, target, delta=0.1, scale=5.0):
    grad = torch.empty_like(rgb)
    loss = torch.zeros((1,), dtype=torch.float32, device=rgb.device)
    _lib.check(_lib.load().xr_huber_loss_grad(_ptr(rgb), _ptr(target), None, rgb.numel(), delta, scale, _ptr(grad),
                                              _ptr(loss), _stream()), 'xr_huber_loss_grad')
    return loss, grad


def huber_loss_grad_mse(rgb, target, alpha, delta=0.1, scale=5.0, out=None):
    """-> out[2] = (scale * HuberLoss_sum, sum ((rgb-target)*alpha)^2), dL/drgb; `out` must be zero-filled"""
    grad = torch.empty_like(rgb)
    if out is None:
        out = torch.zeros((2,), dtype=torch.float32, device=rgb.device)
    _lib.check(_lib.load().xr_huber_loss_grad(_ptr(rgb), _ptr(target), _ptr(alpha), 3 * rgb.shape[0], delta, scale,
                                              _ptr(grad), _ptr(out), _stream()), 'xr_huber_loss_grad')
    return out, grad


def make_batch_buffers(capacity, device):
    """caller-owned output buffers for make_batch(out=...), `capacity` rays"""
    f = lambda c: torch.empty((capacity, c), dtype=torch.float32, device=device)
    return {'rays_o': f(3), 'rays_d': f(3), 'target_s': f(3), 'alpha': f(1), 'bg_color': f(3),
            'img_ids': torch.empty((capacity, 1), dtype=torch.int32, device=device)}


def make_batch(rows, n, call_index, seed=20220901, out=None):
    """rows: contiguous [>=n, 11] slice of the device-resident ray table -> dict of batch tensors
    (views of `out`'s buffers when given: no allocation, nothing for the caching allocator to track across streams)"""
    dev = rows.device
    if out is not None:
        o, d, tgt, alpha, bg, ids = (out[k][:n] for k in ('rays_o', 'rays_d', 'target_s', 'alpha', 'bg_color', 'img_ids'))
    else:
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        o, d, tgt, alpha, bg = f(n, 3), f(n, 3), f(n, 3), f(n, 1), f(n, 3)
        ids = torch.empty((n, 1), dtype=torch.int32, device=dev)
    st, inc = pcg32_host_state(call_index, seed)
    # (one batch = a series of one: rows from the slice's first row, the generator of call `call_index`)
    _lib.check(_lib.load().xr_make_batch_series(_ptr(rows), (C.c_uint64 * 1)(0), n, 1, n, st, inc, _ptr(o), _ptr(d), _ptr(tgt), _ptr(alpha),
                                                _ptr(bg), _ptr(ids), _stream()), 'xr_make_batch_series')
    return {'rays_o': o, 'rays_d': d, 'target_s': tgt, 'alpha': alpha, 'img_ids': ids, 'bg_color': bg}


def adam_step(p, g, m, v, step, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-15, weight_decay=1e-6, ema=None,
              ema_momentum=0.05):
    adam_step_multi([p], [g], [m], [v], step, lr, beta1, beta2, eps, weight_decay, [ema] if ema is not None else None, ema_momentum)


def scale_multi(ts, scale_dev=None, host_factor=1.0):
    """ts[k] *= scale_dev * host_factor in one launch (no memory traffic when the factor is exactly 1)"""
    k = len(ts)
    for t in ts:
        _ptr(t)
    arr = (C.c_void_p * k)(*[t.data_ptr() for t in ts])
    ns = (C.c_size_t * k)(*[t.numel() for t in ts])
    _lib.check(_lib.load().xr_scale_multi(k, arr, ns, _ptr(scale_dev), float(host_factor), _stream()), 'xr_scale_multi')


def adam_step_multi(ps, gs, ms, vs, step, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-15, weight_decay=1e-6, emas=None,
                    ema_momentum=0.05, grad_scale=1.0):
    """one launch for up to 4 tensors; `grad_scale` multiplies the gradients as they are read (the buffers stay as they are)"""
    k = len(ps)
    arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() if t is not None else None for t in ts])
    for t in list(ps) + list(gs) + list(ms) + list(vs):
        _ptr(t)     # validates device / contiguity
    ns = (C.c_size_t * k)(*[p.numel() for p in ps])
    with _span('xr_adam_step', sum(p.numel() for p in ps)):
        _lib.check(_lib.load().xr_adam_step_multi(k, arr(ps), arr(gs), arr(ms), arr(vs), arr(emas) if emas else None, ns,
                                                  step, lr, beta1, beta2, eps, weight_decay, ema_momentum, float(grad_scale),
                                                  _stream()),
                   'xr_adam_step_multi')


# ---------------------------------------------------------------- Mip-NeRF stages (BASELINE config #3)
def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.XrError('xrnerf_amd mip ops take float32 tensors (got %s)' % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def mip_zvals(near, far, n_z, lindisp=False, z_rand=None):
    """GetZvals (create.py:486-531): near/far [R] or [R,1]; z_rand [R,n_z] uniform draws or None -> [R,n_z]"""
    near, far = _f32c(near.reshape(-1)), _f32c(far.reshape(-1))
    R = near.shape[0]
    out = torch.empty((R, n_z), dtype=torch.float32, device=near.device)
    if z_rand is not None:
        z_rand = _f32c(z_rand)
        assert tuple(z_rand.shape) == (R, n_z)
    _lib.check(_lib.load().xr_mip_zvals(_ptr(near), _ptr(far), R, n_z, int(bool(lindisp)), _ptr(z_rand), _ptr(out),
                                        _stream()), 'xr_mip_zvals')
    return out


def mip_encode_channels(min_deg, max_deg, min_deg_view, max_deg_view, append_identity=True):
    return int(_lib.load().xr_mip_encode_channels(min_deg, max_deg, min_deg_view, max_deg_view, int(bool(append_identity))))


def mip_encode(rays_o, rays_d, viewdirs, radii, z_vals, min_deg, max_deg, min_deg_view, max_deg_view,
               append_identity=True, ray_shape='cone', out=None):
    """cast_rays + integrated_pos_enc + pos_enc + concat in one launch -> [R*(n_z-1), channels]"""
    L = _lib.load()
    R, n_z = z_vals.shape
    ch = mip_encode_channels(min_deg, max_deg, min_deg_view, max_deg_view, append_identity)
    if out is None:
        # rows padded to a multiple of 4 floats: the encoding (123 channels in the reference's config) is then a [M, ch] view whose rows start
        # on 16-byte boundaries, which the linear kernels read in place (ops._rows) -- no contiguous copy of the first layer's input
        out = torch.empty((R * (n_z - 1), (ch + 3) // 4 * 4), dtype=torch.float32, device=z_vals.device)[:, :ch]
    if not _on_device(out) or out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1 or out.shape[1] < ch:
        raise _lib.XrError('mip_encode: out must be a float32 device matrix with contiguous rows of >= %d channels' % ch)
    shape = {'cone': 0, 'cylinder': 1}[ray_shape]
    with _span('xr_mip_encode', R * (n_z - 1)):
        _lib.check(L.xr_mip_encode(_ptr(_f32c(rays_o)), _ptr(_f32c(rays_d)), _ptr(_f32c(viewdirs)),
                                   _ptr(_f32c(radii.reshape(-1))), _ptr(_f32c(z_vals)), R, n_z, min_deg, max_deg,
                                   min_deg_view, max_deg_view, int(bool(append_identity)), shape, C.c_void_p(out.data_ptr()),
                                   out.stride(0), _stream()), 'xr_mip_encode')
    return out


def mip_encode_gaussians(means, covs, viewdirs, min_deg, max_deg, min_deg_view, max_deg_view, append_identity=True):
    L = _lib.load()
    R, S = means.shape[:2]
    ch = mip_encode_channels(min_deg, max_deg, min_deg_view, max_deg_view, append_identity)
    out = torch.empty((R * S, ch), dtype=torch.float32, device=means.device)
    _lib.check(L.xr_mip_encode_gaussians(_ptr(_f32c(means)), _ptr(_f32c(covs)), _ptr(_f32c(viewdirs)), R, S, min_deg,
                                         max_deg, min_deg_view, max_deg_view, int(bool(append_identity)), _ptr(out),
                                         ch, _stream()), 'xr_mip_encode_gaussians')
    return out


_MIP_ACT = {'softplus': 0, 'relu': 1}


def mip_render_forward(raw, z_vals, rays_d, density_bias, rgb_padding, white_bkgd, density_activation='softplus'):
    """-> rgb [R,3], distance [R], acc [R], weights [R,S]"""
    L = _lib.load()
    R, n_z = z_vals.shape
    dev = raw.device
    raw = _f32c(raw)
    assert tuple(raw.shape) == (R, n_z - 1, 4), 'raw must be [R, n_z-1, 4] (rgb + density)'
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    dist = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty((R,), dtype=torch.float32, device=dev)
    w = torch.empty((R, n_z - 1), dtype=torch.float32, device=dev)
    with _span('xr_mip_render_forward', R * (n_z - 1)):
        _lib.check(L.xr_mip_render_forward(_ptr(raw), _ptr(_f32c(z_vals)), _ptr(_f32c(rays_d)), R, n_z,
                                           float(density_bias), float(rgb_padding), int(bool(white_bkgd)),
                                           _MIP_ACT[density_activation], _ptr(rgb), _ptr(dist), _ptr(acc), _ptr(w),
                                           _stream()), 'xr_mip_render_forward')
    return rgb, dist, acc, w


def mip_render_backward(raw, z_vals, rays_d, grad_rgb, density_bias, rgb_padding, white_bkgd,
                        density_activation='softplus'):
    L = _lib.load()
    R, n_z = z_vals.shape
    raw = _f32c(raw)
    out = torch.empty_like(raw)
    with _span('xr_mip_render_backward', R * (n_z - 1)):
        _lib.check(L.xr_mip_render_backward(_ptr(raw), _ptr(_f32c(z_vals)), _ptr(_f32c(rays_d)), _ptr(_f32c(grad_rgb)),
                                            R, n_z, float(density_bias), float(rgb_padding), int(bool(white_bkgd)),
                                            _MIP_ACT[density_activation], _ptr(out), _stream()),
                   'xr_mip_render_backward')
    return out


def mip_resample(z_vals, weights, resample_padding, rand=None):
    """resample_along_rays' new (detached) z_vals [R,n_z]; rand [R,n_z] uniform draws (randomized) or None"""
    L = _lib.load()
    R, n_z = z_vals.shape
    z_vals, weights = _f32c(z_vals), _f32c(weights.detach())
    assert tuple(weights.shape) == (R, n_z - 1)
    if rand is not None:
        rand = _f32c(rand)
        assert tuple(rand.shape) == (R, n_z)
    out = torch.empty_like(z_vals)
    with _span('xr_mip_resample', R):
        _lib.check(L.xr_mip_resample(_ptr(z_vals), _ptr(weights), _ptr(rand), float(resample_padding), R, n_z, _ptr(out),
                                     _stream()), 'xr_mip_resample')
    return out


# ---------------------------------------------------------------- KiloNeRF rendering (BASELINE config #5)
def kilo_param_floats(pos_freqs, dir_freqs, n_hidden):
    return int(_lib.load().xr_kilo_param_floats(pos_freqs, dir_freqs, n_hidden))


_KILO_WS_GEN = [0, 0]     # [count bumped by every call that rewrites the 'kilo' workspace's assignment arrays, that workspace's address]


def kilo_ws_generation():
    return _KILO_WS_GEN[0]


def kilo_pack_params(tensors, pos_freqs, dir_freqs, n_hidden, out=None):
    """MultiNetwork.ordered_parameters() -> the packed blocks [N, stride] the kernels take, in one launch (xr_kilo_pack_params)"""
    L = _lib.load()
    N = tensors[0].shape[0]
    stride = int(L.xr_kilo_param_floats(int(pos_freqs), int(dir_freqs), int(n_hidden)))
    ts = [_f32c(t.detach()) for t in tensors]
    if out is None:
        out = torch.empty((N, stride), dtype=torch.float32, device=ts[0].device)
    arr = (C.c_void_p * len(ts))(*[_ptr(t).value for t in ts])
    _lib.check(L.xr_kilo_pack_params(arr, N, int(pos_freqs), int(dir_freqs), int(n_hidden), _ptr(out), out.stride(0), _stream()), 'xr_kilo_pack_params')
    return out


def kilo_unpack_grads(blocks, like, pos_freqs, dir_freqs, n_hidden, clear=True):
    """the packed gradient blocks -> one contiguous tensor per parameter of `like` (views of ONE allocation), in one launch; clear: the
    blocks are left zero-filled for the next backward"""
    L = _lib.load()
    sizes = [t.numel() for t in like]
    flat = torch.empty((sum(sizes),), dtype=torch.float32, device=blocks.device)
    outs, o = [], 0
    for t, n in zip(like, sizes):
        outs.append(flat[o:o + n].view(t.shape))
        o += n
    arr = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    _lib.check(L.xr_kilo_unpack_grads(_ptr(blocks), blocks.stride(0), blocks.shape[0], int(pos_freqs), int(dir_freqs), int(n_hidden), arr,
                                      1 if clear else 0, _stream()), 'xr_kilo_unpack_grads')
    return outs


def kilo_mlp_forward(viewdirs, gmin, gmax, fixed_res, occ_res, occupancy, domain_mins, domain_maxs, params, pos_freqs,
                     dir_freqs, n_hidden, pts=None, rays_o=None, rays_d=None, z_vals=None, want_counts=False):
    """KiloNerfMLP.forward: raw [R,S,4] (zeros where no network is evaluated) (+ batch_size_per_network [N] int32).
    Samples: pts [R,S,3], or rays_o/rays_d [R,3] + z_vals [R,S] (o + d*z, never materialised).
    gmin/gmax: 3 python floats; fixed_res/occ_res: 3 ints; occupancy: bool/uint8 device grid or None."""
    L = _lib.load()
    if pts is not None:
        pts = _f32c(pts)
        R, S = pts.shape[0], pts.shape[1]
        dev = pts.device
    else:
        rays_o, rays_d, z_vals = _f32c(rays_o), _f32c(rays_d), _f32c(z_vals)
        R, S = z_vals.shape
        dev = z_vals.device
    N = params.shape[0]
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=dev)
    counts = torch.empty((N,), dtype=torch.int32, device=dev) if want_counts else None
    ws = _ws(dev, L.xr_kilo_workspace_bytes(R * S, N), 'kilo')
    _KILO_WS_GEN[0] += 1              # the workspace now holds THIS call's assignment (kilo_mlp_backward(reuse=...) checks the count
    _KILO_WS_GEN[1] = ws.data_ptr()   # ... and that it is handed the same memory)
    if occupancy is not None:
        occupancy = occupancy.reshape(-1)
        if occupancy.dtype == torch.bool:
            occupancy = occupancy.view(torch.uint8)
        if occupancy.dtype != torch.uint8:
            raise _lib.XrError('occupancy grid must be bool or uint8')
    f3 = (C.c_float * 3)
    i3 = (C.c_int32 * 3)
    with _span('xr_kilo_mlp_forward', R * S):
        _lib.check(L.xr_kilo_mlp_forward(_ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(z_vals), _ptr(_f32c(viewdirs)), R, S,
                                         f3(*[float(v) for v in gmin]), f3(*[float(v) for v in gmax]),
                                         i3(*[int(v) for v in fixed_res]), i3(*[int(v) for v in occ_res]) if occ_res is not None else None,
                                         _ptr(occupancy), _ptr(_f32c(domain_mins)), _ptr(_f32c(domain_maxs)), _ptr(params),
                                         params.stride(0), N, int(pos_freqs), int(dir_freqs), int(n_hidden), _ptr(raw),
                                         _ptr(counts), _ptr(ws), ws.numel(), _stream()), 'xr_kilo_mlp_forward')
    return (raw, counts) if want_counts else raw


def kilo_mlp_backward(draw, viewdirs, gmin, gmax, fixed_res, occ_res, occupancy, domain_mins, domain_maxs, params, pos_freqs,
                      dir_freqs, n_hidden, pts=None, rays_o=None, rays_d=None, z_vals=None, reuse_generation=None, grad=None):
    """gradient of the packed parameter blocks [N, stride] for dL/draw [R,S,4] (same sample arguments as the forward).
    reuse_generation: kilo_ws_generation() as it stood right after the forward call on these samples -- if nothing has touched the
    workspace since, the assignment / offsets / scatter launches are skipped.  grad: a zero-filled [N, stride] buffer to accumulate into."""
    L = _lib.load()
    draw = _f32c(draw)
    if pts is not None:
        pts = _f32c(pts)
        R, S = pts.shape[0], pts.shape[1]
    else:
        rays_o, rays_d, z_vals = _f32c(rays_o), _f32c(rays_d), _f32c(z_vals)
        R, S = z_vals.shape
    dev = draw.device
    N = params.shape[0]
    if grad is None:
        grad = torch.zeros_like(params)
    ws = _ws(dev, L.xr_kilo_workspace_bytes(R * S, N), 'kilo')
    reuse = reuse_generation is not None and reuse_generation == _KILO_WS_GEN[0] and ws.data_ptr() == _KILO_WS_GEN[1]
    if not reuse:
        _KILO_WS_GEN[0] += 1
    if occupancy is not None:
        occupancy = occupancy.reshape(-1)
        if occupancy.dtype == torch.bool:
            occupancy = occupancy.view(torch.uint8)
    f3 = (C.c_float * 3)
    i3 = (C.c_int32 * 3)
    with _span('xr_kilo_mlp_backward', R * S):
        _lib.check(L.xr_kilo_mlp_backward(_ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(z_vals), _ptr(_f32c(viewdirs)), R, S,
                                          f3(*[float(v) for v in gmin]), f3(*[float(v) for v in gmax]),
                                          i3(*[int(v) for v in fixed_res]), i3(*[int(v) for v in occ_res]) if occ_res is not None else None,
                                          _ptr(occupancy), _ptr(_f32c(domain_mins)), _ptr(_f32c(domain_maxs)), _ptr(params),
                                          params.stride(0), N, int(pos_freqs), int(dir_freqs), int(n_hidden), _ptr(draw),
                                          _ptr(grad), 1 if reuse else 0, _ptr(ws), ws.numel(), _stream()), 'xr_kilo_mlp_backward')
    return grad


def nerf_render_forward(raw, z_vals, rays_d, white_bkgd):
    """NerfRender.forward (inference): -> rgb [R,3], disp [R], acc [R], weights [R,S]"""
    L = _lib.load()
    raw, z_vals = _f32c(raw), _f32c(z_vals)
    R, S = z_vals.shape
    assert tuple(raw.shape) == (R, S, 4)
    dev = raw.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty((R,), dtype=torch.float32, device=dev)
    w = torch.empty((R, S), dtype=torch.float32, device=dev)
    with _span('xr_nerf_render_forward', R * S):
        _lib.check(L.xr_nerf_render_forward(_ptr(raw), _ptr(z_vals), _ptr(_f32c(rays_d)), R, S, int(bool(white_bkgd)),
                                            _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(w), _stream()), 'xr_nerf_render_forward')
    return rgb, disp, acc, w


def kilo_render_rays(rays_o, rays_d, viewdirs, near, far, n_samples, gmin, gmax, fixed_res, occ_res, occupancy, domain_mins,
                     domain_maxs, params, pos_freqs, dir_freqs, n_hidden, white_bkgd=True, lindisp=False):
    """GetZvals(not randomized) + GetPts + KiloNerfMLP.forward + NerfRender.forward in one call (the frame path of the
    real-time bench): -> rgb [R,3], disp [R], acc [R].  near / far: [R] device tensors."""
    L = _lib.load()
    rays_o, rays_d, viewdirs = _f32c(rays_o), _f32c(rays_d), _f32c(viewdirs)
    near, far = _f32c(near.reshape(-1)), _f32c(far.reshape(-1))
    R = rays_o.shape[0]
    dev = rays_o.device
    N = params.shape[0]
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty((R,), dtype=torch.float32, device=dev)
    ws = _ws(dev, L.xr_kilo_render_workspace_bytes(R, int(n_samples), N), 'kilo_frame')
    if occupancy is not None:
        occupancy = occupancy.reshape(-1)
        if occupancy.dtype == torch.bool:
            occupancy = occupancy.view(torch.uint8)
    f3 = (C.c_float * 3)
    i3 = (C.c_int32 * 3)
    with _span('xr_kilo_render_rays', R * n_samples):
        _lib.check(L.xr_kilo_render_rays(_ptr(rays_o), _ptr(rays_d), _ptr(viewdirs), _ptr(near), _ptr(far), R, int(n_samples),
                                         int(bool(lindisp)), f3(*[float(v) for v in gmin]), f3(*[float(v) for v in gmax]),
                                         i3(*[int(v) for v in fixed_res]), i3(*[int(v) for v in occ_res]) if occ_res is not None else None,
                                         _ptr(occupancy), _ptr(_f32c(domain_mins)), _ptr(_f32c(domain_maxs)), _ptr(params),
                                         params.stride(0), N, int(pos_freqs), int(dir_freqs), int(n_hidden),
                                         int(bool(white_bkgd)), _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(ws), ws.numel(),
                                         _stream()), 'xr_kilo_render_rays')
    return rgb, disp, acc


# ---------------------------------------------------------------- fp32 MFMA linear layers (8x256 NeRF MLP)
def linear_ok(x, w):
    """shapes / alignment the MFMA kernel takes: fp32 device tensors, K and N multiples of 4"""
    return (_on_device(x) and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 4 == 0 and w.shape[0] % 4 == 0 and x.shape[0] > 0)


def _rows(t):
    """(tensor as the linear kernels take it, row stride): a [M, C] fp32 matrix whose rows are contiguous and start on 16-byte boundaries --
    dense, or a column range of a wider buffer.  Anything else is copied."""
    if t.dtype != torch.float32:
        raise _lib.XrError('the linear kernels take float32 tensors (got %s)' % t.dtype)
    if not _on_device(t):
        raise _lib.XrError('xrnerf_amd ops need ROCm device tensors (got a %s tensor): there is no CPU fallback' % t.device)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t, t.stride(0)
    t = t.contiguous()
    return t, t.shape[1]


def linear_forward(x, w, bias, relu, out=None):
    """y [M,N] = act(x [M,K] . w [N,K]^T + bias).  x and `out` may be column ranges of wider buffers (row strides)."""
    x, ldx = _rows(x)
    w = _f32c(w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device) if out is None else out
    if tuple(y.shape) != (M, N) or y.stride(1) != 1 or y.stride(0) % 4 or y.data_ptr() % 16 or y.dtype != torch.float32:
        raise _lib.XrError('linear_forward: out must be an [M, N] float32 view with contiguous, 16-byte aligned rows')
    with _span('xr_linear_forward', M):
        _lib.check(_lib.load().xr_linear_forward(C.c_void_p(x.data_ptr()), ldx, _ptr(w), _ptr(_f32c(bias)) if bias is not None else None, M, N, K,
                                                 int(bool(relu)), C.c_void_p(y.data_ptr()), y.stride(0), _stream()), 'xr_linear_forward')
    return y


def _dy_and_mask(dy, mask_src):
    dy, ld = _rows(dy)
    if mask_src is not None:
        mask_src, ldm = _rows(mask_src)
        if ldm != ld:                                  # one stride for both in the kernel: bring them to dense rows
            dy, mask_src = dy.contiguous(), mask_src.contiguous()
            ld = dy.shape[1]
    return dy, mask_src, ld


def linear_backward_input(dy, mask_src, w, w_t=None):
    """dx [M,K] = (dy where mask_src > 0) . w.  The weight goes in transposed (one 64 K-element copy, or the caller's `w_t`): both operands
    of the product are then [rows, contraction] like the forward's and it runs on the forward's split-operand kernel instead of the fp32
    MFMA (1.6x).  dy / mask_src may be column ranges of wider buffers (same row stride)."""
    dy, mask_src, ld = _dy_and_mask(dy, mask_src)
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    if w_t is None:
        w_t = _f32c(w).t().contiguous()
    with _span('xr_linear_backward_input', M):
        _lib.check(_lib.load().xr_linear_backward_input(C.c_void_p(dy.data_ptr()), ld, C.c_void_p(mask_src.data_ptr()) if mask_src is not None else None,
                                                        _ptr(w_t), 1, M, N, K, _ptr(dx), _stream()), 'xr_linear_backward_input')
    return dx


def sum_partials(part):
    """[splits, n] per-range partial sums -> [n]: one fixed-order launch (xr_sum_partials)"""
    if part.shape[0] == 1:
        return part[0]
    out = torch.empty((part.shape[1],), dtype=torch.float32, device=part.device)
    _lib.check(_lib.load().xr_sum_partials(_ptr(part), part.shape[0], part.stride(0), part.shape[1], _ptr(out), _stream()), 'xr_sum_partials')
    return out


def linear_backward_weight(dy, mask_src, x):
    """dw [N,K] = (dy where mask_src > 0)^T . x   (fixed-order sum of the per-M-range partials)"""
    L = _lib.load()
    dy, mask_src, ld = _dy_and_mask(dy, mask_src)
    x, ldx = _rows(x)
    M, N = dy.shape
    K = x.shape[1]
    splits = int(L.xr_linear_backward_splits(M, N, K))
    part = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
    with _span('xr_linear_backward_weight', M):
        _lib.check(L.xr_linear_backward_weight(C.c_void_p(dy.data_ptr()), ld, C.c_void_p(mask_src.data_ptr()) if mask_src is not None else None,
                                               C.c_void_p(x.data_ptr()), ldx, M, N, K, splits, _ptr(part), None, 0, _stream()),
                   'xr_linear_backward_weight')
    return sum_partials(part.view(splits, N * K)).view(N, K)


def linear_backward_weight_bias(dy, mask_src, x):
    """(dw [N,K], db [N]) of one layer from ONE launch and ONE reduction: the bias gradient's column sums ride on the weight-gradient
    product, and both sets of per-M-range partials sit in one [splits, N K + N] buffer"""
    L = _lib.load()
    dy, mask_src, ld = _dy_and_mask(dy, mask_src)
    x, ldx = _rows(x)
    M, N = dy.shape
    K = x.shape[1]
    splits = int(L.xr_linear_backward_splits(M, N, K))
    part = torch.empty((splits, N * K + N), dtype=torch.float32, device=dy.device)
    with _span('xr_linear_backward_weight', M):
        _lib.check(L.xr_linear_backward_weight(C.c_void_p(dy.data_ptr()), ld, C.c_void_p(mask_src.data_ptr()) if mask_src is not None else None,
                                               C.c_void_p(x.data_ptr()), ldx, M, N, K, splits, _ptr(part), C.c_void_p(part.data_ptr() + 4 * N * K),
                                               N * K + N, _stream()), 'xr_linear_backward_weight')
    g = sum_partials(part)
    return g[:N * K].view(N, K), g[N * K:]


def linear_backward_bias(dy, mask_src):
    """db [N] = column sums of (dy where mask_src > 0)   (fixed-order sum of the per-M-range partials)"""
    L = _lib.load()
    dy = _f32c(dy)
    M, N = dy.shape
    splits = int(L.xr_linear_backward_splits(M, 0, 0))
    part = torch.empty((splits, N), dtype=torch.float32, device=dy.device)
    _lib.check(L.xr_linear_backward_bias(_ptr(dy), _ptr(mask_src), M, N, splits, _ptr(part), _stream()), 'xr_linear_backward_bias')
    return sum_partials(part)
