"""TEST INFRASTRUCTURE ONLY: ctypes access to the host-compiled kernel sources (tests/hip_emu/emubuild.py).  The entry points
are the C-ABI of include/xrnerf_mi355.h; "device" pointers are numpy buffers."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emubuild as _build  # noqa: E402

_libs = {}


def lib(name):
    if name not in _libs:
        _libs[name] = C.CDLL(_build.build(name))
    return _libs[name]


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def aligned(shape, dtype=np.float32, align=256, fill=None):
    """numpy buffer whose data pointer is `align`-byte aligned (the C-ABI checks 16- / 256-byte alignment)"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + n].view(dtype).reshape(shape)
    if fill is not None:
        out[...] = fill
    return out


def check(rc, L):
    if rc != 0:
        L.xr_last_error.restype = C.c_char_p
        raise RuntimeError('rc=%d: %s' % (rc, L.xr_last_error().decode()))


HOST_ONLY = ('xr_ngp_train_step', 'xr_ngp_window_march', 'xr_timing_event_create', 'xr_timing_event_destroy', 'xr_timing_event_elapsed_ms',
             'xr_event_record', 'xr_ngp_loop_run', 'xr_rccl_unique_id', 'xr_rccl_create', 'xr_rccl_destroy',
             'xr_rccl_exchange', 'xr_rccl_exposed_ms')
ALL_SOURCES = ('xr_misc', 'xr_grid', 'xr_raymarch', 'xr_encode', 'xr_mlp', 'xr_mip', 'xr_kilo', 'xr_gemm')


# entry points whose workgroups may run on several host threads: no inter-workgroup atomics, no inter-workgroup ordering
# (xrnerf_amd/csrc/xr_mlp.hip, xr_gemm.hip: per-workgroup partial sums in a workspace + a fixed-order reduce kernel), i.e.
# the results do not depend on the schedule.  These are the MFMA kernels -- a rendezvous of 64 fibers per MFMA -- which
# dominate the emulation time.  XR_EMU_THREADS=1 switches it off.
PARALLEL_OK = ('xr_nerf_mlp_fwd', 'xr_nerf_mlp_bwd', 'xr_mlp_fwd', 'xr_mlp_bwd', 'xr_linear_forward', 'xr_linear_backward_input',
               'xr_linear_backward_weight')
EMU_THREADS = max(1, int(os.environ.get('XR_EMU_THREADS', min(8, os.cpu_count() or 1))))


class _Threaded:
    """ctypes function + `emu_set_threads` of the shared object it lives in"""

    def __init__(self, fn, lib):
        self._fn, self._lib = fn, lib

    def __setattr__(self, k, v):
        if k in ('restype', 'argtypes'):
            setattr(self._fn, k, v)
        else:
            object.__setattr__(self, k, v)

    @property
    def __name__(self):
        return self._fn.__name__

    def __call__(self, *a):
        self._lib.emu_set_threads(EMU_THREADS)
        try:
            return self._fn(*a)
        finally:
            self._lib.emu_set_threads(1)


class MultiLib:
    """the host-compiled kernel sources behind one handle: libxrnerf_mi355.so is ONE library, here every source is its own
    shared object (its internal helpers must not be merged by the linker); a symbol is taken from the object that has it"""

    def __init__(self, names=ALL_SOURCES):
        self._libs = [C.CDLL(_build.build(n)) for n in names]
        self._cache = {}

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        if name not in self._cache:
            for L in self._libs:
                try:
                    fn = getattr(L, name)
                    self._cache[name] = _Threaded(fn, L) if (name in PARALLEL_OK and EMU_THREADS > 1) else fn
                    break
                except AttributeError:
                    continue
            else:
                raise AttributeError(name)
        return self._cache[name]

    def last_errors(self):
        out = []
        for L in self._libs:
            L.xr_last_error.restype = C.c_char_p
            m = L.xr_last_error()
            if m:
                out.append(m.decode())
        return ' | '.join(out)


class emulated_ops:
    """context manager: xrnerf_amd.ops runs on the HIP-on-CPU shim with HOST torch tensors -- the GPU tests' bodies can then
    be executed without a GPU (small sizes).  Only for tests: it swaps the library handle of xrnerf_amd._lib and the three
    device helpers of xrnerf_amd.ops (pointer check, stream, workspace alignment), and restores them on exit."""

    def __enter__(self):
        import torch
        from xrnerf_amd import _lib, ops
        self._lib, self._ops, self._torch = _lib, ops, torch
        self.saved = (_lib._lib, ops._ptr, ops._stream, ops._ws, _lib.check, dict(ops._workspaces), ops._on_device)
        ml = MultiLib()
        for name, (res, args) in _lib.SIGNATURES.items():
            try:
                fn = getattr(ml, name)
            except AttributeError:
                if name in HOST_ONLY:
                    continue          # native host-side executors over the entry points: the emulated run uses the Python path
                raise
            fn.restype, fn.argtypes = res, args
        # xr_last_error: one per object
        _lib._lib = ml

        def ptr(t):
            if t is None:
                return None
            if not t.is_contiguous():
                raise _lib.XrError('tensor must be contiguous')
            return C.c_void_p(t.data_ptr())

        def ws(device, nbytes, tag):
            # persistent per tag and grow-only, like ops._ws: callers rely on a workspace keeping its contents between two entry
            # points (the KiloNeRF backward reuses the forward's assignment arrays; the mlpbwd block holds running totals)
            key, need = (str(device), tag), max(int(nbytes), 256)
            w = ops._workspaces.get(key)
            if w is None or w.numel() < need:
                raw = torch.zeros(need + 256, dtype=torch.uint8)
                off = (-raw.data_ptr()) % 256
                w = ops._workspaces[key] = raw[off:off + need]
            return w

        def check(rc, what=''):
            if rc != 0:
                raise _lib.XrError('%s failed (%d): %s' % (what, rc, ml.last_errors()))
        ops._ptr, ops._stream, ops._ws, _lib.check, ops._on_device = ptr, (lambda: None), ws, check, (lambda t: True)
        ops._workspaces.clear()
        self._sync = torch.cuda.synchronize
        torch.cuda.synchronize = lambda *a, **k: None          # the GPU tests' bodies call it; everything is synchronous here
        return torch.device('cpu')

    def __exit__(self, *exc):
        _lib, ops = self._lib, self._ops
        self._torch.cuda.synchronize = self._sync
        _lib._lib, ops._ptr, ops._stream, ops._ws, _lib.check, wsd, ops._on_device = self.saved
        ops._workspaces.clear()
        ops._workspaces.update(wsd)
        return False
