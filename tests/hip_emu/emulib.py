"""TEST INFRASTRUCTURE ONLY: ctypes access to the host-compiled kernel sources (tests/hip_emu/build.py).  The entry points
are the C-ABI of include/xrnerf_mi355.h; "device" pointers are numpy buffers."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build as _build  # noqa: E402

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
