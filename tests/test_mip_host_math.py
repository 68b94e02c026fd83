"""The per-element arithmetic of the Mip-NeRF kernels (xrnerf_amd/csrc/xr_mip_math.h -- the header xr_mip.hip
includes) compiled for the HOST with g++ and held against the numpy oracle / the reference fixture.  Runs without a
GPU; it checks formulas and column order, not the kernels' indexing (tests/test_gpu_mip.py does that on the device).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')
SRC = os.path.join(ROOT, 'tests', 'host_harness', 'mip_math_host.cpp')
SO = os.path.join(ROOT, 'tests', 'host_harness', 'libmip_math_host.so')


@pytest.fixture(scope='module')
def H():
    hdr = os.path.join(ROOT, 'xrnerf_amd', 'csrc', 'xr_mip_math.h')
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', SRC, '-o', SO])
    return C.CDLL(SO)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_linspace_is_torchs(H):
    for n, end in ((129, 1.0), (33, 1.0), (129, float(np.float32(1 - np.finfo(np.float32).eps))), (2, 1.0), (64, 1.0)):
        out = np.zeros(n, np.float32)
        H.hm_linspace(C.c_float(0.0), C.c_float(end), n, p(out))
        assert np.array_equal(out, torch.linspace(0., end, n).numpy())


def test_zvals_and_encoding_against_reference_fixture(H):
    import mip_oracle as M
    g = np.load(os.path.join(G, 'ref_mipnerf.npz'))
    R, n_z = g['z_vals'].shape
    z = np.zeros((R, n_z), np.float32)
    near, far = np.ascontiguousarray(g['ray_near'][:, 0]), np.ascontiguousarray(g['ray_far'][:, 0])
    H.hm_zvals(p(near), p(far), R, n_z, 0, p(z))
    assert np.abs(z - g['z_det']).max() <= 1e-6
    H.hm_zvals(p(near), p(far), R, n_z, 1, p(z))
    assert np.abs(z - g['z_lindisp']).max() <= 1e-6
    o, d, vd = (np.ascontiguousarray(g['ray_' + k]) for k in ('rays_o', 'rays_d', 'viewdirs'))
    radii = np.ascontiguousarray(g['ray_radii'][:, 0])
    zz = np.ascontiguousarray(g['z_vals'])
    for args, key in (((0, 16, 0, 4, 1, 0), 'embedded'), ((2, 7, 1, 3, 0, 1), 'embedded_cyl_2_7_1_3_noid')):
        ch = g[key].shape[1]
        out = np.zeros((R * (n_z - 1), ch), np.float32)
        H.hm_encode(p(o), p(d), p(vd), p(radii), p(zz), R, n_z, *args, ch, p(out))
        assert np.abs(out - g[key]).max() <= 2e-6, key
        e = M.embed(zz, o, d, vd, g['ray_radii'], args[0], args[1], args[2], args[3], bool(args[4]),
                    'cylinder' if args[5] else 'cone')
        assert np.abs(out - e).max() <= 2e-6


def test_activations(H):
    x = np.concatenate([np.linspace(-60, 60, 4001), [19.999, 20.0, 20.001, 0.0]]).astype(np.float32)
    act, dact, sig = (np.zeros_like(x) for _ in range(3))
    xt = torch.tensor(x, requires_grad=True)
    sp = torch.nn.functional.softplus(xt)
    sp.sum().backward()
    H.hm_density(p(x), x.size, 0, p(act), p(dact), p(sig))
    assert np.abs(act - sp.detach().numpy()).max() <= 1e-6 * 60
    assert np.abs(dact - xt.grad.numpy()).max() <= 1e-6
    assert np.abs(sig - torch.sigmoid(torch.tensor(x)).numpy()).max() <= 1e-6
    H.hm_density(p(x), x.size, 1, p(act), p(dact), p(sig))
    assert np.array_equal(act, np.maximum(x, 0)) and np.array_equal(dact, (x > 0).astype(np.float32))
