"""The Mip-NeRF kernels of xrnerf_amd/csrc/xr_mip.hip -- the SAME source the GPU library is built from -- compiled for the
host and executed lane by lane by the HIP-on-CPU shim (tests/hip_emu: fibers, wave64 collectives, LDS as shared memory),
against the numpy oracle and the reference fixture.  Runs without a GPU: it checks indexing, wave scans, LDS staging,
ragged sizes; tests/test_gpu_mip.py checks the same on the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hip_emu'))
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def E():
    import emulib
    return emulib


@pytest.fixture(scope='module')
def M():
    import mip_oracle
    return mip_oracle


def run_all(E, M, R, S, seed):
    L = E.lib('xr_mip')
    rng = np.random.default_rng(seed)
    o = E.f32(rng.normal(0, 1, (R, 3)) * 0.3 + [0, 0, 4])
    d = E.f32(rng.normal(0, 1, (R, 3)) * 0.2 - [0, 0, 1])
    vd = E.f32(d / np.linalg.norm(d, axis=-1, keepdims=True))
    radii = E.f32(rng.uniform(5e-4, 4e-3, (R, 1)))
    near, far = np.full(R, 2, np.float32), np.full(R, 6, np.float32)
    zr = E.f32(rng.uniform(0, 1, (R, S + 1)))
    z = np.zeros((R, S + 1), np.float32)
    E.check(L.xr_mip_zvals(E.p(near), E.p(far), R, S + 1, 0, E.p(zr), E.p(z), None), L)
    zo = M.z_vals(near[:, None], far[:, None], S + 1, False, zr)
    assert np.abs(z - zo).max() <= 1e-6
    out = np.zeros((R * S, 123), np.float32)
    E.check(L.xr_mip_encode(E.p(o), E.p(d), E.p(vd), E.p(radii), E.p(z), R, S + 1, 0, 16, 0, 4, 1, 0, E.p(out), 123, None), L)
    assert np.abs(out - M.embed(z, o, d, vd, radii)).max() <= 2e-6
    raw = E.aligned((R, S, 4), fill=rng.normal(0, 2, (R, S, 4)))
    rgb, dist, acc, w = np.zeros((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32), np.zeros((R, S), np.float32)
    E.check(L.xr_mip_render_forward(E.p(raw), E.p(z), E.p(d), R, S + 1, C.c_float(-1.0), C.c_float(0.001), 1, 0, E.p(rgb), E.p(dist),
                                    E.p(acc), E.p(w), None), L)
    orgb, odist, oacc, ow = M.render(raw, z, d)
    assert np.abs(w - ow).max() <= 2e-6 and np.abs(rgb - orgb).max() <= 1e-5 and np.abs(acc - oacc).max() <= 1e-5
    assert np.abs(dist - odist).max() <= 5e-5
    g = E.f32(rng.normal(0, 1, (R, 3)))
    graw = E.aligned((R, S, 4))
    E.check(L.xr_mip_render_backward(E.p(raw), E.p(z), E.p(d), E.p(g), R, S + 1, C.c_float(-1.0), C.c_float(0.001), 1, 0, E.p(graw), None), L)
    og = M.render_bwd(raw, z, d, g)
    assert np.abs(graw - og).max() <= 1e-5 * max(1.0, np.abs(og).max())
    ur = E.f32(rng.uniform(0, 1, (R, S + 1)))
    zn = np.zeros_like(z)
    E.check(L.xr_mip_resample(E.p(z), E.p(w), E.p(ur), C.c_float(0.01), R, S + 1, E.p(zn), None), L)
    assert np.abs(zn - M.resample(z, ow, 0.01, ur)).max() <= 5e-5
    E.check(L.xr_mip_resample(E.p(z), E.p(w), None, C.c_float(0.01), R, S + 1, E.p(zn), None), L)
    assert np.abs(zn - M.resample(z, ow, 0.01)).max() <= 5e-5


@pytest.mark.parametrize('R,S', [(37, 128), (9, 200), (5, 7), (1, 1), (130, 64)])
def test_mip_kernels_on_the_host(E, M, R, S):
    run_all(E, M, R, S, 100 + R + S)


def test_mip_kernels_on_the_reference_fixture(E):
    gold = np.load(os.path.join(G, 'ref_mipnerf.npz'))
    L = E.lib('xr_mip')
    R, n_z = gold['z_vals'].shape
    z = E.f32(gold['z_vals'])
    out = np.zeros((R * (n_z - 1), 123), np.float32)
    E.check(L.xr_mip_encode(E.p(E.f32(gold['ray_rays_o'])), E.p(E.f32(gold['ray_rays_d'])), E.p(E.f32(gold['ray_viewdirs'])),
                            E.p(E.f32(gold['ray_radii'])), E.p(z), R, n_z, 0, 16, 0, 4, 1, 0, E.p(out), 123, None), L)
    assert np.abs(out - gold['embedded']).max() <= 2e-6
    zn = np.zeros_like(z)
    E.check(L.xr_mip_resample(E.p(z), E.p(E.f32(gold['render_weights'])), E.p(E.f32(gold['resample_rand'])), C.c_float(0.01), R, n_z,
                              E.p(zn), None), L)
    assert np.abs(zn - gold['resample_z_rand']).max() <= 2e-5
