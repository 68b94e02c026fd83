"""Pins the CPU oracle (oracle/ngp_oracle.c) to the reference, without a GPU:
  1. the PCG32 known-answer vector obtained from the reference's own pcg32.h;
  2. the committed golden fixtures tests/golden/ref_raymarch.npz, produced by the reference's OWN
     kernels compiled for the CPU (tests/golden/make_golden.py) -- bit-exact for integer/index work,
     1e-6 for the fp32 compositor;
  3. the reference's Python pieces (ray generation, pose conversion, loss) via ref_python.npz;
  4. live, when /root/reference is present: port vs the freshly built reference kernels on larger inputs.
"""
import os

import numpy as np
import pytest

from conftest import bits

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_raymarch.npz'))


@pytest.fixture(scope='module')
def sphere():
    from xrnerf_amd import synthetic as S
    return S.sphere_density_grid(0.3)


def test_pcg32_known_answer(O):
    # pcg32 rng{9121} (raymarch_shared.h:38): state, inc, first five next_uint, first three next_float
    s, i, u5, f3 = O.pcg32_probe(9121, 0)
    assert s == 0xc8e29e7d80289fd7 and i == 3
    assert [int(x) for x in u5] == [0x2ac4520e, 0x047d20fe, 0x534368c3, 0xa882486f, 0x10fc77bb]
    assert np.allclose(f3, [0.16705811, 0.017534256, 0.325247288], rtol=0, atol=1e-9)
    # K1's per-ray jitter on the first launch: advance(i*8) then next_float
    for ray, want in ((0, 0.16705811), (1, 0.431861639), (2, 0.310610652), (4095, 0.152866364)):
        assert abs(O.pcg32_probe(9121, ray * 8)[3][0] - want) < 1e-8
    # host state after one launch (advance() = 2^32) and ray 1 of the second launch
    assert O.pcg32_host_state(1) == (0x93fbc7ba80289fd7, 3)
    s1, _, _, f = O.pcg32_probe(9121, (1 << 32) + 8)
    assert abs(f[0] - 0.227541447) < 1e-8


def test_hip_library_rng_matches(O):
    from xrnerf_amd import ops
    for c in (0, 1, 7):
        assert ops.pcg32_host_state(c) == O.pcg32_host_state(c)


def test_k11_k10_golden(O, gold, sphere):
    grid = sphere * np.float32(0.02)
    mean = float(gold['k11_mean'][0])
    assert abs(O.density_mean(grid) - mean) <= 2e-3 * mean          # serial fp32 sum vs the shim's serial sum
    bf = O.bitfield_given_mean(grid, np.float32(mean))
    per_level = [int(np.unpackbits(bf[l * 128 ** 3 // 8:(l + 1) * 128 ** 3 // 8]).sum()) for l in range(8)]
    assert per_level == [int(v) for v in gold['k11_bits_per_level']]
    crc = np.bitwise_xor.reduce(bf.view(np.uint32) * (np.arange(bf.size // 4, dtype=np.uint32) | 1))
    assert int(crc) == int(gold['k11_crc'][0])


def test_k1_k2_golden(O, gold, sphere):
    bf = O.bitfield_given_mean(sphere, np.float32(0.5))
    o, d = gold['k1_rays_o'], gold['k1_rays_d']
    for call in (0, 1):
        c, ri, ns, cnt = O.rays_sampler(o, d, bf, rng_calls=call)
        assert np.array_equal(cnt, gold['k1_call%d_counter' % call])
        assert np.array_equal(ns, gold['k1_call%d_numsteps' % call])
        assert np.array_equal(ri, gold['k1_call%d_index' % call])
        s = int(cnt[1])
        assert np.array_equal(bits(c[:s]), bits(gold['k1_call%d_coords' % call]))
    cap = int(gold['k2_cap'][0])
    co, nc, rc, sc = O.compacted_coord(gold['k1_call0_coords'], gold['k1_call0_numsteps'], cap)
    assert np.array_equal(nc, gold['k2_numsteps']) and [int(rc[0]), int(sc[0])] == [int(v) for v in gold['k2_counters']]
    assert np.array_equal(bits(co[:cap]), bits(gold['k2_coords']))


def test_compositor_golden(O, gold):
    c, ns, nc = gold['k1_call0_coords'], gold['k1_call0_numsteps'], gold['k2_numsteps']
    raw, bg, grad = gold['k3_raw'], gold['k3_bg'], gold['k4_grad']
    rgb = O.calc_rgb_forward(raw, c, ns, nc, bg, 2, 3)
    assert np.abs(rgb - gold['k3_rgb']).max() <= 1e-6
    for name, mean in (('k4_draw_mean_small', 0.001), ('k4_draw_mean_large', 0.5)):
        d = O.calc_rgb_backward(raw, nc, c, grad, gold['k3_rgb'], mean, 2, 3)
        assert np.abs(d - gold[name]).max() <= 1e-6 * max(1.0, np.abs(gold[name]).max())
    assert not np.array_equal(gold['k4_draw_mean_small'], gold['k4_draw_mean_large'])   # the L1 switch is exercised
    r5, a5 = O.calc_rgb_inference(raw, c, ns, [0.2, 0.5, 0.9], 2, 3)
    assert np.abs(r5 - gold['k5_rgb']).max() <= 1e-6 and np.abs(a5 - gold['k5_alpha']).max() <= 1e-6


def test_grid_upkeep_golden(O, gold, sphere):
    from xrnerf_amd import synthetic as S
    g6 = (sphere * np.float32(0.02)).astype(np.float32)
    for call, (n, step, casc, thr) in enumerate([(4096, 0, 0, -0.01), (4096, 5, 0, 0.01)]):
        p, i = O.generate_grid_samples(g6, step, n, casc, thr, rng_calls=call)
        assert np.array_equal(i, gold['k6_call%d_idx' % call]) and np.array_equal(bits(p), bits(gold['k6_call%d_pos' % call]))
    focal = np.full((6, 2), S.LEGO_FOCAL, np.float32)
    m = O.mark_untrained(focal, gold['k7_poses'], 2 * S.G3, (800, 800))
    assert int((m < 0).sum()) == int(gold['k7_count_neg'][0]) and np.array_equal(m[::4099], gold['k7_sample'])
    tmp = O.splat(gold['k8_mlp'], gold['k8_idx'], np.zeros(8 * S.G3, np.float32))
    assert np.array_equal(bits(tmp[gold['k8_idx']]), bits(gold['k8_vals']))
    g9 = g6.copy(); g9[::7] = -1.0
    e = O.ema(tmp, g9)
    assert np.array_equal(bits(e[gold['k8_idx']]), bits(gold['k9_vals']))
    assert abs(e.astype(np.float64).sum() - float(gold['k9_sum'][0])) < 1e-9


def test_reference_python_pieces(O):
    """ray generation / pose conversion / loss vs outputs of the reference's own Python functions"""
    from xrnerf_amd import synthetic as S
    import torch
    from xrnerf_amd.networks import HuberLoss, img2mse, mse2psnr
    from xrnerf_amd.mlps import get_per_level_scale
    py = np.load(os.path.join(G, 'ref_python.npz'))
    ngp = S.poses_nerf2ngp(py['poses44'])
    assert np.array_equal(ngp, py['poses_ngp'])
    f = np.float32(S.LEGO_FOCAL)
    o, d = O.gen_rays(ngp[0], 800, 800, f, f, 400.0, 400.0)
    sel = py['rays_sel']
    flat = sel[:, 0] * 800 + sel[:, 1]
    # the reference computes in float64 under numpy >= 2; fp32 result within 1 ulp-ish
    assert np.abs(o[flat] - py['rays_o']).max() <= 1e-7 and np.abs(d[flat] - py['rays_d']).max() <= 2e-7
    so, sd = S.camera_rays(ngp[0], 800, 800, S.LEGO_FOCAL, flat)
    assert np.abs(sd - py['rays_d']).max() <= 2e-7
    x, y = py['loss_x'], py['loss_y']
    loss, grad = O.huber_loss_grad(x, y, 0.1, 1.0)
    assert abs(loss - py['huber_sum'][0]) <= 1e-5 * py['huber_sum'][0]
    assert abs(float(HuberLoss(torch.tensor(x), torch.tensor(y), 0.1, 'sum')) - py['huber_sum'][0]) <= 1e-4
    assert abs(float(mse2psnr(img2mse(torch.tensor(x), torch.tensor(y)))) - py['psnr'][0]) <= 1e-4
    assert get_per_level_scale(1) == float(py['per_level_scale'][0])
    xt = torch.tensor(x, requires_grad=True)
    HuberLoss(xt, torch.tensor(y), 0.1, 'sum').backward()
    assert np.abs(xt.grad.numpy() - grad).max() <= 1e-6


def test_tcnn_half_restatement_vs_torch(O):
    """the un-pinned half: the C restatement of hash grid / SH / MLP agrees with an independent
    numpy/torch statement of the same published algorithm, forward and backward"""
    import torch
    from xrnerf_amd import synthetic as S
    meta = O.GridMeta()
    assert meta.n_params == 12196240 and list(meta.resolution[:5]) == [16, 23, 31, 43, 59]
    assert list(meta.offset[:6]) == [0, 4096, 16264, 46056, 125568, 330952]
    rng = np.random.default_rng(0)
    n = 300
    table = S.hash_table(meta.n_params, scale=0.5)
    x = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    enc = O.hashgrid_fwd(table, x, meta)
    # independent numpy evaluation of level 0 (dense) and level 15 (hashed)
    for l in (0, 15):
        sc, res, off, hs = meta.scale[l], int(meta.resolution[l]), int(meta.offset[l]), int(meta.offset[l + 1] - meta.offset[l])
        p = x * sc + np.float32(0.5); g = np.floor(p).astype(np.uint32); w = p - np.floor(p)
        out = np.zeros((n, 2), np.float32)
        for c in range(8):
            q = g + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1], np.uint32)
            wt = np.prod(np.where([c & 1, (c >> 1) & 1, (c >> 2) & 1], w, 1 - w), axis=1).astype(np.float32)
            if res ** 3 > hs:
                idx = (q[:, 0] * np.uint32(1)) ^ (q[:, 1] * np.uint32(2654435761)) ^ (q[:, 2] * np.uint32(805459861))
            else:
                idx = q[:, 0] + q[:, 1] * np.uint32(res) + q[:, 2] * np.uint32(res * res)
            idx = idx % np.uint32(hs)
            out += wt[:, None] * table.reshape(-1, 2)[off + idx.astype(np.int64)]
        assert np.abs(out - enc[:, 2 * l:2 * l + 2]).max() <= 1e-6
    wd, wc = S.mlp_weights(32, 64, 1, 16, 4), S.mlp_weights(32, 64, 2, 16, 5)
    dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    draw = rng.normal(0, 1, (n, 4)).astype(np.float32)
    raw = O.nerf_mlp_fwd(table, wd, wc, x, dirs, meta)
    Wd, Wc = torch.tensor(wd, requires_grad=True), torch.tensor(wc, requires_grad=True)
    e = torch.tensor(enc, requires_grad=True)
    h = torch.relu(e @ Wd[:2048].view(64, 32).T); dout = h @ Wd[2048:].view(16, 64).T
    cin = torch.cat([dout[:, 1:], torch.tensor(O.sh4(dirs)), torch.ones(n, 1)], 1)
    c = torch.relu(cin @ Wc[:2048].view(64, 32).T); c = torch.relu(c @ Wc[2048:6144].view(64, 64).T)
    rt = torch.cat([(c @ Wc[6144:].view(16, 64).T)[:, :3], dout[:, :1]], 1)
    assert np.abs(rt.detach().numpy() - raw).max() <= 1e-5
    (rt * torch.tensor(draw)).sum().backward()
    gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, x, dirs, draw, meta)
    assert np.abs(gd - Wd.grad.numpy()).max() <= 1e-4 and np.abs(gc - Wc.grad.numpy()).max() <= 1e-4
    assert np.abs(O.hashgrid_bwd(x, e.grad.numpy(), meta) - gt).max() <= 1e-5


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference tree only exists in the build container')
def test_port_vs_reference_kernels_live(O, lego):
    """larger inputs than the fixtures, against the reference kernels built on the spot"""
    from xrnerf_amd import synthetic as S
    assert O.have_ref()
    o, d, ids = S.training_rays(lego['poses'], 4096, seed=13)
    for call in (0, 2):
        a = O.rays_sampler(o, d, lego['bitfield'], rng_calls=call)
        b = O.rays_sampler(o, d, lego['bitfield'], rng_calls=call, backend='ref', img_ids=ids * 0)
        assert all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
        s = int(a[3][1])
        assert np.array_equal(bits(a[0][:s]), bits(b[0][:s]))
    total = int(a[3][1])
    for cap in (total // 2, 7):       # sample-buffer overflow (ray_sampler.cu:76-82)
        p = O.rays_sampler(o, d, lego['bitfield'], max_samples=cap)
        q = O.rays_sampler(o, d, lego['bitfield'], max_samples=cap, backend='ref')
        assert np.array_equal(p[2], q[2]) and np.array_equal(p[1], q[1]) and np.array_equal(p[3], q[3])
    assert np.array_equal(O.bitfield_given_mean(lego['grid'], lego['mean']),
                          O.bitfield_given_mean(lego['grid'], lego['mean'], backend='ref'))


def test_k1_multi_cascade_golden(O):
    """aabb_scale = 16 (five active cascades): mip selection from position and step size,
    coarse-cascade voxel skipping, rays entering the box from outside -- port == the reference's ray_sampler.cu"""
    import sys
    sys.path.insert(0, G)
    from make_golden_cascades import cascade_inputs
    g = np.load(os.path.join(G, 'ref_raymarch_cascades.npz'))
    grid, o, d, aabb = cascade_inputs()
    bf = O.bitfield_given_mean(grid, np.float32(0.5))
    c, ri, ns, cnt = O.rays_sampler(o, d, bf, aabb=aabb, rng_calls=0)
    assert np.array_equal(cnt, g['counter']) and np.array_equal(ns, g['numsteps']) and np.array_equal(ri, g['index'])
    assert np.array_equal(bits(c[:int(cnt[1])]), bits(g['coords']))
    # the fixture really exercises the coarse cascades: warped dt spans more than the two values of a unit cube
    assert len(np.unique(g['coords'][:, 3])) > 1000 and int(g['numsteps'][:, 0].max()) > 40


@pytest.mark.skipif(not __import__('oracle').have_ref(), reason='needs oracle/_ref (the reference kernels compiled for the CPU)')
def test_k6_unbounded_boxes_bit_exact_against_the_reference_kernels(O, lego):
    """BASELINE config #4's setting: K6 (generate_grid_samples_nerf_nonuniform) with aabb = [-7.5, 8.5] (aabb_scale 16, five
    cascades) and an intermediate box -- the restatement against the reference's own kernel, positions bit for bit (they are
    warped into the box: inside [0, 1])."""
    rng = np.random.default_rng(3)
    grid = (lego['grid'] * rng.uniform(0.0, 0.05, lego['grid'].shape)).astype(np.float32)
    grid[rng.uniform(0, 1, grid.shape) < 0.1] = -1.0
    for aabb in ((-7.5, 8.5), (-1.5, 2.5)):
        for casc, thr, calls in ((5, 0.01, 4), (5, -0.01, 0), (3, 0.01, 2)):
            rp, ri = O.generate_grid_samples(grid, 3, 50000, casc - 1, thr, aabb=aabb, rng_calls=calls, backend='ref')
            pp, pi = O.generate_grid_samples(grid, 3, 50000, casc - 1, thr, aabb=aabb, rng_calls=calls, backend='port')
            assert np.array_equal(ri, pi) and np.array_equal(rp.view(np.uint32), pp.view(np.uint32)), (aabb, casc, thr)
    assert 0.0 <= rp.min() and rp.max() <= 1.0
