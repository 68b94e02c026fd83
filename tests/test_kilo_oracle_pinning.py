"""Pins oracle/kilo_oracle.py (numpy restatement of the reference's KiloNeRF rendering path, BASELINE config #5) to
the reference's own in-tree PyTorch code: tests/golden/ref_kilonerf.npz (make_golden_kilo.py) and, live, on a larger
scene when /root/reference exists.  Integer work (network index, active set, per-network counts) is bit-exact; fp32
outputs hold the absolute tolerances written below."""
import os
import sys

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LAYERS = ['pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear']


@pytest.fixture(scope='module')
def K():
    import kilo_oracle
    return kilo_oracle


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_kilonerf.npz'))


def nets_from(K, g):
    return K.TinyNets([g['w.pts_linears.0'], g['w.pts_linears.1']], [g['b.pts_linears.0'], g['b.pts_linears.1']],
                      g['w.alpha_linear'], g['b.alpha_linear'], g['w.feature_linear'], g['b.feature_linear'],
                      g['w.direction_layer'], g['b.direction_layer'], g['w.rgb_linear'], g['b.rgb_linear'])


def test_assignment_is_bit_exact(K, gold):
    pts = K.get_pts(gold['rays_o'], gold['rays_d'], gold['z_vals'])
    assert np.array_equal(pts, gold['pts'])
    net, active = K.assign(pts.reshape(-1, 3), gold['gmin'], gold['gmax'], gold['fixed_res'], gold['res'],
                           gold['occupancy'], 24)
    assert np.array_equal(np.nonzero(active)[0], gold['active_samples'])
    order, counts = K.group(net, active, 24)
    assert np.array_equal(counts, gold['batch_size_per_network'])
    assert np.all(np.diff(net[order]) >= 0)


def test_local_coords_features_and_tiny_mlp(K, gold):
    n = int(gold['probe_net'])
    pts = gold['pts'].reshape(-1, 3)
    net, active = K.assign(pts, gold['gmin'], gold['gmax'], gold['fixed_res'], gold['res'], gold['occupancy'], 24)
    rows = np.nonzero(active & (net == n))[0]
    local = K.to_local(pts[rows], gold['domain_mins'][n], gold['domain_maxs'][n])
    # the reference's unstable sort may permute rows inside the segment: compare as sorted row sets
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.abs(key(local) - key(gold['probe_local'])).max() <= 1e-6
    assert np.abs(local).max() <= 1.0 + 1e-5
    dirs = np.repeat(gold['viewdirs'], gold['z_vals'].shape[1], axis=0)[rows]
    e = np.concatenate([K.fourier(local, 10), K.fourier(dirs, 4)], -1)
    assert e.shape[1] == 63 + 27
    assert np.abs(key(e) - key(gold['probe_embedded'])).max() <= 2e-4          # cos/sin at 2^9 * x: argument rounding
    raw = nets_from(K, gold).forward(n, e[:, :63], e[:, 63:])
    assert np.abs(key(raw) - key(gold['probe_raw'])).max() <= 2e-4


def test_whole_path(K, gold):
    raw, net, active, counts = K.mlp_raw(gold['rays_o'], gold['rays_d'], gold['viewdirs'], gold['z_vals'], gold['gmin'],
                                         gold['gmax'], gold['fixed_res'], gold['res'], gold['occupancy'],
                                         gold['domain_mins'], gold['domain_maxs'], nets_from(K, gold))
    assert np.array_equal((np.abs(raw).sum(-1) != 0).reshape(-1), np.isin(np.arange(raw.shape[0] * raw.shape[1]), gold['active_samples']))
    assert np.abs(raw - gold['raw']).max() <= 1e-4
    rgb, disp, acc, w = K.nerf_render(gold['raw'], gold['z_vals'], gold['rays_d'], True)
    assert np.abs(w - gold['weights']).max() <= 1e-6
    assert np.abs(rgb - gold['rgb']).max() <= 2e-6 and np.abs(acc - gold['acc']).max() <= 2e-6
    ok = np.isfinite(gold['disp'])
    assert np.array_equal(ok, np.isfinite(disp))
    assert np.abs(disp[ok] - gold['disp'][ok]).max() <= 1e-5 * max(1.0, np.abs(gold['disp'][ok]).max())


@pytest.mark.skipif(not os.path.isdir('/root/reference/xrnerf'), reason='live check needs /root/reference')
def test_live_assignment_and_render_on_a_larger_scene(K):
    import torch
    sys.path.insert(0, G)
    import ref_import
    ns = ref_import.load_kilo()
    rng = np.random.default_rng(5)
    R, S = 512, 96
    fixed, res = [9, 16, 10], [144, 256, 160]                     # the Lego grid of the reference's configs
    gmin, gmax = np.float32([-0.67, -1.2, -0.37]), np.float32([0.67, 1.2, 1.03])
    cam = rng.normal(0, 1, (R, 3)); cam = (3.0 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(np.float32)
    d = (rng.uniform(-0.6, 0.6, (R, 3)) - cam).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    z = np.sort(rng.uniform(1.5, 4.5, (R, S)), -1).astype(np.float32)
    occ = rng.uniform(0, 1, int(np.prod(res))) < 0.3
    T = torch.tensor
    pts = T(cam)[:, None, :] + T(d)[:, None, :] * T(z)[:, :, None]
    data = ns.reorder_points_and_dirs({'pts': pts, 'viewdirs': T(d), 'global_domain_min': T(gmin), 'global_domain_max': T(gmax)},
                                      [9, 16, 10], res, T(occ), 1440)
    net, active = K.assign(K.get_pts(cam, d, z).reshape(-1, 3), gmin, gmax, fixed, res, occ, 1440)
    assert np.array_equal(np.nonzero(active)[0], data['active_samples_mask'].numpy())
    assert np.array_equal(K.group(net, active, 1440)[1], data['batch_size_per_network'].numpy())
    raw = rng.normal(0, 1.5, (R, S, 4)).astype(np.float32)
    raw[~active.reshape(R, S)] = 0
    dd, ret = ns.NerfRender(white_bkgd=True, raw_noise_std=0)({'raw': T(raw), 'z_vals': T(z), 'rays_d': T(d)}, True)
    rgb, disp, acc, w = K.nerf_render(raw, z, d, True)
    assert np.abs(w - dd['weights'].numpy()).max() <= 1e-6 and np.abs(rgb - ret['rgb'].numpy()).max() <= 5e-6


@pytest.mark.skipif(not os.path.isdir('/root/reference/xrnerf'), reason='live check needs /root/reference')
def test_tiny_mlp_gradients_against_reference_autograd(K, gold):
    """oracle adjoint of MultiNetwork.forward against torch autograd through the reference's own MultiNetwork('bmm')"""
    import torch
    sys.path.insert(0, G)
    import ref_import
    ns = ref_import.load_kilo()
    torch.manual_seed(3)
    multi = ns.MultiNetwork(1, 63, 27, 4, 32, 2, None, True, 32, 'relu', linear_implementation='bmm')
    with torch.no_grad():
        for p in multi.parameters():
            p.mul_(2.0)
    mods = dict(multi.named_modules())
    g = lambda nm: (mods[nm].weight.detach().permute(0, 2, 1).contiguous().numpy(), mods[nm].bias.detach().numpy())
    (w0, b0), (w1, b1), (wa, ba), (wf, bf), (wd, bd), (wr, br) = (g(n) for n in LAYERS)
    nets = K.TinyNets([w0, w1], [b0, b1], wa, ba, wf, bf, wd, bd, wr, br)
    rng = np.random.default_rng(8)
    m = 200
    pe = K.fourier(rng.uniform(-1, 1, (m, 3)).astype(np.float32), 10)
    de = K.fourier(rng.normal(0, 0.6, (m, 3)).astype(np.float32), 4)
    d_raw = rng.normal(0, 1, (m, 4)).astype(np.float32)
    out = multi(torch.tensor(np.concatenate([pe, de], -1))[None])
    (out[0] * torch.tensor(d_raw)).sum().backward()
    got = K.tiny_backward(nets, 0, pe, de, d_raw)
    for nm in LAYERS:
        rw = mods[nm].weight.grad[0].t().numpy()
        assert np.abs(got[nm + '.weight'] - rw).max() <= 2e-5 * max(1.0, np.abs(rw).max()), nm
        rb = mods[nm].bias.grad[0].numpy()
        assert np.abs(got[nm + '.bias'] - rb).max() <= 2e-5 * max(1.0, np.abs(rb).max()), nm
    # forward of the same network, for completeness
    assert np.abs(nets.forward(0, pe, de) - out[0].detach().numpy()).max() <= 1e-5
