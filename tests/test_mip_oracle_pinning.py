"""Pins oracle/mip_oracle.py (numpy restatement of the reference's Mip-NeRF sampling / IPE / render path,
BASELINE config #3) to the reference, without a GPU:
  1. tests/golden/ref_mipnerf.npz -- outputs of the reference's OWN torch functions (make_golden_mip.py);
  2. live, when /root/reference exists: the same functions on a larger seeded input (config-sized rays).
fp32 everywhere; tolerances are absolute and written per check."""
import os
import sys

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def M():
    import mip_oracle
    return mip_oracle


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_mipnerf.npz'))


def rays(gold):
    return {k[4:]: gold[k] for k in gold.files if k.startswith('ray_')}


def test_z_vals(M, gold):
    r = rays(gold)
    n = gold['z_vals'].shape[1]
    assert np.abs(M.z_vals(r['near'], r['far'], n, False, gold['z_rand']) - gold['z_vals']).max() <= 1e-6
    assert np.abs(M.z_vals(r['near'], r['far'], n) - gold['z_det']).max() <= 1e-6
    assert np.abs(M.z_vals(r['near'], r['far'], n, True) - gold['z_lindisp']).max() <= 1e-6


def test_cast_rays(M, gold):
    r = rays(gold)
    for shape in ('cone', 'cylinder'):
        means, covs = M.cast_rays(gold['z_vals'], r['rays_o'], r['rays_d'], r['radii'], shape)
        assert np.abs(means - gold['means_' + shape]).max() <= 2e-6
        assert np.abs(covs - gold['covs_' + shape]).max() <= 1e-7 + 1e-5 * np.abs(gold['covs_' + shape]).max()


def test_embedded(M, gold):
    r = rays(gold)
    e = M.embed(gold['z_vals'], r['rays_o'], r['rays_d'], r['viewdirs'], r['radii'])
    assert e.shape == gold['embedded'].shape == (40 * 32, 123)
    # sin of arguments up to 2^15 * |x|: one ulp of the argument is ~4e-3 there, but the integrated encoding damps
    # those columns by exp(-0.5 * var * 4^k); what is left is library-level differences in sin / exp
    assert np.abs(e - gold['embedded']).max() <= 2e-6
    e2 = M.embed(gold['z_vals'], r['rays_o'], r['rays_d'], r['viewdirs'], r['radii'], 2, 7, 1, 3, False, 'cylinder')
    assert e2.shape == gold['embedded_cyl_2_7_1_3_noid'].shape
    assert np.abs(e2 - gold['embedded_cyl_2_7_1_3_noid']).max() <= 2e-6


@pytest.mark.parametrize('tag,kw', [('', dict(white_bkgd=True, density_bias=-1., rgb_padding=0.001, activation='softplus')),
                                    ('_relu_black', dict(white_bkgd=False, density_bias=0., rgb_padding=0., activation='relu'))])
def test_render_and_gradient(M, gold, tag, kw):
    r = rays(gold)
    rgb, disp, acc, w = M.render(gold['raw'], gold['z_vals'], r['rays_d'], **kw)
    assert np.abs(w - gold['render%s_weights' % tag]).max() <= 1e-6
    assert np.abs(rgb - gold['render%s_rgb' % tag]).max() <= 2e-6
    assert np.abs(acc - gold['render%s_acc' % tag]).max() <= 2e-6
    assert np.abs(disp - gold['render%s_disp' % tag]).max() <= 1e-5
    g = M.render_bwd(gold['raw'], gold['z_vals'], r['rays_d'], gold['grad_rgb'], **kw)
    ref = gold['render%s_grad_raw' % tag]
    assert np.abs(g - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_resample(M, gold):
    # tolerance: the inverse CDF amplifies a 1-ulp difference in torch.sum's (unspecified) fp32 summation order by
    # dz/dcdf, up to ~40 where only the resample padding carries the pdf; the cumulative sum itself is reproduced
    # exactly (torch's CPU cumsum accumulates in double, as the oracle does)
    z, w = gold['z_vals'], gold['render_weights']
    assert np.abs(M.resample(z, w, 0.01, gold['resample_rand']) - gold['resample_z_rand']).max() <= 1e-5
    assert np.abs(M.resample(z, w, 0.01) - gold['resample_z_det']).max() <= 1e-5
    # eps-padding branch: all-zero, tiny and half-empty weight rows without resample padding (worse conditioned:
    # nothing bounds the pdf from below next to the empty half)
    assert np.abs(M.resample(z, gold['resample_w_zero'], 0.0) - gold['resample_z_det_pad0']).max() <= 5e-5
    assert np.array_equal(M.resample(z, gold['resample_w_zero'], 0.0)[:2], gold['resample_z_det_pad0'][:2])
    r = rays(gold)
    means, covs = M.cast_rays(gold['resample_z_rand'], r['rays_o'], r['rays_d'], r['radii'], 'cone')
    assert np.abs(means - gold['resample_means']).max() <= 2e-6
    assert np.abs(covs - gold['resample_covs']).max() <= 1e-7 + 1e-5 * np.abs(gold['resample_covs']).max()
    out = M.resample(z, w, 0.01, gold['resample_rand'])
    assert np.all(np.diff(out, axis=-1) >= 0) and np.all(out >= z[:, :1]) and np.all(out <= z[:, -1:])


@pytest.mark.skipif(not os.path.isdir('/root/reference/xrnerf'), reason='live check needs /root/reference')
def test_live_against_reference_config_size(M):
    """1024 rays x 128 intervals (the config's batch), both levels, the reference's functions called live"""
    import torch
    sys.path.insert(0, G)
    import ref_import
    ns = ref_import.load_mip()
    rng = np.random.default_rng(77)
    R, S = 1024, 128
    o = rng.normal(0, 1, (R, 3)).astype(np.float32) * 0.3 + np.float32([0, 0, 4])
    d = rng.normal(0, 1, (R, 3)).astype(np.float32) * 0.2 - np.float32([0, 0, 1])
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    radii = rng.uniform(5e-4, 4e-3, (R, 1)).astype(np.float32)
    near, far = np.full((R, 1), 2, np.float32), np.full((R, 1), 6, np.float32)
    zr = rng.uniform(0, 1, (R, S + 1)).astype(np.float32)
    z = M.z_vals(near, far, S + 1, False, zr)
    T = lambda a: torch.tensor(a)
    emb = ns.MipNerfEmbedder(0, 16, 0, 4, use_viewdirs=True)
    data = {'rays_o': T(o), 'rays_d': T(d), 'viewdirs': T(vd), 'radii': T(radii), 'z_vals': T(z)}
    data = ns.mip.sample_along_rays(data, 'cone')
    ref = emb(data)['embedded'].numpy()
    assert np.abs(M.embed(z, o, d, vd, radii) - ref).max() <= 2e-6
    raw = rng.normal(0, 2, (R, S, 4)).astype(np.float32)
    render = ns.MipNerfRender(white_bkgd=True, density_bias=-1., rgb_padding=0.001, density_activation='softplus')
    rt = T(raw).requires_grad_(True)
    dd, ret = render({'raw': rt, 'z_vals': T(z), 'rays_d': T(d)}, False)
    Gr = rng.normal(0, 1, (R, 3)).astype(np.float32)
    (ret['rgb'] * T(Gr)).sum().backward()
    rgb, disp, acc, w = M.render(raw, z, d)
    assert np.abs(rgb - ret['rgb'].detach().numpy()).max() <= 5e-6
    assert np.abs(w - dd['weights'].detach().numpy()).max() <= 2e-6
    assert np.abs(disp - ret['disp'].detach().numpy()).max() <= 2e-5
    g = M.render_bwd(raw, z, d, Gr)
    assert np.abs(g - rt.grad.numpy()).max() <= 1e-5 * max(1.0, np.abs(rt.grad.numpy()).max())
    ur = rng.uniform(0, 1, (R, S + 1)).astype(np.float32)
    torch.manual_seed(3)
    ur = torch.rand([R, S + 1]).numpy()
    torch.manual_seed(3)
    data = {'rays_o': T(o), 'rays_d': T(d), 'radii': T(radii), 'z_vals': T(z), 'weights': T(w.copy())}
    znew = ns.mip.resample_along_rays(data, True, 'cone', 0.01)['z_vals'].numpy()
    assert np.abs(M.resample(z, w, 0.01, ur) - znew).max() <= 2e-5


def test_torch_statement_against_fixture(M, gold):
    """the pure-PyTorch statement (bench.py's CPU baseline for config #3) against the reference fixture, incl. one
    whole training step's losses with the reference network's weights"""
    import copy
    import json
    import torch
    r = {k: torch.tensor(v) for k, v in rays(gold).items()}
    z = torch.tensor(gold['z_vals'])
    e = M.torch_embed(z, r['rays_o'], r['rays_d'], r['viewdirs'], r['radii'])
    assert np.abs(e.numpy() - gold['embedded']).max() <= 2e-6
    rgb, dist, acc, w = M.torch_render(torch.tensor(gold['raw']), z, r['rays_d'])
    assert np.abs(rgb.numpy() - gold['render_rgb']).max() <= 2e-6 and np.abs(w.numpy() - gold['render_weights']).max() <= 1e-6
    assert np.abs(dist.numpy() - gold['render_disp']).max() <= 1e-5
    zn = M.torch_resample(z, torch.tensor(gold['render_weights']), 0.01, torch.tensor(gold['resample_rand']))
    assert np.abs(zn.numpy() - gold['resample_z_rand']).max() <= 1e-5
    zn = M.torch_resample(z, torch.tensor(gold['render_weights']), 0.01)
    assert np.abs(zn.numpy() - gold['resample_z_det']).max() <= 1e-5
    from xrnerf_amd import vanilla
    cfg = json.load(open(os.path.join(G, 'mip_model_cfg.json')))
    mcfg = copy.deepcopy(cfg['model']['mlp']); mcfg.pop('type'); mcfg.update(netdepth=4, netwidth=64, skips=[2])
    mlp = vanilla.NerfMLP(**mcfg)
    mlp.load_state_dict({k[len('net_sd.mlp.'):]: torch.tensor(gold[k]) for k in gold.files if k.startswith('net_sd.mlp.')})
    data = dict(r); data['z_vals'] = z; data['target_s'] = torch.tensor(gold['net_target'])
    loss, (lc, lf) = M.torch_train_step(mlp, data, rand=torch.tensor(gold['net_train_rand']))
    loss.backward()
    assert abs(float(lf) - float(gold['net_train_loss_fine'])) <= 2e-6
    assert abs(float(lc) - float(gold['net_train_loss_coarse'])) <= 2e-6
    assert abs(float(loss) - float(gold['net_train_loss'])) <= 2e-6
    ref = gold['net_grad.mlp.rgb_linear.weight']
    assert np.abs(mlp.rgb_linear.weight.grad.numpy() - ref).max() <= 1e-5 * max(1e-3, np.abs(ref).max())
