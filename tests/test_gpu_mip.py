"""Mip-NeRF stages (BASELINE config #3) on the MI355X, through the C-ABI, against
  * tests/golden/ref_mipnerf.npz -- outputs of the reference's OWN torch code (make_golden_mip.py), and
  * the numpy oracle (oracle/mip_oracle.py, pinned to the same reference) at the config's sizes and at ragged ones.
Tolerances (absolute, fp32): north_star's 1e-4 is the ceiling; each check states the tighter one it holds."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def M():
    import mip_oracle
    return mip_oracle


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_mipnerf.npz'))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rays(gold, dev):
    return {k[4:]: T(gold[k], dev) for k in gold.files if k.startswith('ray_')}


def test_zvals_fixture(dev, gold):
    from xrnerf_amd import ops
    r = rays(gold, dev)
    n = gold['z_vals'].shape[1]
    z = ops.mip_zvals(r['near'], r['far'], n, False, T(gold['z_rand'], dev)).cpu().numpy()
    assert np.abs(z - gold['z_vals']).max() <= 1e-6
    assert np.abs(ops.mip_zvals(r['near'], r['far'], n).cpu().numpy() - gold['z_det']).max() <= 1e-6
    assert np.abs(ops.mip_zvals(r['near'], r['far'], n, True).cpu().numpy() - gold['z_lindisp']).max() <= 1e-6


def test_encode_fixture(dev, gold):
    from xrnerf_amd import ops
    r = rays(gold, dev)
    z = T(gold['z_vals'], dev)
    e = ops.mip_encode(r['rays_o'], r['rays_d'], r['viewdirs'], r['radii'], z, 0, 16, 0, 4, True, 'cone')
    assert tuple(e.shape) == gold['embedded'].shape
    assert np.abs(e.cpu().numpy() - gold['embedded']).max() <= 5e-6
    e2 = ops.mip_encode(r['rays_o'], r['rays_d'], r['viewdirs'], r['radii'], z, 2, 7, 1, 3, False, 'cylinder')
    assert np.abs(e2.cpu().numpy() - gold['embedded_cyl_2_7_1_3_noid']).max() <= 5e-6
    # the (means, covs) entry point: the reference's own gaussians in, same rows out
    e3 = ops.mip_encode_gaussians(T(gold['means_cone'], dev), T(gold['covs_cone'], dev), r['viewdirs'], 0, 16, 0, 4, True)
    assert np.abs(e3.cpu().numpy() - gold['embedded']).max() <= 5e-6


@pytest.mark.parametrize('tag,kw', [('', dict(density_bias=-1., rgb_padding=0.001, white_bkgd=True, density_activation='softplus')),
                                    ('_relu_black', dict(density_bias=0., rgb_padding=0., white_bkgd=False, density_activation='relu'))])
def test_render_fixture(dev, gold, tag, kw):
    from xrnerf_amd import ops
    r = rays(gold, dev)
    raw, z = T(gold['raw'], dev), T(gold['z_vals'], dev)
    rgb, dist, acc, w = ops.mip_render_forward(raw, z, r['rays_d'], **kw)
    assert np.abs(w.cpu().numpy() - gold['render%s_weights' % tag]).max() <= 2e-6
    assert np.abs(rgb.cpu().numpy() - gold['render%s_rgb' % tag]).max() <= 5e-6
    assert np.abs(acc.cpu().numpy() - gold['render%s_acc' % tag]).max() <= 5e-6
    assert np.abs(dist.cpu().numpy() - gold['render%s_disp' % tag]).max() <= 2e-5
    g = ops.mip_render_backward(raw, z, r['rays_d'], T(gold['grad_rgb'], dev), **kw).cpu().numpy()
    ref = gold['render%s_grad_raw' % tag]
    assert np.abs(g - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_resample_fixture(dev, gold):
    from xrnerf_amd import ops
    z, w = T(gold['z_vals'], dev), T(gold['render_weights'], dev)
    # conditioning: see tests/test_mip_oracle_pinning.py::test_resample
    out = ops.mip_resample(z, w, 0.01, T(gold['resample_rand'], dev)).cpu().numpy()
    assert np.abs(out - gold['resample_z_rand']).max() <= 2e-5
    assert np.all(np.diff(out, axis=-1) >= 0)
    assert np.abs(ops.mip_resample(z, w, 0.01).cpu().numpy() - gold['resample_z_det']).max() <= 2e-5
    out0 = ops.mip_resample(z, T(gold['resample_w_zero'], dev), 0.0).cpu().numpy()
    assert np.abs(out0 - gold['resample_z_det_pad0']).max() <= 1e-4
    assert np.abs(out0[:2] - gold['resample_z_det_pad0'][:2]).max() <= 1e-6       # all-zero / tiny rows: eps padding


@pytest.mark.parametrize('R,S', [(1024, 128), (1023, 200), (5, 7), (1, 1), (257, 64)])
def test_all_stages_against_oracle(dev, M, R, S):
    """config size (1024 x 128), ragged sizes (partial waves / partial sweeps / partial tiles), single interval"""
    from xrnerf_amd import ops
    rng = np.random.default_rng(1000 + R + S)
    o = (rng.normal(0, 1, (R, 3)) * 0.3 + [0, 0, 4]).astype(np.float32)
    d = (rng.normal(0, 1, (R, 3)) * 0.2 - [0, 0, 1]).astype(np.float32)
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    radii = rng.uniform(5e-4, 4e-3, (R, 1)).astype(np.float32)
    near, far = np.full((R, 1), 2, np.float32), np.full((R, 1), 6, np.float32)
    zr = rng.uniform(0, 1, (R, S + 1)).astype(np.float32)
    z = M.z_vals(near, far, S + 1, False, zr)
    zg = ops.mip_zvals(T(near, dev), T(far, dev), S + 1, False, T(zr, dev))
    assert np.abs(zg.cpu().numpy() - z).max() <= 1e-6
    zt = T(z, dev)
    e = ops.mip_encode(T(o, dev), T(d, dev), T(vd, dev), T(radii, dev), zt, 0, 16, 0, 4, True, 'cone').cpu().numpy()
    assert np.abs(e - M.embed(z, o, d, vd, radii)).max() <= 5e-6
    raw = rng.normal(0, 2, (R, S, 4)).astype(np.float32)
    rgb, dist, acc, w = ops.mip_render_forward(T(raw, dev), zt, T(d, dev), -1., 0.001, True, 'softplus')
    orgb, odist, oacc, ow = M.render(raw, z, d)
    assert np.abs(w.cpu().numpy() - ow).max() <= 2e-6
    assert np.abs(rgb.cpu().numpy() - orgb).max() <= 1e-5
    assert np.abs(acc.cpu().numpy() - oacc).max() <= 1e-5
    assert np.abs(dist.cpu().numpy() - odist).max() <= 5e-5
    Gr = rng.normal(0, 1, (R, 3)).astype(np.float32)
    g = ops.mip_render_backward(T(raw, dev), zt, T(d, dev), T(Gr, dev), -1., 0.001, True, 'softplus').cpu().numpy()
    og = M.render_bwd(raw, z, d, Gr)
    assert np.abs(g - og).max() <= 1e-5 * max(1.0, np.abs(og).max())
    ur = rng.uniform(0, 1, (R, S + 1)).astype(np.float32)
    zn = ops.mip_resample(zt, w, 0.01, T(ur, dev)).cpu().numpy()
    assert np.abs(zn - M.resample(z, ow, 0.01, ur)).max() <= 5e-5
    assert np.all(np.diff(zn, axis=-1) >= 0) and np.all(zn >= z[:, :1] - 1e-6) and np.all(zn <= z[:, -1:] + 1e-6)
    zn = ops.mip_resample(zt, w, 0.01).cpu().numpy()
    assert np.abs(zn - M.resample(z, ow, 0.01)).max() <= 5e-5


def test_render_backward_is_the_adjoint(dev):
    """size-independent property at the config's size: <J v, g> == <v, J^T g> by central differences in fp64-ish
    (directional derivative of sum(G * rgb) along a random direction vs the kernel's gradient)"""
    from xrnerf_amd import ops
    torch.manual_seed(0)
    R, S = 1024, 128
    raw = (torch.randn(R, S, 4, device=dev) * 1.5).contiguous()
    z = torch.sort(torch.rand(R, S + 1, device=dev) * 4 + 2, -1)[0].contiguous()
    d = torch.randn(R, 3, device=dev)
    Gr = torch.randn(R, 3, device=dev)
    v = torch.randn_like(raw)
    g = ops.mip_render_backward(raw, z, d, Gr, -1., 0.001, True, 'softplus')
    eps = 1e-2
    fp = ops.mip_render_forward(raw + eps * v, z, d, -1., 0.001, True, 'softplus')[0]
    fm = ops.mip_render_forward(raw - eps * v, z, d, -1., 0.001, True, 'softplus')[0]
    num = ((fp - fm).double() * Gr.double()).sum() / (2 * eps)
    ana = (g.double() * v.double()).sum()
    assert abs(float(num - ana)) <= 2e-3 * max(1.0, abs(float(ana)))


def test_edge_cases_and_validation(dev):
    from xrnerf_amd import _lib, ops
    z0 = torch.zeros((0, 9), device=dev)
    assert ops.mip_resample(z0, torch.zeros((0, 8), device=dev), 0.01).shape == (0, 9)
    e = ops.mip_encode(torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev),
                       torch.zeros((0, 1), device=dev), z0, 0, 16, 0, 4)
    assert e.shape == (0, 123)
    with pytest.raises(_lib.XrError):
        ops.mip_zvals(torch.zeros(4), torch.ones(4), 9)                       # host tensors: no CPU fallback
    with pytest.raises(_lib.XrError):
        ops.mip_resample(torch.zeros((2, 4000), device=dev), torch.zeros((2, 3999), device=dev), 0.01)   # n_z > 2048
    # an empty ray renders the background and the far-clamped distance
    raw = torch.full((3, 16, 4), -50.0, device=dev)
    z = torch.linspace(2, 6, 17, device=dev).expand(3, 17).contiguous()
    rgb, dist, acc, w = ops.mip_render_forward(raw, z, torch.ones(3, 3, device=dev), -1., 0.001, True, 'softplus')
    assert float(acc.abs().max()) == 0.0 and float((rgb - 1).abs().max()) == 0.0 and float((dist - 6).abs().max()) == 0.0


def test_network_against_reference_fixture(dev, gold):
    """MipNerfNetwork behind the registry (the reference config's model dict, MLP shrunk to 4 x 64 like the fixture's)
    with the reference network's state dict: test-mode outputs, one training step's losses and parameter gradients"""
    import copy
    import xrnerf_amd
    cfg = json.load(open(os.path.join(G, 'mip_model_cfg.json')))
    model = copy.deepcopy(cfg['model'])
    model['mlp'].update(netdepth=4, netwidth=64, skips=[2])
    net = xrnerf_amd.build_network(model).to(dev)
    sd = {k[len('net_sd.'):]: torch.tensor(gold[k]) for k in gold.files if k.startswith('net_sd.')}
    net.load_state_dict(sd, strict=True)                 # same parameter names as the reference
    r = rays(gold, dev)
    data = dict(r); data['z_vals'] = T(gold['z_det'], dev)
    with torch.no_grad():
        ret = net.forward(data, is_test=True)
    for k in ('rgb', 'coarse_rgb', 'acc', 'coarse_acc'):
        assert np.abs(ret[k].cpu().numpy() - gold['net_test_' + k]).max() <= 1e-4, k
    for k in ('disp', 'coarse_disp'):
        assert np.abs(ret[k].cpu().numpy() - gold['net_test_' + k]).max() <= 2e-4, k
    # training step with the reference's resampler draws
    from xrnerf_amd import mip
    data = {k: v.clone() for k, v in r.items()}
    data['z_vals'], data['target_s'] = T(gold['z_vals'], dev), T(gold['net_target'], dev)
    data = mip.sample_along_rays(data, 'cone')
    data, coarse = net.render(net.mlp(data), False)
    data = mip.resample_along_rays(data, True, 'cone', 0.01, rand=T(gold['net_train_rand'], dev))
    data, fine = net.render(net.mlp(data), False)
    mask = torch.broadcast_to(data['lossmult'], fine['rgb'].shape)
    lf = (mask * (fine['rgb'] - data['target_s']) ** 2).sum() / mask.sum()
    lc = (mask * (coarse['rgb'] - data['target_s']) ** 2).sum() / mask.sum()
    loss = lf + 0.1 * lc
    loss.backward()
    assert abs(float(lf) - float(gold['net_train_loss_fine'])) <= 1e-5
    assert abs(float(lc) - float(gold['net_train_loss_coarse'])) <= 1e-5
    assert abs(float(loss) - float(gold['net_train_loss'])) <= 1e-5
    for name in ('mlp.rgb_linear.weight', 'mlp.pts_linears.0.weight', 'mlp.alpha_linear.bias'):
        ref = gold['net_grad.' + name]
        got = dict(net.named_parameters())[name].grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-4 * max(1e-3, np.abs(ref).max()) + 1e-7, name


def test_config3_trains(dev):
    """the reference's config #3 model dict unchanged (8 x 256 MLP, 128 + 128 samples, 1024 rays): a few Adam steps
    through train_step reduce the loss"""
    import xrnerf_amd
    from xrnerf_amd import mip
    cfg = json.load(open(os.path.join(G, 'mip_model_cfg.json')))
    torch.manual_seed(0)
    net = xrnerf_amd.build_network(cfg['model']).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=cfg['optimizer']['lr'])
    rays_ = mip.synthetic_multiscale_rays(cfg['N_rand_per_sampler'], dev, seed=1)
    rays_['target_s'] = torch.sigmoid(rays_['rays_o'] * 2.0)          # a learnable (view-dependent) target
    losses = []
    for it in range(12):
        data = {k: v[None] for k, v in rays_.items()}                 # DataLoader's batch dimension
        data = {k: v for k, v in data.items()}
        d0 = {k: v[0] for k, v in data.items()}
        d0 = mip.get_z_vals(d0, cfg['num_samples'] + 1, randomized=True)
        data['z_vals'] = d0['z_vals'][None]
        out = net.train_step(data, opt)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        losses.append(float(out['log_vars']['loss']))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
