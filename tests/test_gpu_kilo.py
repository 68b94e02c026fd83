"""KiloNeRF rendering path (BASELINE config #5) on the MI355X, through the C-ABI, against
  * tests/golden/ref_kilonerf.npz -- outputs of the reference's OWN in-tree PyTorch code (make_golden_kilo.py), and
  * the numpy oracle (oracle/kilo_oracle.py, pinned to the same code) on the Lego grid (1440 networks, 384 samples/ray).
Integer work (network assignment, active set, per-network counts) is bit-exact; raw / colours within 1e-4 abs fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LAYERS = ['pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear']
EMB = dict(type='KiloNerfFourierEmbedder', num_networks=1, input_ch=3, multires=10, multires_dirs=4)


@pytest.fixture(scope='module')
def K():
    import kilo_oracle
    return kilo_oracle


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(G, 'ref_kilonerf.npz'))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def mlp_from_gold(gold, dev):
    from xrnerf_amd import kilo
    sd = {}
    for nm in LAYERS:
        sd[nm + '.weight'], sd[nm + '.bias'] = torch.tensor(gold['w.' + nm]), torch.tensor(gold['b.' + nm])
    res = [int(v) for v in gold['res']]
    return kilo.KiloNerfMLP.from_arrays(res, gold['occupancy'], gold['domain_mins'], gold['domain_maxs'], sd, EMB).to(dev)


def oracle_nets(K, sd):
    g = lambda k: sd[k].detach().cpu().numpy()
    n_hidden = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('pts_linears.'))
    return K.TinyNets([g('pts_linears.%d.weight' % l) for l in range(n_hidden)], [g('pts_linears.%d.bias' % l) for l in range(n_hidden)],
                      g('alpha_linear.weight'), g('alpha_linear.bias'), g('feature_linear.weight'), g('feature_linear.bias'),
                      g('direction_layer.weight'), g('direction_layer.bias'), g('rgb_linear.weight'), g('rgb_linear.bias'))


def test_reference_fixture(dev, gold):
    from xrnerf_amd import ops
    mlp = mlp_from_gold(gold, dev)
    R, S = gold['z_vals'].shape
    data = {'pts': T(gold['pts'], dev), 'viewdirs': T(gold['viewdirs'], dev), 'global_domain_min': torch.tensor(gold['gmin']),
            'global_domain_max': torch.tensor(gold['gmax'])}
    with torch.no_grad():                                   # inference: the plain forward call (no autograd node)
        raw = mlp(data)['raw']
    assert tuple(raw.shape) == (R, S, 4) and not raw.requires_grad
    flat = raw.reshape(-1, 4).cpu().numpy()
    active = np.zeros(R * S, bool); active[gold['active_samples']] = True
    assert np.array_equal(flat[~active], np.zeros_like(flat[~active]))           # exact zeros where nothing is evaluated
    assert np.abs(flat - gold['raw'].reshape(-1, 4)).max() <= 1e-4
    # per-network counts = the reference's batch_size_per_network, bit-exact; sample positions built on the fly
    # (o + d*z) give the same rows as the materialised data['pts']
    fixed = [r // 16 for r in mlp.resolution]
    raw2, counts = ops.kilo_mlp_forward(data['viewdirs'], gold['gmin'], gold['gmax'], fixed, mlp.resolution, mlp.occupancy_grid,
                                        mlp.domain_mins, mlp.domain_maxs, mlp.multi_network.packed(), 10, 4, 2,
                                        rays_o=T(gold['rays_o'], dev), rays_d=T(gold['rays_d'], dev), z_vals=T(gold['z_vals'], dev),
                                        want_counts=True)
    assert np.array_equal(counts.cpu().numpy().astype(np.int64), gold['batch_size_per_network'])
    assert torch.equal(raw2, raw)
    # NerfRender on the reference's raw: colours, weights, disparity (NaN where the ray is empty, like torch.max)
    rgb, disp, acc, w = ops.nerf_render_forward(T(gold['raw'], dev), T(gold['z_vals'], dev), T(gold['rays_d'], dev), True)
    assert np.abs(w.cpu().numpy() - gold['weights']).max() <= 2e-6
    assert np.abs(rgb.cpu().numpy() - gold['rgb']).max() <= 5e-6 and np.abs(acc.cpu().numpy() - gold['acc']).max() <= 5e-6
    d, ok = disp.cpu().numpy(), np.isfinite(gold['disp'])
    assert np.array_equal(ok, np.isfinite(d))
    assert np.abs(d[ok] - gold['disp'][ok]).max() <= 1e-5 * max(1.0, np.abs(gold['disp'][ok]).max())


def test_network_behind_the_registry(dev, gold, tmp_path):
    """the reference's finetune config model dict (checkpoint paths pointed at files written here) builds and renders"""
    import copy
    import json
    import xrnerf_amd
    cfg = json.load(open(os.path.join(G, 'kilo_model_cfg.json')))
    model = copy.deepcopy(cfg['model'])
    sd = {}
    for nm in LAYERS:
        sd[nm + '.weight'], sd[nm + '.bias'] = torch.tensor(gold['w.' + nm]), torch.tensor(gold['b.' + nm])
    torch.save(torch.tensor(gold['occupancy']), tmp_path / 'occupancy.pth')
    torch.save({'domain_mins': torch.tensor(gold['domain_mins']), 'domain_maxs': torch.tensor(gold['domain_maxs']), 'state_dict': sd}, tmp_path / 'checkpoint.pth')
    model['mlp'].update(occupancy_checkpoint=str(tmp_path / 'occupancy.pth'), distilled_checkpoint=str(tmp_path / 'checkpoint.pth'),
                        resolution=[int(v) for v in gold['res']])        # core/apis/helper.py:57-64 injects the resolution
    net = xrnerf_amd.build_network(model).to(dev)
    data = {'pts': T(gold['pts'], dev), 'viewdirs': T(gold['viewdirs'], dev), 'z_vals': T(gold['z_vals'], dev),
            'rays_o': T(gold['rays_o'], dev), 'rays_d': T(gold['rays_d'], dev),
            'global_domain_min': T(gold['gmin'], dev), 'global_domain_max': T(gold['gmax'], dev)}
    with torch.no_grad():
        ret = net.batchify_forward(data, is_test=True)
    assert np.abs(ret['rgb'].cpu().numpy() - gold['rgb']).max() <= 1e-4
    assert np.abs(ret['acc'].cpu().numpy() - gold['acc']).max() <= 1e-4
    # one fine-tuning step through the registry classes: loss = MSE + lambda * L2(view-dependent parameters)
    batch = {k: v[None] for k, v in data.items()}
    batch['target_s'] = T(gold['rgb'], dev)[None] * 0.5
    out = net.train_step(batch, None)
    out['loss'].backward()
    vd = net.mlp.get_view_dependent_parameters()
    reg = sum(float(p.detach().norm(2)) for p in vd)
    assert abs(float(out['log_vars']['L2 reg']) - 1e-6 * reg) <= 1e-9 and out['num_samples'] == gold['rgb'].shape[0]
    grads = [p.grad for p in net.mlp.multi_network.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0


@pytest.mark.parametrize('R,S', [(2048, 384), (333, 97)])
def test_lego_grid_against_oracle(dev, K, R, S):
    """1440 networks on the 9x16x10 Lego grid, 144x256x160 occupancy, 384 samples per ray (the config's sizes)"""
    from xrnerf_amd import kilo, ops
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=3)
    rng = np.random.default_rng(R)
    cam = rng.normal(0, 1, (R, 3)); cam = (3.2 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(np.float32)
    d = (rng.uniform(-0.5, 0.5, (R, 3)) + [0, 0, 0.3] - cam).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.9, 1.1, (R, 1))).astype(np.float32)
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    z = np.tile(np.linspace(1.5, 5.0, S, dtype=np.float32), (R, 1))
    fixed = [r // 16 for r in mlp.resolution]
    raw, counts = ops.kilo_mlp_forward(T(vd, dev), gmin.tolist(), gmax.tolist(), fixed, mlp.resolution, mlp.occupancy_grid,
                                       mlp.domain_mins, mlp.domain_maxs, mlp.multi_network.packed(), 10, 4, 2,
                                       rays_o=T(cam, dev), rays_d=T(d, dev), z_vals=T(z, dev), want_counts=True)
    oraw, net, active, ocounts = K.mlp_raw(cam, d, vd, z, gmin.numpy(), gmax.numpy(), fixed, mlp.resolution,
                                           mlp.occupancy_grid.cpu().numpy(), mlp.domain_mins.cpu().numpy(),
                                           mlp.domain_maxs.cpu().numpy(), oracle_nets(K, mlp.multi_network.state_dict()))
    assert active.sum() > 0.01 * R * S
    assert np.array_equal(counts.cpu().numpy().astype(np.int64), ocounts)
    g = raw.cpu().numpy()
    assert np.array_equal((g.reshape(-1, 4) == 0).all(1), ~active) or np.array_equal(g.reshape(-1, 4)[~active], np.zeros((int((~active).sum()), 4), np.float32))
    assert np.abs(g - oraw).max() <= 1e-4
    rgb, disp, acc, w = ops.nerf_render_forward(raw, T(z, dev), T(d, dev), True)
    orgb, odisp, oacc, ow = K.nerf_render(oraw, z, d, True)
    assert np.abs(w.cpu().numpy() - ow).max() <= 1e-4 and np.abs(rgb.cpu().numpy() - orgb).max() <= 1e-4


def test_many_networks_and_other_architectures(dev, K):
    """> 16384 networks (no LDS histogram: the global-atomic path), 1 hidden layer, 1 / 0 Fourier frequencies"""
    from xrnerf_amd import ops
    fixed = [26, 26, 26]
    N = 26 ** 3
    gmin, gmax = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
    nfl = ops.kilo_param_floats(1, 0, 1)
    assert nfl == (9 * 32 + 32) + 36 + (32 * 32 + 32) + ((32 + 3) * 32 + 32) + 132
    g = torch.Generator(device='cpu').manual_seed(0)
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * 0.4)
    sd = {'pts_linears.0.weight': mk(N, 9, 32), 'pts_linears.0.bias': mk(N, 32), 'alpha_linear.weight': mk(N, 32, 1),
          'alpha_linear.bias': mk(N, 1), 'feature_linear.weight': mk(N, 32, 32), 'feature_linear.bias': mk(N, 32),
          'direction_layer.weight': mk(N, 35, 32), 'direction_layer.bias': mk(N, 32), 'rgb_linear.weight': mk(N, 32, 3),
          'rgb_linear.bias': mk(N, 3)}
    from xrnerf_amd import kilo
    mn = kilo.MultiNetwork(N, 9, 3, num_hidden_layers=1)
    mn.load_state_dict(sd)
    packed = mn.to(dev).packed()
    assert packed.shape == (N, nfl)
    rng = np.random.default_rng(2)
    R, S = 500, 33
    pts = rng.uniform(-1.1, 1.1, (R, S, 3)).astype(np.float32)
    vd = rng.normal(0, 1, (R, 3)).astype(np.float32); vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    idx = np.stack(np.meshgrid(*[np.arange(26)] * 3, indexing='ij'), -1).reshape(-1, 3)
    dmins = (-1.0 + idx * (2.0 / 26)).astype(np.float32); dmaxs = (-1.0 + (idx + 1) * (2.0 / 26)).astype(np.float32)
    raw, counts = ops.kilo_mlp_forward(T(vd, dev), gmin, gmax, fixed, None, None, T(dmins, dev), T(dmaxs, dev), packed, 1, 0, 1,
                                       pts=T(pts, dev), want_counts=True)
    oraw, net, active, ocounts = K.mlp_raw(None, None, vd, np.zeros((R, S), np.float32), np.float32(gmin), np.float32(gmax), fixed,
                                           None, None, dmins, dmaxs, oracle_nets(K, sd), pos_freqs=1, dir_freqs=0, pts=pts)
    assert np.array_equal(counts.cpu().numpy().astype(np.int64), ocounts)
    assert np.abs(raw.cpu().numpy() - oraw).max() <= 1e-4


def test_edge_cases_and_validation(dev, gold):
    from xrnerf_amd import _lib, ops
    mlp = mlp_from_gold(gold, dev)
    fixed = [r // 16 for r in mlp.resolution]
    args = (gold['gmin'], gold['gmax'], fixed, mlp.resolution, mlp.occupancy_grid, mlp.domain_mins, mlp.domain_maxs,
            mlp.multi_network.packed(), 10, 4, 2)
    vd = T(gold['viewdirs'], dev)
    far = torch.full((48, 40, 3), 50.0, device=dev)                           # every sample outside the domain
    raw, counts = ops.kilo_mlp_forward(vd, *args, pts=far, want_counts=True)
    assert float(raw.abs().max()) == 0.0 and int(counts.sum()) == 0
    empty = ops.kilo_mlp_forward(vd[:0], *args, pts=far[:0])
    assert tuple(empty.shape) == (0, 40, 4)
    with pytest.raises(_lib.XrError):
        ops.kilo_mlp_forward(vd.cpu(), *args, pts=far.cpu())                    # no CPU fallback
    with pytest.raises(_lib.XrError):
        ops.kilo_mlp_forward(vd, *args[:7], mlp.multi_network.packed()[:, :100].contiguous(), 10, 4, 2, pts=far)   # short blocks
    # an empty ray: white background, acc 0, disparity NaN (0/0 through torch.max), as NerfRender gives
    rgb, disp, acc, w = ops.nerf_render_forward(torch.zeros(2, 8, 4, device=dev), torch.linspace(2, 6, 8, device=dev).expand(2, 8).contiguous(),
                                                torch.ones(2, 3, device=dev), True)
    assert float((rgb - 1).abs().max()) == 0.0 and float(acc.abs().max()) == 0.0 and bool(torch.isnan(disp).all())


def test_fused_frame_path_gives_the_same_pixels(dev):
    """xr_kilo_render_rays (no per-sample tensor but the network id, empty rows neither filled nor read) against the
    module-level path (z_vals, dense raw [R,S,4], NerfRender) on a 160x160 view of the Lego-shaped scene: bit-identical"""
    from xrnerf_amd import kilo
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=1)
    pose = kilo.orbit_poses(5)[2]
    H = W = 160
    a = kilo.render_frame(mlp, gmin, gmax, pose, H, W, 1111.111 * W / 800, fused=True)
    b = kilo.render_frame(mlp, gmin, gmax, pose, H, W, 1111.111 * W / 800, fused=False)
    assert float(b[2].max()) > 0.1 and float((b[2] > 0).float().mean()) > 0.02        # the object is in view
    for x, y in zip(a, b):
        assert torch.equal(torch.nan_to_num(x, nan=-1.0), torch.nan_to_num(y, nan=-1.0))


def test_fused_frame_path_spans_on_adversarial_rays(dev):
    """the per-ray sample spans of the fused call must never drop a sample the reference evaluates: axis-parallel rays
    (zero direction components), origins inside the domain, grazing rays along the domain faces, rays that miss, far < near
    spans -- fused call == module-level dense path, bit for bit"""
    from xrnerf_amd import kilo, ops
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=4, fill=0.25)
    rng = np.random.default_rng(12)
    lo, hi = np.float32(kilo.LEGO_GMIN), np.float32(kilo.LEGO_GMAX)
    o, d = [], []
    for k in range(600):
        kind = k % 6
        tgt = rng.uniform(lo, hi)
        org = rng.normal(0, 1, 3); org = 3.5 * org / np.linalg.norm(org)
        if kind == 0:                                        # axis-parallel through the box
            a = rng.integers(3); org = tgt.copy(); org[a] = -4.0; dd = np.zeros(3); dd[a] = 1.0
        elif kind == 1:                                      # origin inside the domain
            org = rng.uniform(lo, hi); dd = rng.normal(0, 1, 3)
        elif kind == 2:                                      # grazing a face
            a = rng.integers(3); org = tgt.copy(); org[(a + 1) % 3] = -4.0; org[a] = (hi if k % 2 else lo)[a] + rng.normal(0, 2e-3)
            dd = np.zeros(3); dd[(a + 1) % 3] = 1.0; dd[a] = rng.normal(0, 5e-4)
        elif kind == 3:                                      # misses the box
            dd = org + rng.normal(0, 0.3, 3)
        else:
            dd = tgt - org
        dd = dd / np.linalg.norm(dd) * rng.uniform(0.7, 1.3)
        o.append(org); d.append(dd)
    o, d = np.float32(o), np.float32(d)
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    rays = (T(o, dev), T(d, dev), T(vd, dev))
    for near, far in ((0.5, 8.0), (2.0, 6.0)):
        a = kilo.render_frame(mlp, gmin, gmax, None, 0, 0, 0, near=near, far=far, n_samples=384, rays=rays, fused=True)
        b = kilo.render_frame(mlp, gmin, gmax, None, 0, 0, 0, near=near, far=far, n_samples=384, rays=rays, fused=False)
        assert float((b[2] > 0).float().mean()) > 0.15
        for x, y in zip(a, b):
            assert torch.equal(torch.nan_to_num(x, nan=-1.0), torch.nan_to_num(y, nan=-1.0))


def _oracle_grads(K, mlp, cam, d, vd, z, gmin, gmax, draw, pos_freqs=10, dir_freqs=4, pts=None):
    fixed = [r // 16 for r in mlp.resolution]
    return K.mlp_raw_backward(draw, cam, d, vd, z, np.float32(gmin), np.float32(gmax), fixed, mlp.resolution,
                              mlp.occupancy_grid.cpu().numpy() if mlp.occupancy_grid is not None else None,
                              mlp.domain_mins.cpu().numpy(), mlp.domain_maxs.cpu().numpy(),
                              oracle_nets(K, mlp.multi_network.state_dict()), pos_freqs, dir_freqs, pts=pts)


def test_parameter_gradients_on_the_reference_fixture(dev, K, gold):
    """xr_kilo_mlp_backward against the oracle's adjoint (pinned to torch autograd through the reference's MultiNetwork):
    every parameter tensor of all 24 networks, dL/draw random, rows without a network ignored"""
    from xrnerf_amd import kilo, ops
    mlp = mlp_from_gold(gold, dev)
    rng = np.random.default_rng(21)
    R, S = gold['z_vals'].shape
    draw = rng.normal(0, 1, (R, S, 4)).astype(np.float32)
    fixed = [r // 16 for r in mlp.resolution]
    mn = mlp.multi_network
    g = ops.kilo_mlp_backward(T(draw, dev), T(gold['viewdirs'], dev), gold['gmin'], gold['gmax'], fixed, mlp.resolution,
                              mlp.occupancy_grid, mlp.domain_mins, mlp.domain_maxs, mn.packed(), 10, 4, 2, pts=T(gold['pts'], dev))
    got = kilo.MultiNetwork.unpack_like(g, mn.ordered_parameters())
    ref = _oracle_grads(K, mlp, None, None, gold['viewdirs'], gold['z_vals'], gold['gmin'], gold['gmax'], draw, pts=gold['pts'])
    for (name, _), t in zip(mn.named_parameters(), got):
        r = ref[name]
        assert np.abs(t.cpu().numpy().astype(np.float64) - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), name
    # padding slots of the blocks stay untouched
    pad = torch.cat([g[:, 63 * 32 + 32 + 1056 + 33:63 * 32 + 32 + 1056 + 36], g[:, -1:]], 1)
    assert float(pad.abs().max()) == 0.0


def test_parameter_gradients_on_the_lego_grid_through_autograd(dev, K):
    """1440 networks, 384 samples per ray: the autograd node (KiloNerfMLP.forward under grad, .backward()) against the
    oracle's adjoint for every parameter tensor"""
    from xrnerf_amd import kilo, ops
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=6)
    rng = np.random.default_rng(5)
    R, S = 384, 384
    cam = rng.normal(0, 1, (R, 3)); cam = (3.2 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)).astype(np.float32)
    d = (rng.uniform(-0.4, 0.4, (R, 3)) + [0, 0, 0.3] - cam).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    z = np.tile(np.linspace(1.5, 5.0, S, dtype=np.float32), (R, 1))
    draw = rng.normal(0, 1, (R, S, 4)).astype(np.float32)
    mn = mlp.multi_network
    data = {'rays_o': T(cam, dev), 'rays_d': T(d, dev), 'viewdirs': T(d, dev), 'z_vals': T(z, dev),
            'global_domain_min': gmin, 'global_domain_max': gmax}
    raw = mlp(dict(data))['raw']
    assert raw.requires_grad
    (raw * T(draw, dev)).sum().backward()
    ref = _oracle_grads(K, mlp, cam, d, d, z, gmin.numpy(), gmax.numpy(), draw)
    for name, p in mn.named_parameters():
        r = ref[name]
        assert np.abs(p.grad.cpu().numpy().astype(np.float64) - r).max() <= 2e-4 * max(1.0, np.abs(r).max()), name


def test_parameter_gradients_other_architecture(dev, K):
    """one hidden layer, 1 / 0 Fourier frequencies, > 16384 networks (global-atomic assignment path)"""
    from xrnerf_amd import kilo, ops
    fixed = [26, 26, 26]
    N = 26 ** 3
    g0 = torch.Generator(device='cpu').manual_seed(0)
    mk = lambda *s: ((torch.rand(*s, generator=g0) * 2 - 1) * 0.4)
    sd = {'pts_linears.0.weight': mk(N, 9, 32), 'pts_linears.0.bias': mk(N, 32), 'alpha_linear.weight': mk(N, 32, 1),
          'alpha_linear.bias': mk(N, 1), 'feature_linear.weight': mk(N, 32, 32), 'feature_linear.bias': mk(N, 32),
          'direction_layer.weight': mk(N, 35, 32), 'direction_layer.bias': mk(N, 32), 'rgb_linear.weight': mk(N, 32, 3),
          'rgb_linear.bias': mk(N, 3)}
    mn = kilo.MultiNetwork(N, 9, 3, num_hidden_layers=1)
    mn.load_state_dict(sd)
    mn = mn.to(dev)
    rng = np.random.default_rng(2)
    R, S = 300, 21
    pts = rng.uniform(-1.1, 1.1, (R, S, 3)).astype(np.float32)
    vd = rng.normal(0, 1, (R, 3)).astype(np.float32); vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    idx = np.stack(np.meshgrid(*[np.arange(26)] * 3, indexing='ij'), -1).reshape(-1, 3)
    dmins = (-1.0 + idx * (2.0 / 26)).astype(np.float32); dmaxs = (-1.0 + (idx + 1) * (2.0 / 26)).astype(np.float32)
    draw = rng.normal(0, 1, (R, S, 4)).astype(np.float32)
    g = ops.kilo_mlp_backward(T(draw, dev), T(vd, dev), [-1.0] * 3, [1.0] * 3, fixed, None, None, T(dmins, dev), T(dmaxs, dev),
                              mn.packed(), 1, 0, 1, pts=T(pts, dev))
    got = kilo.MultiNetwork.unpack_like(g, mn.ordered_parameters())
    ref = K.mlp_raw_backward(draw, None, None, vd, np.zeros((R, S), np.float32), np.float32([-1] * 3), np.float32([1] * 3), fixed,
                             None, None, dmins, dmaxs, oracle_nets(K, sd), 1, 0, pts=pts)
    for (name, _), t in zip(mn.named_parameters(), got):
        r = ref[name]
        assert np.abs(t.cpu().numpy().astype(np.float64) - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), name


def test_fine_tuning_reduces_the_loss(dev):
    """Adam on all 1440 networks' parameters through KiloNerfNetwork.train_step: the loss goes down"""
    import xrnerf_amd
    from xrnerf_amd import kilo, vanilla
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=7, weight_scale=1.0)
    net = kilo.KiloNerfNetwork.__new__(kilo.KiloNerfNetwork)
    torch.nn.Module.__init__(net)
    net.phase, net.chunk, net.bs_data, net.N_importance, net.is_perturb = 'train', 40000, 'rays_o', 0, True
    net.mlp, net.render, net.l2_regularization_lambda = mlp, vanilla.NerfRender(white_bkgd=True, raw_noise_std=0), 1e-6
    rays_o, rays_d, viewdirs = kilo.camera_rays(kilo.orbit_poses(4)[1], 96, 96, 1111.111 * 96 / 800, dev)
    z = torch.linspace(2.0, 6.0, 192, device=dev).expand(rays_o.shape[0], 192).contiguous()
    target = torch.rand(rays_o.shape[0], 3, device=dev) * 0.5 + 0.25
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for it in range(8):
        batch = {'rays_o': rays_o[None], 'rays_d': rays_d[None], 'viewdirs': viewdirs[None], 'z_vals': z[None],
                 'target_s': target[None], 'global_domain_min': gmin, 'global_domain_max': gmax}
        out = net.train_step(batch, opt)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        losses.append(float(out['log_vars']['loss']))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
