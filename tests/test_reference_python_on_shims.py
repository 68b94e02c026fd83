"""The REFERENCE's own Instant-NGP Python executed, unmodified, on this library (boundary levels 1-2):
`/root/reference/xrnerf/models/samplers/utils/*.py`, `samplers/ngp_grid_sampler.py`, `renders/hashnerf_render.py`,
`mlps/hashnerf_mlp.py`, `networks/hashnerf.py` are imported from where they lie with
    sys.modules['raymarch_cuda'] = xrnerf_amd.raymarch_cuda ;  sys.modules['tinycudann'] = xrnerf_amd.tcnn
and run next to this package's registry classes from identical state and rays.  The kernels are the real sources
(xrnerf_amd/csrc/*.hip) executed on the host by tests/hip_emu.  Asserted: identical occupancy bitfield / density grid after
every refresh, identical rays-per-batch trajectory, identical per-ray sample counts and sample rows, loss within 1e-5.
Needs /root/reference (build container); the same trajectory is replayed on the GPU from a committed fixture
(tests/test_gpu_trajectory.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hip_emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='needs the reference tree (/root/reference)')


@pytest.fixture(scope='module')
def edev():
    import emulib
    ctx = emulib.emulated_ops()
    dev = ctx.__enter__()
    yield dev
    ctx.__exit__(None, None, None)


@pytest.fixture(scope='module')
def ref(edev):
    import xrnerf_amd.raymarch_cuda as rc
    import xrnerf_amd.tcnn as tc
    return ref_import.load_ngp(rc, tc)


def test_reference_sampler_and_wrappers_run_unmodified_and_match_ours(edev, ref):
    """33 training iterations of NGPGridSampler.sample (3 grid refreshes: iterations 0, 16, 32; 2 batch-size adaptations:
    15, 31) -- the reference class through its own ten wrappers on the `raymarch_cuda` drop-in, this package's class on
    `ops` -- sharing one density / colour model."""
    import ngp_ref_harness as Hn
    import xrnerf_amd
    import xrnerf_amd.raymarch_cuda as rc
    from xrnerf_amd.samplers import NGPGridSampler as Ours
    poses, alldata, info = Hn.scene()
    kw = dict(update_grid_freq=16, update_block_size=5000000, n_rays_per_batch=1024, cone_angle_constant=0.00390625,
              near_distance=0.2, target_batch_size=1 << 16, rgb_activation=2, density_activation=3)
    mlp = Hn.OracleMlp()
    rs, os_ = ref.NGPGridSampler(**kw), Ours(**kw)
    rs.set_data(alldata, info); os_.set_data(alldata, info)
    rc.reset_rng()
    traj = []
    for it in range(33):
        assert rs.n_rays_per_batch == os_.n_rays_per_batch
        b = Hn.batch(poses, rs.n_rays_per_batch, it, edev)
        rs.set_iter(it); os_.set_iter(it)
        rs.sample({k: v.clone() for k, v in b.items()}, mlp, False)
        os_.sample({k: v.clone() for k, v in b.items()}, mlp, False)
        a, o = Hn.sampler_state(rs), Hn.sampler_state(os_)
        Hn.compare_states(a, o, it)
        traj.append((a['n_rays_per_batch'], int(a['numsteps'][:, 0].sum()), int(np.unpackbits(a['bitfield']).sum())))
    # the trajectory is not trivial: the batch size moved at the first adaptation (the formula reaches its fixed point in
    # one step unless the samples per ray drift by a 128-ray rounding step), the grid changed at every refresh
    sizes = [t[0] for t in traj]
    print('trajectory (rays/batch, marched samples, occupied bits):', traj[::4])
    assert sizes[14] != sizes[15] and sizes[15] == sizes[16], sizes
    assert len({traj[0][2], traj[16][2], traj[32][2]}) == 3, [traj[k][2] for k in (0, 16, 32)]
    assert rs.density_grid_ema_step == os_.density_grid_ema_step == 3
    # test-mode marching of the reference class (is_test=True) against ours
    b = Hn.batch(poses, 512, 99, edev)
    ra = rs.sample({k: v.clone() for k, v in b.items()}, mlp, True)
    oa = os_.sample({k: v.clone() for k, v in b.items()}, mlp, True)
    assert np.array_equal(rs.rays_numsteps.numpy(), os_.rays_numsteps.numpy())
    assert np.array_equal(ra['pts'].numpy(), oa['pts'].numpy()) and np.array_equal(ra['viewdirs'].numpy(), oa['viewdirs'].numpy())


def test_reference_sampler_unbounded_scene_five_cascades(edev, ref):
    """BASELINE config #4's setting, aabb_scale = 16 (max_cascade = 4: K6 samples 5 x 128^3 cells per refresh, K1 picks the
    cascade from step size and position, positions are warped into the 16-unit box): the reference's sampler on the
    `raymarch_cuda` drop-in against ours, refresh + 3 marches (+ a test-mode march)."""
    import ngp_ref_harness as Hn
    import xrnerf_amd.raymarch_cuda as rc
    from xrnerf_amd.samplers import NGPGridSampler as Ours
    poses, alldata, info = Hn.scene(aabb_scale=16)
    kw = dict(update_grid_freq=16, n_rays_per_batch=512, target_batch_size=1 << 15)
    mlp = Hn.OracleMlp(seed=1)
    rs, os_ = ref.NGPGridSampler(**kw), Ours(**kw)
    rs.set_data(alldata, info); os_.set_data(alldata, info)
    assert rs.max_cascade == os_.max_cascade == 4
    rc.reset_rng()
    for it in range(3):
        b = Hn.batch(poses, rs.n_rays_per_batch, it, edev)
        rs.set_iter(it); os_.set_iter(it)
        rs.sample({k: v.clone() for k, v in b.items()}, mlp, False)
        os_.sample({k: v.clone() for k, v in b.items()}, mlp, False)
        a, o = Hn.sampler_state(rs), Hn.sampler_state(os_)
        Hn.compare_states(a, o, it)
    per_cascade = [int(np.unpackbits(a['bitfield'][c * 262144:(c + 1) * 262144]).sum()) for c in range(8)]
    assert all(v > 0 for v in per_cascade[:5]), per_cascade          # (cascades 5..7 only hold K11's max-pooled copies)
    assert a['coords'][:, :3].min() >= 0 and a['coords'][:, :3].max() <= 1          # warped positions
    b = Hn.batch(poses, 300, 50, edev)
    ra = rs.sample({k: v.clone() for k, v in b.items()}, mlp, True)
    oa = os_.sample({k: v.clone() for k, v in b.items()}, mlp, True)
    assert np.array_equal(rs.rays_numsteps.numpy(), os_.rays_numsteps.numpy()) and np.array_equal(ra['pts'].numpy(), oa['pts'].numpy())


def _preset_grid(sampler, edev):
    """occupancy state of a trained scene (the synthetic Lego boxes) instead of a 2 M-point density query through the
    emulated MLP: density grid -> mean / bitfield by the real K10 / K11"""
    from xrnerf_amd import ops, synthetic as S
    grid = torch.from_numpy(S.lego_density_grid())
    sampler.density_grid = grid.clone()
    ops.update_bitfield(sampler.density_grid, sampler.density_grid_mean, sampler.density_grid_bitfield)
    sampler.density_grid_ema_step = 1


@pytest.mark.parametrize('arith', ['b2', 'b2x'])
def test_reference_network_mlp_render_on_the_tcnn_and_raymarch_shims_match_ours(edev, ref, arith, monkeypatch):
    """HashNerfNetwork.train_step of the reference (its sampler, its HashNerfMLP on `tinycudann` = xrnerf_amd.tcnn, its
    HashNerfRender autograd Functions on `raymarch_cuda`, its HuberLoss) against this package's network from the same
    weights, for three iterations with an Adam step in between: rgb, loss, PSNR, every parameter gradient, the parameters
    after the steps.

    The reference's module-by-module route reaches the single-network kernels (fp32 MFMA throughout), this package's fused
    backward runs its dX chain on the bf16 matrix cores with 2-way split operands by default (XR_MLP_BWD_DW=b2x: 2^-16
    relative per product instead of 2^-24).  Gradients agree to 2e-5 of their scale either way.  After three Adam steps
    (eps = 1e-15: a table entry whose gradient is rounding noise moves by +-lr per step whatever the noise's size) the
    parameters agree to 1e-4 everywhere with the dX chain on fp32-grade products (arith = b2), and everywhere but on
    < 1e-5 of the table entries with the default (arith = b2x)."""
    monkeypatch.setenv('XR_MLP_BWD_DW', arith)
    import ngp_ref_harness as Hn
    import xrnerf_amd
    import xrnerf_amd.raymarch_cuda as rc
    from xrnerf_amd.train import ngp_lego_model_cfg, FusedAdam
    poses, alldata, info = Hn.scene()
    cfg = ngp_lego_model_cfg(n_rays=512)
    cfg['sampler']['target_batch_size'] = 1 << 14
    mine = xrnerf_amd.build_network(cfg)
    rcfg = ngp_lego_model_cfg(n_rays=512)
    rcfg['sampler']['target_batch_size'] = 1 << 14
    rcfg.pop('type')
    theirs = ref.HashNerfNetwork(ref_import.Cfg(rcfg.pop('cfg')), **{k: dict(v) for k, v in rcfg.items()})
    # identical weights (same state_dict keys: mlp.embedder_pos.params, mlp.density_net.params, mlp.color_net.params)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for name in ('embedder_pos', 'density_net', 'color_net'):
            p = getattr(theirs.mlp, name).params
            p.copy_(torch.empty_like(p).uniform_(-0.3, 0.3, generator=g) if name != 'embedder_pos'
                    else torch.empty_like(p).uniform_(-0.5, 0.5, generator=g))
    missing = mine.load_state_dict(theirs.state_dict(), strict=False)
    assert not missing.unexpected_keys and all('params' not in k for k in missing.missing_keys), missing
    for net in (theirs, mine):
        net.sampler.set_data(alldata, info)
        net.sampler.check_device({'rays_o': torch.zeros(1, 3)})
        _preset_grid(net.sampler, edev)
    rc.reset_rng()
    mine.sampler.k1_calls = 0
    opt_t = torch.optim.Adam(theirs.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    opt_m = FusedAdam([p for p in mine.parameters() if p.numel() > 0], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    for it in (1, 2, 3):
        b = Hn.batch(poses, 512, it, edev)
        out = {}
        for name, net, opt in (('ref', theirs, opt_t), ('ours', mine, opt_m)):
            net.sampler.set_iter(it)
            o = net.train_step({k: v.clone()[None] for k, v in b.items()}, opt)
            opt.zero_grad(set_to_none=True)
            o['loss'].backward()
            out[name] = (float(o['loss'].detach()), float(o['log_vars']['psnr']),
                         {n: getattr(net.mlp, n).params.grad.clone() for n in ('embedder_pos', 'density_net', 'color_net')})
            opt.step()
        (lr, pr, gr), (lo, po, go) = out['ref'], out['ours']
        assert abs(lr - lo) <= 1e-5 * abs(lr), (it, lr, lo)
        assert abs(pr - po) <= 1e-4, (it, pr, po)
        assert np.array_equal(theirs.sampler.rays_numsteps.numpy(), mine.sampler.rays_numsteps.numpy())
        for n in gr:
            scale = float(gr[n].abs().max())
            assert scale > 0 and float((gr[n] - go[n]).abs().max()) <= 2e-5 * scale, (it, n, scale, float((gr[n] - go[n]).abs().max()))
    for n in ('embedder_pos', 'density_net', 'color_net'):
        a, c = getattr(theirs.mlp, n).params.detach(), getattr(mine.mlp, n).params.detach()
        # Adam divides by sqrt(v): entries whose gradients differ at summation-order level move apart by more than that
        d = (a - c).abs()
        if arith == 'b2':
            assert float(d.max()) <= 1e-4, (n, float(d.max()))
        else:
            assert float(d.max()) <= 3.5e-2 and int((d > 1e-4).sum()) <= 1e-5 * d.numel(), (n, float(d.max()), int((d > 1e-4).sum()))
    # validation forward (is_test=True: K1 without clipping, K5 compositor) through both
    b = Hn.batch(poses, 300, 77, edev)
    with torch.no_grad():
        rt = theirs.forward({k: b[k].clone() for k in ('rays_o', 'rays_d', 'img_ids')}, is_test=True)
        rm = mine.forward({k: b[k].clone() for k in ('rays_o', 'rays_d', 'img_ids')}, is_test=True)
    assert float((rt['rgb'] - rm['rgb']).abs().max()) <= 1e-5 and float((rt['alpha'] - rm['alpha']).abs().max()) <= 1e-5
