"""GPU replay of tests/golden/ref_ngp_trajectory.npz: the fixture is 33 training iterations of the REFERENCE's own Python
stack (its NGPGridSampler / HashNerfMLP / HashNerfRender / HashNerfNetwork files, unmodified, on this library's
`raymarch_cuda` + `tinycudann` drop-ins, kernels on the host emulator; tests/golden/make_golden_ngp_trajectory.py).  Here
this package's registry classes run the same 33 iterations on the MI355X -- fused training step, fused Adam -- from the
same seeded weights and batches, and must follow the same trajectory: rays per batch, marched samples, occupancy grid,
loss, final MLP weights.  The two runs differ in libm vs device exp / sigmoid and in gradient summation order only."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def test_training_trajectory_follows_the_reference_python_stack(dev):
    import ngp_ref_harness as Hn
    import make_golden_ngp_trajectory as G
    import xrnerf_amd
    from xrnerf_amd.train import FusedAdam
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_ngp_trajectory.npz'))
    assert int(fx['n_rays0']) == G.N_RAYS0 and int(fx['target']) == G.TARGET and int(fx['n_img']) == Hn.N_IMG
    poses, alldata, info = Hn.scene()
    net = xrnerf_amd.build_network(G.model_cfg()).to(dev)
    G.init_weights(net.mlp)
    net.sampler.set_data(alldata, info)
    opt = FusedAdam([p for p in net.parameters() if p.numel() > 0], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    assert net._fused_ok()
    rec = {k: [] for k in ('n_rays', 'marched', 'loss', 'psnr', 'popcount')}
    bitfields = {}
    # The first refresh depends on the initial weights only, and differs from the fixture's in a few hundred of 2 M bits (device exp
    # against libm's at the occupancy threshold).  Right behind it -- before iteration 0 marches -- cascade 0 of the bitfield is
    # replaced by the fixture's: iterations 0..15 then march the SAME occupancy with the same rays and the same jitter stream,
    # so their sample counts (and the batch size adapted from them at iteration 15) must equal the fixture's exactly.
    from xrnerf_amd import ops
    real_update, seeded = ops.ema_update_bitfield, {}          # (the sampler's refresh tail: K9 + K10 + K11 as one entry point)

    def update_then_seed(grid_tmp, n_elements, decay, grid, mean, bitfield):
        out = real_update(grid_tmp, n_elements, decay, grid, mean, bitfield)
        if not seeded:
            seeded['own'] = bitfield[:128 ** 3 // 8].cpu().numpy().copy()
            bitfield[:128 ** 3 // 8].copy_(torch.from_numpy(fx['bitfield_it0']).to(bitfield.device))
        return out
    ops.ema_update_bitfield = update_then_seed
    try:
        rec, bitfields = _run(net, opt, poses, fx, dev, Hn, rec, bitfields)
    finally:
        ops.ema_update_bitfield = real_update
    bitfields[0] = seeded['own']
    _check(rec, bitfields, fx, net)


def _run(net, opt, poses, fx, dev, Hn, rec, bitfields):
    for it in range(int(fx['n_iters'])):
        n_rays = int(net.sampler.n_rays_per_batch)
        b = Hn.batch(poses, n_rays, it, dev)
        net.sampler.set_iter(it)
        out = net.train_step({k: v[None] for k, v in b.items()}, opt)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        rec['n_rays'].append(n_rays)
        rec['marched'].append(int(net.sampler.rays_numsteps[:, 0].sum()))
        rec['loss'].append(float(out['log_vars']['loss']))
        rec['psnr'].append(float(out['log_vars']['psnr']))
        bf = net.sampler.density_grid_bitfield.cpu().numpy()
        rec['popcount'].append(int(np.unpackbits(bf).sum()))
        if it % 16 == 0:
            bitfields[it] = bf[:128 ** 3 // 8].copy()
    return rec, bitfields


def _check(rec, bitfields, fx, net):
    print('rays/batch', rec['n_rays'][::8], 'ref', fx['n_rays'][::8].tolist())
    print('marched   ', rec['marched'][::8], 'ref', fx['marched'][::8].tolist())
    print('loss      ', [round(v, 4) for v in rec['loss'][::8]], 'ref', [round(float(v), 4) for v in fx['loss'][::8]])
    # the first refresh and the first 15 marches depend on the initial weights only: identical up to exp() rounding at the
    # occupancy threshold
    ham0 = int(np.unpackbits(bitfields[0] ^ fx['bitfield_it0']).sum())
    assert ham0 <= 2e-4 * 128 ** 3, ham0
    # iterations 0..15 on the fixture's first bitfield: exact sample counts, hence the exact batch size from iteration 16 on
    assert rec['marched'][:16] == fx['marched'][:16].tolist(), (rec['marched'][:16], fx['marched'][:16].tolist())
    assert rec['n_rays'][:17] == fx['n_rays'][:17].tolist()
    # (from the refresh at iteration 16 on the occupancy depends on 16 optimiser steps of weights: the looser bounds below)
    # whole trajectory
    for it in range(int(fx['n_iters'])):
        assert abs(rec['n_rays'][it] - int(fx['n_rays'][it])) <= 128, (it, rec['n_rays'][it], int(fx['n_rays'][it]))
        assert abs(rec['marched'][it] - int(fx['marched'][it])) <= 0.02 * int(fx['marched'][it]), (it, rec['marched'][it], int(fx['marched'][it]))
        assert abs(rec['loss'][it] - float(fx['loss'][it])) <= 5e-3 * abs(float(fx['loss'][it])), (it, rec['loss'][it], float(fx['loss'][it]))
        assert abs(rec['psnr'][it] - float(fx['psnr'][it])) <= 0.05, (it, rec['psnr'][it], float(fx['psnr'][it]))
        assert abs(rec['popcount'][it] - int(fx['popcount'][it])) <= 0.01 * int(fx['popcount'][it]), it
    for it in (16, 32):
        ham = int(np.unpackbits(bitfields[it] ^ fx['bitfield_it%d' % it]).sum())
        assert ham <= 0.01 * 128 ** 3, (it, ham)
    for name in ('density_net', 'color_net'):
        a = getattr(net.mlp, name).params.detach().cpu().numpy()
        r = fx['final_' + name]
        # 33 Adam steps of 1e-2 each: the update is g / sqrt(v), so single weights whose gradients sit at summation-order
        # noise level drift apart by several steps (seen: up to 0.09 on one of 7168 weights, and run to run -- the dense
        # levels' atomics are unordered); the bulk must agree
        # (every weight moves by ~1e-2 per step, up to 0.33 in total: the two runs agree to ~1 % of that on average)
        assert np.abs(a - r).mean() <= 2e-2 * np.abs(r).max(), (name, float(np.abs(a - r).mean()))
