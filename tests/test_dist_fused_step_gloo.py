"""The data-parallel training step on two CPU processes: torch.distributed over gloo for the one real
exchange of the path (the bucketed gradient all-reduce), the kernels executed from their real sources by tests/hip_emu.
Each rank marches ITS OWN rays through the fused step of xrnerf_amd.networks (live-row list, MLP backward, the two scatter
halves handed to the reduction one after the other), then the fused Adam.  Asserted on every rank and iteration:
  * the reduced gradient equals the mean of the two ranks' local gradients (computed by a second, un-synchronised network
    from the same weights and rays) to summation-order accuracy,
  * parameters stay bit-identical across the ranks after the optimiser step."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
ROOT = %r
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'hip_emu'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
import emulib
import ngp_ref_harness as Hn
import xrnerf_amd
from xrnerf_amd import dist as xd, ops
from xrnerf_amd.train import ngp_lego_model_cfg, FusedAdam

rank, local, world = xd.init_from_env('gloo')
assert world == 2
N_RAYS = 96
with emulib.emulated_ops() as edev:
    poses, alldata, info = Hn.scene()
    def build():
        cfg = ngp_lego_model_cfg(n_rays=N_RAYS)
        cfg['sampler']['target_batch_size'] = 1 << 13
        net = xrnerf_amd.build_network(cfg)
        g = torch.Generator().manual_seed(11)                       # the same weights on every rank
        with torch.no_grad():
            for name in ('embedder_pos', 'density_net', 'color_net'):
                p = getattr(net.mlp, name).params
                p.copy_(torch.empty_like(p).uniform_(-0.4, 0.4, generator=g))
        net.sampler.set_data(alldata, info)
        net.sampler.check_device({'rays_o': torch.zeros(1, 3)})
        from xrnerf_amd import synthetic as S
        net.sampler.density_grid = torch.from_numpy(S.lego_density_grid()).clone()
        ops.update_bitfield(net.sampler.density_grid, net.sampler.density_grid_mean, net.sampler.density_grid_bitfield)
        net.sampler.density_grid_ema_step = 1
        return net
    dp, solo, dp2 = build(), build(), build()
    dp.grad_sync = xd.BucketedGradSync(world)
    opt = FusedAdam([p for p in dp.parameters() if p.numel() > 0], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    # the trainer's form of the same step (xrnerf_amd/train.py): unit root gradient, .grad keeps the all-reduced SUM and
    # the optimiser applies the 1/world_size while reading it (FusedAdam.step(grad_scale=...))
    dp2.grad_sync = xd.BucketedGradSync(world)
    dp2._defer_grad_scale = True
    dp2._unit_root_grad = one = torch.ones(())
    opt2 = FusedAdam([p for p in dp2.parameters() if p.numel() > 0], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    # SURVEY.md section 8e's form (XRNERF_DP=zero1): reduce-scatter -> Adam on this rank's shard of the table -> all-gather
    dp3 = build()
    dp3.grad_sync = z1 = xd.Zero1GradSync(world, rank)
    dp3._defer_grad_scale = True
    dp3._unit_root_grad = one
    shard = z1.attach(dp3.mlp.embedder_pos.params)
    opt3 = FusedAdam([shard, dp3.mlp.density_net.params, dp3.mlp.color_net.params], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
    names = ('embedder_pos', 'density_net', 'color_net')
    for it in (1, 2):
        b = Hn.batch(poses, N_RAYS, 100 * rank + it, edev)          # every rank its own rays
        grads = {}
        for tag, net in (('solo', solo), ('dp', dp), ('dp2', dp2), ('dp3', dp3)):
            net.sampler.set_iter(it)
            net.sampler.k1_calls = 7 * it + rank                    # same march jitter for both networks of this rank
            o = net.train_step({k: v.clone()[None] for k, v in b.items()}, None)
            for n in names:
                getattr(net.mlp, n).params.grad = None
            if net is dp2 or net is dp3:
                torch.autograd.backward(o['loss'], grad_tensors=one)
            else:
                o['loss'].backward()
            if net is dp3:                                           # zero1: the table gradient goes to the reduce-scatter, not to .grad
                assert net.mlp.embedder_pos.params.grad is None
                continue
            grads[tag] = {n: getattr(net.mlp, n).params.grad.detach().clone() for n in names}
        assert dp2._pending_grad_scale == 0.5
        for n in names:                                              # the deferred form holds the SUM: mean = sum / 2 exactly
            assert torch.equal(grads['dp2'][n] * 0.5, grads['dp'][n]), (rank, it, n)
        for n in names:
            mine = grads['solo'][n]
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            want = (both[0] + both[1]) * 0.5
            scale = float(want.abs().max())
            err = float((grads['dp'][n] - want).abs().max())
            assert scale > 0 and err <= 2e-6 * scale, (rank, it, n, err, scale)
            assert float((both[0] - both[1]).abs().max()) > 1e-3 * scale        # the ranks really saw different rays
        opt.step()
        opt2.step(grad_scale=dp2._pending_grad_scale)
        dp2._pending_grad_scale = 1.0
        # zero1: this rank's shard of the reduce-scattered SUM equals the all-reduced sum's slice; after Adam on the shard and
        # the all-gather the whole table equals the replicated-optimiser result bit for bit (two ranks: a + b is one rounding)
        lo, hi = rank * z1.shard, min((rank + 1) * z1.shard, z1.n)
        assert torch.equal(z1.shard_grad[:hi - lo], grads['dp2']['embedder_pos'][lo:hi]), (rank, it)
        assert dp3._pending_grad_scale == 0.5
        opt3.step(grad_scale=dp3._pending_grad_scale)
        dp3._pending_grad_scale = 1.0
        z1.gather_params()
        for n in names:
            assert torch.equal(getattr(dp3.mlp, n).params.detach(), getattr(dp2.mlp, n).params.detach()), (rank, it, n, 'zero1')
        for n in names:                                              # same update, bit for bit, without the scaling pass
            assert torch.equal(getattr(dp2.mlp, n).params.detach(), getattr(dp.mlp, n).params.detach()), (rank, it, n)
        with torch.no_grad():                                        # keep the un-synchronised twin on the same weights
            for n in names:
                getattr(solo.mlp, n).params.copy_(getattr(dp.mlp, n).params)
        assert z1.bytes_reduced > 0 and z1.bytes_gathered == it * 4 * z1.param_padded.numel()
        for n in names:
            p = getattr(dp.mlp, n).params.detach()
            both = [torch.empty_like(p) for _ in range(world)]
            dist.all_gather(both, p.contiguous())
            assert torch.equal(both[0], both[1]), (rank, it, n)      # replicas stay bit-identical
    # validation frames: every rank renders its band of image rows, one all-gather puts the image together (networks._render_rows)
    Hh, Ww = 9, 8                                                     # 9 rows over 2 ranks: bands of 5 and 4 rows
    o, d = ops.gen_rays(poses[0], Hh, Ww, 11.0, 11.0, 0.5 * Ww, 0.5 * Hh, device=edev)
    frame = lambda: {'rays_o': o.clone(), 'rays_d': d.clone(), 'img_ids': torch.zeros((Hh * Ww, 1), dtype=torch.int32),
                     'src_shape': np.array([Hh, Ww, 3])}
    dp.set_val_pipeline(lambda q: frame())
    dp.sampler.k1_calls = 1000                                        # same march jitter stream on both ranks
    with torch.no_grad():
        out = dp.val_step({'poses': torch.zeros((1, 1, 4, 3)), 'images': torch.ones((1, 1, Hh, Ww, 4))})
    row0, nrows = xd.row_band(Hh, rank, world)
    dp.sampler.k1_calls = 1000
    with torch.no_grad():                                             # this rank's band on its own, same sampler state
        fr = frame()
        band = {k: (v[row0 * Ww:(row0 + nrows) * Ww] if torch.is_tensor(v) else v) for k, v in fr.items()}
        mine = dp.batchify_forward(band, is_test=True)['rgb'].reshape(nrows, Ww, 3)
    both = [torch.empty((5, Ww, 3)) for _ in range(world)]
    pad = torch.zeros((5, Ww, 3)); pad[:nrows] = mine
    dist.all_gather(both, pad)
    if rank == 0:
        img = torch.from_numpy(out['rgbs'][0])
        assert img.shape == (Hh, Ww, 3) and len(out['elapsed_time']) == 1
        assert torch.equal(img[:5], both[0][:5]) and torch.equal(img[5:], both[1][:4])      # rows 5.. came from rank 1
    else:
        assert out == {}
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


def test_data_parallel_fused_step_two_ranks_gloo(tmp_path):
    script = tmp_path / 'dp_worker.py'
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', XR_EMU_THREADS='4')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-3000:] for o in outs)
    assert all('ok' in o for o in outs)
