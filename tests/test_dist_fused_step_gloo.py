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
    names = ('embedder_pos', 'density_net', 'color_net')
    for it in (1, 2):
        b = Hn.batch(poses, N_RAYS, 100 * rank + it, edev)          # every rank its own rays
        grads = {}
        for tag, net in (('solo', solo), ('dp', dp), ('dp2', dp2)):
            net.sampler.set_iter(it)
            net.sampler.k1_calls = 7 * it + rank                    # same march jitter for both networks of this rank
            o = net.train_step({k: v.clone()[None] for k, v in b.items()}, None)
            for n in names:
                getattr(net.mlp, n).params.grad = None
            if net is dp2:
                torch.autograd.backward(o['loss'], grad_tensors=one)
            else:
                o['loss'].backward()
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
        for n in names:                                              # same update, bit for bit, without the scaling pass
            assert torch.equal(getattr(dp2.mlp, n).params.detach(), getattr(dp.mlp, n).params.detach()), (rank, it, n)
        with torch.no_grad():                                        # keep the un-synchronised twin on the same weights
            for n in names:
                getattr(solo.mlp, n).params.copy_(getattr(dp.mlp, n).params)
        for n in names:
            p = getattr(dp.mlp, n).params.detach()
            both = [torch.empty_like(p) for _ in range(world)]
            dist.all_gather(both, p.contiguous())
            assert torch.equal(both[0], both[1]), (rank, it, n)      # replicas stay bit-identical
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
