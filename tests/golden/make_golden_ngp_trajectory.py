"""Generates tests/golden/ref_ngp_trajectory.npz: 33 training iterations of the REFERENCE's own Instant-NGP stack --
HashNerfNetwork / NGPGridSampler / HashNerfMLP / HashNerfRender / HuberLoss from /root/reference, imported unmodified
(tests/golden/ref_import.py::load_ngp) on `raymarch_cuda` = xrnerf_amd.raymarch_cuda and `tinycudann` = xrnerf_amd.tcnn,
torch.optim.Adam with the config's hyper-parameters -- with the kernels executed on the host by tests/hip_emu.  Three
occupancy-grid refreshes (2 M-point density queries through the reference's run_density), two batch-size adaptations.
tests/test_gpu_trajectory.py replays the same iterations with this package's classes on the MI355X and compares.

    python tests/golden/make_golden_ngp_trajectory.py        (build container only; ~10 minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'hip_emu'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import emulib  # noqa: E402
import ngp_ref_harness as Hn  # noqa: E402
import ref_import  # noqa: E402

N_ITERS, N_RAYS0, TARGET = 33, 1024, 1 << 16


def init_weights(mlp):
    """the seeded initial parameters both sides start from (large enough for a structured density field)"""
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for name, lim in (('embedder_pos', 0.5), ('density_net', 0.3), ('color_net', 0.3)):
            p = getattr(mlp, name).params
            p.copy_(torch.empty(p.shape, dtype=torch.float32).uniform_(-lim, lim, generator=g).to(p.device))


def model_cfg():
    from xrnerf_amd.train import ngp_lego_model_cfg
    cfg = ngp_lego_model_cfg(n_rays=N_RAYS0)
    cfg['sampler']['target_batch_size'] = TARGET
    return cfg


def main():
    with emulib.emulated_ops() as dev:
        import xrnerf_amd.raymarch_cuda as rc
        import xrnerf_amd.tcnn as tc
        ref = ref_import.load_ngp(rc, tc)
        poses, alldata, info = Hn.scene()
        cfg = model_cfg()
        cfg.pop('type')
        net = ref.HashNerfNetwork(ref_import.Cfg(cfg.pop('cfg')), **{k: dict(v) for k, v in cfg.items()})
        init_weights(net.mlp)
        net.sampler.set_data(alldata, info)
        rc.reset_rng()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6)
        rec = {k: [] for k in ('n_rays', 'marched', 'compacted', 'loss', 'psnr', 'popcount', 'grid_mean')}
        bitfields = {}
        for it in range(N_ITERS):
            t0 = time.time()
            n_rays = int(net.sampler.n_rays_per_batch)
            b = Hn.batch(poses, n_rays, it, dev)
            net.sampler.set_iter(it)
            out = net.train_step({k: v[None] for k, v in b.items()}, opt)
            opt.zero_grad(set_to_none=True)
            out['loss'].backward()
            opt.step()
            s = net.sampler
            rec['n_rays'].append(n_rays)
            rec['marched'].append(int(s.rays_numsteps[:, 0].sum()))
            rec['compacted'].append(int(s.rays_numsteps_compacted[:, 0].sum()))
            rec['loss'].append(float(out['log_vars']['loss']))
            rec['psnr'].append(float(out['log_vars']['psnr']))
            bf = s.density_grid_bitfield.cpu().numpy()
            rec['popcount'].append(int(np.unpackbits(bf).sum()))
            rec['grid_mean'].append(float(s.density_grid_mean[0]))
            if it % 16 == 0:
                bitfields[it] = bf[:128 ** 3 // 8].copy()          # cascade 0 (aabb_scale = 1: the only active one)
            print('it %2d  rays %5d  marched %7d  loss %.5f  psnr %.3f  occupied %d  (%.1f s)' % (
                it, n_rays, rec['marched'][-1], rec['loss'][-1], rec['psnr'][-1], rec['popcount'][-1], time.time() - t0), flush=True)
        final = {n: getattr(net.mlp, n).params.detach().cpu().numpy() for n in ('density_net', 'color_net')}
        table = net.mlp.embedder_pos.params.detach().cpu().numpy()
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ref_ngp_trajectory.npz'),
                            n_iters=N_ITERS, n_rays0=N_RAYS0, target=TARGET, n_img=Hn.N_IMG,
                            **{k: np.asarray(v) for k, v in rec.items()},
                            bitfield_it0=bitfields[0], bitfield_it16=bitfields[16], bitfield_it32=bitfields[32],
                            final_density_net=final['density_net'], final_color_net=final['color_net'],
                            final_table_checksum=np.array([float(np.abs(table).sum()), float(table.sum())]),
                            final_table_head=table[:4096])


if __name__ == '__main__':
    main()
