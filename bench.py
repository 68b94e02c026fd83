#!/usr/bin/env python
"""Headline benchmark: Instant-NGP Lego training throughput (rays/s) on N MI355X, fp32.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full training iteration of configs/instant_ngp/nerf_blender_local01.py on one batch
of synthetic 800x800 Lego-shaped rays that are already resident in HBM: hooks -> batch slice + random
bg -> K1 ray march (+ the every-16th occupancy-grid refresh K6..K11 with its density queries) ->
hash-grid encode -> fused MLP -> K3 composite -> 5*Huber -> K4 -> MLP backward -> hash-grid scatter ->
[gradient all-reduce over RCCL when N>1] -> fused Adam(+EMA) over all 12.2 M parameters.  The batch is the
reference's adaptive one (4096 rays initially, re-sized every 16 iterations so that ~2^18 samples are
marched).  Weak scaling: every rank trains on its own rays, value = rays of all ranks / max-rank time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from xrnerf_amd import dist as xdist  # noqa: E402
from xrnerf_amd import ops  # noqa: E402
from xrnerf_amd.train import Trainer, render_frame, render_frame_ert  # noqa: E402

HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: fp32-input MFMA = the fp32 vector rate
MFMA_PEAK = {'f32': MFMA_F32_PEAK_TFLOPS, 'f16': 2500.0}   # dense peaks; f16 = v_mfma_f32_32x32x16_f16

# ALGORITHMIC work per unit (SURVEY.md section 8d, fp32 parity mode)
L_, F_ = 16, 2
ALGO = {
    # per sample: 12 B position + L*8 corners*F*4 B table reads + L*F*4 B features written
    'xr_hashgrid_fwd': ('hbm', 12 + L_ * 8 * F_ * 4 + L_ * F_ * 4),
    # per sample: 12 B position + L*F*4 B feature gradients + L*8*F atomic read-modify-writes (4+4 B)
    'xr_hashgrid_bwd': ('hbm', 12 + L_ * F_ * 4 + L_ * 8 * F_ * 8),
    # per sample: density 32*64+64*16, color 32*64+64*64+64*16 MACs = 10240 MAC = 20480 flop
    'xr_nerf_mlp_fwd': ('mfma', 20480),
    # recompute (without the color output layer) + dX chain + dW: 3 x 20480 - 2048
    'xr_nerf_mlp_bwd': ('mfma', 3 * 20480 - 2048),
    # per sample: 16 B raw + 4 B dt read (+ per ray 40 B, folded in as 3 B/sample at ~14 samples/ray)
    'xr_calc_rgb_forward': ('hbm', 20 + 3),
    'xr_calc_rgb_backward': ('hbm', 36 + 2),
    # K3 + Huber + K4 in one launch: per sample 16 B raw + 4 B dt read, 16 B written (+ per ray ~70 B folded as 3 B/sample)
    'xr_composite_train': ('hbm', 36 + 3),
    # per emitted sample 28 B written + per ray 36 B (folded: ~3 B/sample)
    # per RAY: 36 B in/out + 28 B per emitted sample at ~14 samples/ray
    'xr_rays_sampler': ('hbm', 36 + 28 * 14),
    # per PARAMETER: read p, g, m, v, ema (20 B) + write p, m, v, ema (16 B)
    'xr_adam_step': ('hbm', 36),
    # live-row list of the backward: dL/d(raw) [16 B] read by the count and by the fill pass, 4 B of list per live row (~0.46)
    'xr_live_rows': ('hbm', 16 + 16 + 2),
}


def cpu_vanilla_nerf(seconds_budget=8.0):
    """BASELINE config #1 on the host: the vanilla-NeRF pure-PyTorch path (xrnerf_amd/vanilla.py, validated
    against the reference's own modules in tests/test_vanilla_nerf.py), 1024-ray batch, 64 coarse + 128 fine
    samples, forward + backward, all host cores."""
    import xrnerf_amd
    from xrnerf_amd import vanilla
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ngp_model_cfg.json')))['vanilla_model']
    # cores actually usable by this process, capped: 256 oversubscribed threads ran this 20x SLOWER on the GPU box
    threads = min(32, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = xrnerf_amd.build_network(cfg)
    n = 1024
    rays_o = torch.tensor([[0., 0., 4.]]).repeat(n, 1)
    rays_d = torch.nn.functional.normalize(torch.randn(n, 3) * 0.15 - torch.tensor([0., 0., 1.]), dim=-1)
    tgt = torch.rand(n, 3)
    its, t0 = 0, None
    while True:
        z = vanilla.get_z_vals(rays_o, 2., 6., 64, randomized=True)
        data = {'rays_o': rays_o[None], 'rays_d': rays_d[None], 'viewdirs': rays_d[None], 'z_vals': z[None],
                'pts': vanilla.get_pts(rays_o, rays_d, z)[None], 'target_s': tgt[None]}
        net.zero_grad()
        net.train_step(data, None)['loss'].backward()
        if t0 is None:
            t0 = time.time()          # first iteration = warm-up
            continue
        its += 1
        if time.time() - t0 > seconds_budget or its >= 20:
            break
    el = time.time() - t0
    return {'value': its * n / el, 'unit': 'rays/s', 'cores': threads, 'kind': 'port',
            'sample': '%d forward+backward iterations of configs/nerf/nerf_blender_base01.py (1024 rays, 64+128 samples, '
                      '2 x 8x256 MLP), pure PyTorch on the host' % its}


def mipnerf_config3(dev, steps=20, warmup=5, cpu_seconds=10.0):
    """Secondary line (BASELINE config #3): Mip-NeRF multiscale training step -- 1024 rays,
    128 + 128 conical-frustum samples, one shared 8x256 MLP -- with the sampling / IPE / render / resample stages as
    single HIP launches (xrnerf_amd/csrc/xr_mip.hip) and the MLP on rocBLAS; beside it the same step in pure PyTorch on
    the host cores (oracle/mip_oracle.py::torch_train_step, pinned to the reference's code)."""
    import xrnerf_amd
    from xrnerf_amd import mip
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'mip_model_cfg.json')))
    R, S = cfg['N_rand_per_sampler'], cfg['num_samples']
    torch.manual_seed(0)
    net = xrnerf_amd.build_network(cfg['model']).to(dev)
    # the config's Adam (torch.optim.Adam's update rule and defaults) as this package's multi-tensor launches: four tensors per launch
    # instead of torch's chain of foreach kernels over the 44 parameter tensors
    from xrnerf_amd.train import FusedAdam
    opt = FusedAdam(list(net.parameters()), lr=cfg['optimizer']['lr'], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_momentum=None)
    rays = mip.synthetic_multiscale_rays(R, dev, seed=1)

    def step():
        data = mip.get_z_vals(dict(rays), S + 1, randomized=True)
        out = net.train_step({k: v[None] for k, v in data.items()}, opt)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
        return out

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer(only={'xr_mip_encode', 'xr_mip_render_forward', 'xr_mip_render_backward', 'xr_mip_resample'})
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    summ = timer.summary()
    ch = net.mlp.input_ch + net.mlp.input_ch_dirs
    # algorithmic HBM bytes per launch: encode writes ch floats per sample and reads 44 B + (S+1)*4 B
    # per ray; render fwd reads raw 16 B and writes the weight 4 B per sample (+ per ray z 4(S+1), d 12, out 20);
    # render bwd reads raw 16 B, writes 16 B per sample; resample reads z and weights, writes z per ray
    algo = {'xr_mip_encode': R * S * ch * 4 + R * (44 + (S + 1) * 4),
            'xr_mip_render_forward': R * S * 20 + R * ((S + 1) * 4 + 32),
            'xr_mip_render_backward': R * S * 32 + R * ((S + 1) * 4 + 24),
            'xr_mip_resample': R * ((S + 1) * 12 + S * 4)}
    kern = {}
    for k, (n, ms, _) in summ.items():
        us = ms * 1e3 / max(n, 1)
        kern[k] = {'avg_launch_us': us, 'launches': n, 'algorithmic_bytes_per_launch': algo[k],
                   'achieved_GBs': algo[k] / (us * 1e-6) / 1e9, 'frac_of_hbm_peak': algo[k] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    res = {'workload': 'Mip-NeRF multiscale (configs/mipnerf/mipnerf_multiscale.py): %d rays x (%d + %d) samples, 8x256 MLP, '
                       'forward + backward + Adam, synthetic multiscale rays resident in HBM' % (R, S, S),
           'value': R * steps / el, 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'steps': steps, 'dtype': 'f32',
           'final_loss': float(out['log_vars']['loss']), 'kernels': kern}
    # the same step on the host cores, pure PyTorch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import mip_oracle as MO
    from xrnerf_amd import vanilla
    threads = min(32, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    mcfg = dict(cfg['model']['mlp']); mcfg.pop('type')
    mlp = vanilla.NerfMLP(**mcfg)
    copt = torch.optim.Adam(mlp.parameters(), lr=cfg['optimizer']['lr'])
    cdata = {k: v.cpu() for k, v in rays.items()}
    r = cfg['model']['render']
    rkw = dict(density_bias=r['density_bias'], rgb_padding=r['rgb_padding'], white_bkgd=r['white_bkgd'],
               activation=r['density_activation'])
    its, t0 = 0, None
    while True:
        zr = torch.rand(R, S + 1)
        cdata['z_vals'] = torch.from_numpy(MO.z_vals(cdata['near'].numpy(), cdata['far'].numpy(), S + 1, False, zr.numpy()))
        loss, _ = MO.torch_train_step(mlp, cdata, render_kw=rkw)
        copt.zero_grad(set_to_none=True)
        loss.backward()
        copt.step()
        if t0 is None:
            t0 = time.perf_counter()          # first iteration = warm-up
            continue
        its += 1
        if time.perf_counter() - t0 > cpu_seconds or its >= 8:
            break
    cel = time.perf_counter() - t0
    res['cpu_baseline'] = {'value': its * R / cel, 'unit': 'rays/s', 'cores': threads, 'kind': 'port',
                           'sample': '%d iterations of the same step (1024 rays x 256 samples), pure PyTorch on the host '
                                     '(oracle/mip_oracle.py::torch_train_step)' % its}
    return res


def kilonerf_config5(dev, frames=8, cpu_seconds=10.0):
    """Secondary line (BASELINE config #5): KiloNeRF real-time rendering -- 800x800 rays x 384
    samples through 1440 tiny MLPs on the Lego grid (synthetic occupancy ~7 %, random weights), the reference's test
    path (GetZvals + GetPts + KiloNerfMLP + NerfRender) as one C-ABI call; beside it the numpy oracle of the same path
    on a bounded sample of the same rays, one host thread."""
    from xrnerf_amd import kilo
    mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=1)
    H = W = 800
    focal = 1111.111
    poses = kilo.orbit_poses(6)
    for _ in range(2):
        kilo.render_frame(mlp, gmin, gmax, poses[0], H, W, focal)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        rgb, disp, acc = kilo.render_frame(mlp, gmin, gmax, poses[f % len(poses)], H, W, focal)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    # evaluated samples of one frame (per-network counts of the module-level path) -> flops of the tiny MLPs
    rays_o, rays_d, viewdirs = kilo.camera_rays(poses[0], H, W, focal, dev)
    z = ops.mip_zvals(torch.full((H * W,), 2.0, device=dev), torch.full((H * W,), 6.0, device=dev), 384)
    _, counts = ops.kilo_mlp_forward(viewdirs, mlp._host3(gmin), mlp._host3(gmax), [x // 16 for x in mlp.resolution],
                                     mlp.resolution, mlp.occupancy_grid, mlp.domain_mins, mlp.domain_maxs,
                                     mlp.multi_network.packed(), 10, 4, 2, rays_o=rays_o, rays_d=rays_d, z_vals=z, want_counts=True)
    evaluated = int(counts.sum())
    flop_per_sample = 2 * (63 * 32 + 32 * 32 + 32 * 33 + 59 * 32 + 32 * 3)
    res = {'workload': 'KiloNeRF Lego grid (configs/kilonerf/kilonerf_finetune_Synthetic_NeRF_base01.py): 800x800 rays x 384 samples, '
                       '1440 networks (9x16x10) of 2x32 hidden units, 144x256x160 occupancy, synthetic scene and weights',
           'value': ms, 'unit': 'ms per 800x800 frame (ray generation included)', 'higher_is_better': False, 'frames': frames,
           'dtype': 'f32', 'evaluated_samples_per_frame': evaluated, 'networks_used': int((counts > 0).sum()),
           'tiny_mlp_gflop_per_frame': evaluated * flop_per_sample / 1e9,
           'reference_published_ms_per_frame_other_hw': 365.16}
    # fine-tuning step of configs/kilonerf/kilonerf_finetune_Synthetic_NeRF_base01.py: 8192 rays x 384 samples through
    # KiloNerfMLP.forward under autograd (xr_kilo_mlp_forward) and its backward (xr_kilo_mlp_backward = the reference's
    # six AddMultiMatMul.backward's), timed with events on the launch stream
    try:
        nft = 8192
        sel = torch.randperm(H * W, device=dev)[:nft]
        data = {'rays_o': rays_o[sel].contiguous(), 'rays_d': rays_d[sel].contiguous(), 'viewdirs': viewdirs[sel].contiguous(),
                'z_vals': z[sel].contiguous(), 'global_domain_min': gmin, 'global_domain_max': gmax}
        for p_ in mlp.multi_network.parameters():
            p_.requires_grad_(True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        reps = 6
        for r in range(reps + 2):
            for p_ in mlp.multi_network.parameters():
                p_.grad = None
            ev[0].record()
            raw = mlp(dict(data))['raw']
            ev[1].record()
            raw.backward(torch.ones_like(raw))
            ev[2].record()
            torch.cuda.synchronize()
            if r >= 2:
                tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        _, cft = ops.kilo_mlp_forward(data['viewdirs'], mlp._host3(gmin), mlp._host3(gmax), [x // 16 for x in mlp.resolution],
                                      mlp.resolution, mlp.occupancy_grid, mlp.domain_mins, mlp.domain_maxs,
                                      mlp.multi_network.packed(), 10, 4, 2, rays_o=data['rays_o'], rays_d=data['rays_d'],
                                      z_vals=data['z_vals'], want_counts=True)
        ev_ft = int(cft.sum())
        res['finetune_step'] = {'rays': nft, 'samples_per_ray': 384, 'evaluated_samples': ev_ft,
                                'forward_ms': tf / reps, 'backward_ms': tb / reps,
                                'backward_includes': 'ones-like dL/draw fill, xr_kilo_mlp_backward, unpacking the packed gradient blocks',
                                'backward_tflops': ev_ft * 3 * flop_per_sample / (tb / reps * 1e-3) / 1e12,
                                'forward_tflops': ev_ft * flop_per_sample / (tf / reps * 1e-3) / 1e12}
        for p_ in mlp.multi_network.parameters():
            p_.requires_grad_(False); p_.grad = None
    except Exception as e:  # noqa: BLE001
        res['finetune_step'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    del z, rays_o, rays_d, viewdirs
    torch.cuda.empty_cache()
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import kilo_oracle as KO
    nets = KO.TinyNets(*[[v.cpu().numpy() for v in grp] if isinstance(grp, list) else grp.cpu().numpy() for grp in (
        [l.weight.detach() for l in mlp.multi_network.pts_linears], [l.bias.detach() for l in mlp.multi_network.pts_linears],
        mlp.multi_network.alpha_linear.weight.detach(), mlp.multi_network.alpha_linear.bias.detach(),
        mlp.multi_network.feature_linear.weight.detach(), mlp.multi_network.feature_linear.bias.detach(),
        mlp.multi_network.direction_layer.weight.detach(), mlp.multi_network.direction_layer.bias.detach(),
        mlp.multi_network.rgb_linear.weight.detach(), mlp.multi_network.rgb_linear.bias.detach())])
    ro, rd, vd = (t.cpu().numpy() for t in kilo.camera_rays(poses[0], H, W, focal, 'cpu'))
    occ, dmn, dmx = mlp.occupancy_grid.cpu().numpy(), mlp.domain_mins.cpu().numpy(), mlp.domain_maxs.cpu().numpy()
    done, t0 = 0, time.perf_counter()
    rows = np.arange(0, H * W, 41)                       # a strided sample of the frame's rays, 2048 at a time
    zc = np.tile(np.linspace(2.0, 6.0, 384, dtype=np.float32), (2048, 1))
    while done < rows.size and time.perf_counter() - t0 < cpu_seconds:
        sel = rows[done:done + 2048]
        raw, _, _, _ = KO.mlp_raw(ro[sel], rd[sel], vd[sel], zc[:sel.size], gmin.numpy(), gmax.numpy(), [9, 16, 10], mlp.resolution,
                                  occ, dmn, dmx, nets)
        KO.nerf_render(raw, zc[:sel.size], rd[sel], True)
        done += sel.size
    cel = time.perf_counter() - t0
    res['cpu_baseline'] = {'value': cel / max(done, 1) * H * W * 1e3, 'unit': 'ms per 800x800 frame (extrapolated)', 'cores': 1,
                           'kind': 'port', 'sample': '%d rays (every 41st of the frame) x 384 samples, oracle/kilo_oracle.py '
                                                      '(numpy), ray generation excluded' % done}
    return res


def ngp_f16_mlp_mode(dev, n_img, steps):
    """Second line in the reference's own arithmetic (tiny-cuda-nn runs FullyFusedMLP in fp16 with fp32 accumulation,
    hashnerf_mlp.py:76-77): the same training iterations with the fused MLP on v_mfma_f32_32x32x16_f16 (fp32 tables,
    parameters, gradients and outputs; xrnerf_amd.ops.set_precision('f16')).  The headline stays the fp32 parity mode."""
    ops.set_precision('f16')
    try:
        tr = Trainer(dev, n_img=n_img)
        sampler = tr.net.sampler
        pre, hist = 0, [sampler.n_rays_per_batch]
        while pre < PREROLL_MAX:
            tr.run(16)
            pre += 16
            hist.append(sampler.n_rays_per_batch)
            if pre >= PREROLL_MIN and len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= 0.02 * hist[-2] and abs(hist[-2] - hist[-3]) <= 0.02 * hist[-3]:
                break
        freq = sampler.update_grid_freq
        want = max(1, int(round(steps / float(freq))))
        first_off = max(0, min(freq - 1, (steps - 1 - (want - 1) * freq) // 2))
        align = ((-tr.iter) % freq - first_off) % freq
        if align:
            tr.run(align)
        torch.cuda.synchronize()
        r0, s0, it0 = tr.rays_done, tr.samples_done, tr.iter
        t0 = time.perf_counter()
        tr.run(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rays, samples = tr.rays_done - r0, tr.samples_done - s0
        s1 = tr.samples_done
        ops.TIMER = ops.KernelTimer(only={'xr_nerf_mlp_fwd', 'xr_nerf_mlp_bwd'}, train_only=True)
        for _ in range(32):
            tr.step()
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None
        s_win = tr.samples_done - s1
        kern = {}
        for k, (n_l, ms_l, _) in timer.summary().items():
            fl = ALGO[k][1] * s_win
            kern[k] = {'avg_launch_us': ms_l * 1e3 / max(n_l, 1), 'achieved_TFLOPs': fl / (ms_l * 1e-3) / 1e12,
                       'peak_TFLOPs': MFMA_PEAK['f16'], 'frac': fl / (ms_l * 1e-3) / 1e12 / MFMA_PEAK['f16']}
        # what the precision costs in the image: the same weights rendered in both modes
        H = W = 800
        rgb16, _ = render_frame(tr.net, tr.data.poses[0], H, W, tr.data.focal)
        k1 = tr.net.sampler.k1_calls
        ops.set_precision('f32')
        tr.net.sampler.k1_calls = k1 - 1                    # same jitter as the fp16 frame
        rgb32, _ = render_frame(tr.net, tr.data.poses[0], H, W, tr.data.focal)
        dev_rgb = (rgb16 - rgb32).abs()
        return {'workload': 'the headline iterations with the fused MLP in fp16 (fp32 accumulate), %d timed iterations %d..%d after %d '
                            'pre-roll iterations' % (steps, it0, it0 + steps - 1, pre),
                'dtype': 'f16 MLP operands / f32 accumulate, f32 hash tables, parameters, gradients, compositor',
                'value': rays / el, 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'rays_per_step': rays / steps,
                'samples_per_ray': samples / max(rays, 1), 'kernels': kern,
                'rendered_rgb_deviation_vs_f32': {'max_abs': float(dev_rgb.max()), 'mean_abs': float(dev_rgb.mean())},
                'final_train_psnr': float(tr.step()['log_vars']['psnr'])}
    finally:
        ops.set_precision('f32')


def ngp_tcnn_strict_defaults(dev, n_img, steps):
    """The topology tiny-cuda-nn would build from the reference's UNCHANGED config if it ignores the config's `num_layers` key
    (its key is `n_hidden_layers`, default 5): 5-hidden-layer density and colour nets, 77 824 flop per
    sample forward.  Round 5: the streamed fused kernels (k_nerf_mlp_fwd_deep / _bwd_deep: the layers' weights pass through LDS) inside
    the same native step and loop as the headline; timed at THIS network's adaptive fixed point (pre-roll like the headline's)."""
    os.environ['XRNERF_TCNN_STRICT_DEFAULTS'] = '1'
    try:
        tr = Trainer(dev, n_img=n_img)
        assert tr.net.mlp.density_net.n_hidden == 5 and tr.net.mlp.color_net.n_hidden == 5
        sampler = tr.net.sampler
        pre, hist = 0, [sampler.n_rays_per_batch]
        while pre < PREROLL_MAX:
            tr.run(16)
            pre += 16
            hist.append(sampler.n_rays_per_batch)
            if pre >= PREROLL_MIN and len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= 0.02 * hist[-2] and abs(hist[-2] - hist[-3]) <= 0.02 * hist[-3]:
                break
        tr.run((-tr.iter) % sampler.update_grid_freq + 1)
        torch.cuda.synchronize()
        if ops.LIVE_STATS is not None:
            ops.LIVE_STATS[1:3].zero_()
        r0, s0, it0 = tr.rays_done, tr.samples_done, tr.iter
        t0 = time.perf_counter()
        tr.run(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rays, samples = tr.rays_done - r0, tr.samples_done - s0
        live_frac = 1.0
        if ops.LIVE_STATS is not None:
            lv, vd = (int(v) & 0xffffffff for v in ops.LIVE_STATS[1:3].tolist())
            live_frac = lv / vd if vd else 1.0
        native = tr._loop is not None and tr._loop.enqueued > 0
        mac = 32 * 64 + 4 * 64 * 64 + 64 * 16
        flop = {'xr_nerf_mlp_fwd': 2 * mac * 2, 'xr_nerf_mlp_bwd': 3 * 2 * mac * 2 - 2 * 64 * 16}
        kern = {}
        for name in ('xr_nerf_mlp_fwd', 'xr_nerf_mlp_bwd'):
            ops.TIMER = ops.KernelTimer(only={name}, train_only=True)
            tr.run((-tr.iter) % sampler.update_grid_freq + 1)      # (past the next refresh: the timed launches are the training step's)
            tr.run(15)
            torch.cuda.synchronize()
            timer, ops.TIMER = ops.TIMER, None
            n_l, ms_l, _ = timer.summary()[name]
            units = (samples / steps) * (live_frac if name == 'xr_nerf_mlp_bwd' else 1.0) * n_l
            fl = flop[name] * units
            kern[name] = {'avg_launch_us': ms_l * 1e3 / max(n_l, 1), 'launches': n_l, 'achieved_TFLOPs': fl / (ms_l * 1e-3) / 1e12,
                          'peak_TFLOPs': MFMA_PEAK['f16'], 'frac': fl / (ms_l * 1e-3) / 1e12 / MFMA_PEAK['f16'],
                          'issued_frac': 3.0 * fl / (ms_l * 1e-3) / 1e12 / MFMA_PEAK['f16'], 'flop_per_sample': flop[name],
                          'units': 'marched samples' if name == 'xr_nerf_mlp_fwd' else 'live rows (fraction %.3f)' % live_frac}
        return {'workload': 'the headline iterations with 5 + 5 hidden layers (XRNERF_TCNN_STRICT_DEFAULTS=1: what tcnn builds if it ignores '
                            'the config\'s num_layers key), %d timed iterations %d..%d after %d pre-roll iterations at this network\'s own '
                            'adaptive fixed point; streamed fused MLP kernels (fp32 results on the 16-bit matrix cores: forward and recompute '
                            'on 2-way split fp16 operands, gradients on 2-way split bf16 operands), native step and loop: %s'
                            % (steps, it0, it0 + steps - 1, it0, 'yes' if native else 'NO'),
                'value': rays / el, 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'rays_per_step': rays / steps,
                'samples_per_ray': samples / max(rays, 1), 'backward_live_row_fraction': live_frac, 'rays_per_batch_history': hist[-6:],
                'kernels': kern, 'final_train_psnr': float(tr.step()['log_vars']['psnr'])}
    finally:
        os.environ.pop('XRNERF_TCNN_STRICT_DEFAULTS', None)


def ngp_real_lego_fixture(dev, steps=64):
    """The reference's own 5-image Lego fixture (4 training views; staged under oracle/_ref/data by tools/stage_ref_lego.py): does the
    headline hinge on the synthetic boxes?  Rays/s and the share of live backward rows on real pixels after 512 iterations."""
    from xrnerf_amd import datasets
    datadir = os.path.join(ROOT, 'oracle', '_ref', 'data', 'lego')
    if not os.path.isdir(datadir):
        return {'skipped': 'fixture not staged (%s)' % datadir}
    ds = datasets.HashNerfDataset(dict(datadir=datadir, half_res=False, testskip=1, white_bkgd=False, load_alpha=True,
                                       N_rand_per_sampler=4096, mode='train', val_n=2), device=dev)
    tr = Trainer(dev, dataset=ds)
    tr.run(512)
    tr.run((-tr.iter) % tr.net.sampler.update_grid_freq + 1)
    if ops.LIVE_STATS is not None:
        ops.LIVE_STATS[1:3].zero_()
    torch.cuda.synchronize()
    r0, s0 = tr.rays_done, tr.samples_done
    t0 = time.perf_counter()
    tr.run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    live, valid = (int(v) & 0xffffffff for v in ops.LIVE_STATS[1:3].tolist()) if ops.LIVE_STATS is not None else (0, 0)
    rays, samples = tr.rays_done - r0, tr.samples_done - s0
    return {'workload': 'reference test fixture nerf_synthetic/lego, %d training views 800x800, iterations %d..%d' % (ds.n_img, tr.iter - steps, tr.iter - 1),
            'value': rays / el, 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'rays_per_step': rays / steps,
            'samples_per_ray': samples / max(rays, 1), 'live_row_fraction': live / valid if valid else None,
            'train_psnr_batch': float(tr.step()['log_vars']['psnr']),
            'note': 'a 4-view over-fit check of the pipeline on real pixels, not comparable with the reference\'s 100-view 35.1 dB'}


def registry_frame_ms(tr, H=800, W=800, frames=6):
    """The frame time a user of the unchanged config gets: HashNerfNetwork.val_step (networks/hashnerf.py:54-93 semantics: per frame
    the pipeline's ray generation, batchify_forward in chunk = 4096 pieces -- 157 chunks -- and the device-to-host copy of the image
    inside the timer)."""
    net, data = tr.net, tr.data
    focal = data.focal * H / data.H

    def pipeline(q):
        o, d = ops.gen_rays(q['pose'], H, W, focal, focal, 0.5 * W, 0.5 * H, device=tr.device)
        return {'rays_o': o, 'rays_d': d, 'img_ids': torch.full((o.shape[0], 1), float(q['idx']), dtype=torch.float32, device=tr.device),
                'src_shape': np.array([H, W, 3])}
    net.set_val_pipeline(pipeline)
    poses = np.stack([data.poses[k % data.n_img] for k in range(frames + 2)])[None]
    images = torch.ones((1, frames + 2, H, W, 4), dtype=torch.float32)
    out = {}
    for key, env in (('render_ms_per_800x800_frame_registry_chunk4096', 'one_launch'), ('render_ms_per_800x800_frame_registry_chunk4096_loop_of_157_chunks', 'async')):
        os.environ['XRNERF_FRAME'] = env
        try:
            with torch.no_grad():
                r = net.val_step({'poses': poses, 'images': images})
            ts = np.array(r['elapsed_time'][2:]) * 1e3                      # the first two frames size the persistent buffers
            out[key] = float(np.median(ts))
            out[key + '_mean_min_max'] = [float(ts.mean()), float(ts.min()), float(ts.max())]
        finally:
            os.environ.pop('XRNERF_FRAME', None)
    out['registry_chunk'] = int(net.chunk)
    out['registry_frame_note'] = ('median over %d frames of val_step\'s own per-frame timer (pipeline + batchify_forward + device-to-host copy); between two '
                                  'frames val_step multiplies the image by its alpha on the host (~5 ms of numpy, outside the timer, as in the reference), '
                                  'the GPU idles meanwhile and single frames take up to 4x the median (clock ramp)' % frames)
    return out


def value_all_rows(args):
    """the headline once more with the backward over EVERY marched row (XR_MLP_LIVE=0, read once per process: a child process)"""
    import subprocess
    env = dict(os.environ, XR_MLP_LIVE='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup), '--n-img', str(args.n_img),
                        '--headline-only'], env=env, capture_output=True, text=True, timeout=1200)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not line:
        raise RuntimeError('child failed: %s' % r.stderr[-300:])
    d = json.loads(line[-1])
    res = {'value': d['value'], 'ms_per_step': d['ms_per_step'], 'note': 'MLP backward and table scatter over every marched row (XR_MLP_LIVE=0), same results'}
    if d.get('dominant') == 'xr_hashgrid_bwd' and d.get('dominant_avg_launch_us'):
        # the dominant entry point's roofline in THIS mode -- the table scatter over every marched row (live fraction 1) with the update inside:
        # 2188 B per row + 32 B per table parameter, like the headline's `roofline`; three times the rows, the same fixed part
        by = d['samples_per_step'] * ALGO['xr_hashgrid_bwd'][1] + 32.0 * d['table_params']
        gbs = by / (d['dominant_avg_launch_us'] * 1e-6) / 1e9
        res['roofline'] = {'kernel': 'xr_hashgrid_bwd', 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                           'avg_launch_us': d['dominant_avg_launch_us'], 'launches': d['dominant_launches'],
                           'algorithmic_bytes_or_flops_per_launch': by, 'live_row_fraction': 1.0}
    return res


def ngp_config4_unbounded(dev, steps=64):
    """Secondary line (BASELINE config #4): the same Instant-NGP model on an UNBOUNDED forward-facing scene, 1008 x 756,
    aabb_scale = 16 (five occupancy cascades: 10.5 M-point grid queries below iteration 256, 2 x 2.6 M after), synthetic
    fern-shaped cameras and geometry (xrnerf_amd.train.SyntheticFern; the reference ships no such config).  Training
    rays/s at the adaptive fixed point + ms per 1008x756 frame on this GPU's row band (all rows at N = 1)."""
    from xrnerf_amd.train import SyntheticFern
    data = SyntheticFern(dev, n_img=20)
    tr = Trainer(dev, dataset=data)
    sampler = tr.net.sampler
    pre, hist = 0, [sampler.n_rays_per_batch]
    while pre < PREROLL_MAX:
        for _ in range(16):
            tr.step()
        pre += 16
        hist.append(sampler.n_rays_per_batch)
        # (no early stop here: this scene's batch size keeps growing for ~1000 iterations as the far cascades empty -- stopping at
        # the first 2 % plateau gave 2.2 M rays/s at 91 samples/ray in one run and 9.2 M at 20 in the next, the same ~190 M samples/s)
    torch.cuda.synchronize()
    r0, s0 = tr.rays_done, tr.samples_done
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rays, samples = tr.rays_done - r0, tr.samples_done - s0
    H, W = data.H, data.W
    for _ in range(2):
        render_frame(tr.net, data.poses[0], H, W, data.focal)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for f in range(3):
        rgb, alpha = render_frame(tr.net, data.poses[f], H, W, data.focal)
    torch.cuda.synchronize()
    ms_frame = (time.perf_counter() - t1) * 1e3 / 3
    return {'workload': 'Instant-NGP, unbounded forward-facing synthetic scene (fern-shaped), 1008x756, 20 images, aabb_scale 16 '
                        '(max_cascade 4), same model / sampler / kernels as the headline; %d timed iterations after %d pre-roll '
                        'iterations (%d grid refreshes in the window)' % (steps, pre, steps // 16),
            'value': rays / el, 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'dtype': 'f32',
            'rays_per_step': rays / steps, 'samples_per_ray': samples / max(rays, 1), 'samples_per_s': samples / el,
            'rays_per_batch_history': hist[:8] + ['...'] + hist[-4:] if len(hist) > 14 else hist,
            'render_ms_per_1008x756_frame': ms_frame, 'render_samples_per_ray': float(sampler.coords.shape[0]) / (H * W),
            'occupied_cells_per_cascade': [int(np.unpackbits(sampler.density_grid_bitfield[c * 262144:(c + 1) * 262144].cpu().numpy()).sum())
                                           for c in range(5)]}


def _cpu_worker(seed, seconds_budget):
    """one ray shard of the CPU baseline: the oracle's plain-C port of the SAME training iteration, one thread"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    from xrnerf_amd import synthetic as S
    O.set_threads(1)
    meta = O.GridMeta()
    grid = S.lego_density_grid()
    bf = O.bitfield_given_mean(grid, O.density_mean(grid))
    poses = S.lego_cameras(20)
    table = S.hash_table(meta.n_params)
    wd, wc = S.mlp_weights(32, 64, 1, 16, 4), S.mlp_weights(32, 64, 2, 16, 5)
    params = [table, wd, wc]
    ms = [np.zeros_like(p) for p in params]
    vs = [np.zeros_like(p) for p in params]
    n_rays, rays_done, it = 1024, 0, 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    while True:
        o, d, _ = S.training_rays(poses, n_rays, seed=100 + 1000 * seed + it)
        tgt = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
        bg = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
        coords, _, ns, cnt = O.rays_sampler(o, d, bf, rng_calls=it, max_samples=n_rays * 256)
        s = int(cnt[1])
        c = coords[:s]
        pts, dirs = np.ascontiguousarray(c[:, :3]), np.ascontiguousarray(c[:, 4:])
        raw = O.nerf_mlp_fwd(table, wd, wc, pts, dirs, meta)
        rgb = O.calc_rgb_forward(raw, c, ns, ns, bg)
        _, g = O.huber_loss_grad(rgb, tgt)
        draw = O.calc_rgb_backward(raw, ns, c, g, rgb, 0.05)
        gt, gd, gc = O.nerf_mlp_bwd(table, wd, wc, pts, dirs, draw, meta)
        for p, gr, m, v in zip(params, (gt, gd, gc), ms, vs):
            O.adam(p, gr, m, v, it + 1)
        it += 1
        rays_done += n_rays
        el = time.time() - t0
        if el > seconds_budget or it >= 64:
            break
    return rays_done, el, it


def cpu_baseline(seconds_budget=15.0):
    """The oracle's plain-C port of the SAME training iteration on the host cores of this box, RAY-SHARDED: one
    single-threaded worker process per core (capped at 64: every worker streams its own 12.2 M-parameter Adam state,
    ~250 MB, and the socket's memory bandwidth saturates long before 256 workers), each on its own 1024-ray batches of
    the same synthetic workload; value = rays of all workers / slowest worker's time."""
    import subprocess
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    procs = max(1, min(avail, 64))
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='')
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(k), '--cpu-seconds', str(seconds_budget)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for k in range(procs)]
    rays, els, its = 0, [], 0
    for p_ in ps:
        out, _ = p_.communicate(timeout=seconds_budget * 6 + 120)
        r = json.loads(out.strip().splitlines()[-1])
        rays += r['rays']; els.append(r['seconds']); its += r['iterations']
    el = max(els)
    return {'value': rays / el, 'unit': 'rays/s', 'cores': procs, 'kind': 'port', 'host_cores_available': avail,
            'single_core_rays_per_s': rays / sum(els),
            'sample': '%d worker processes x 1 thread (ray-sharded), %d training iterations of 1024 rays in total '
                      '(K1+encode+MLP+K3+Huber+K4+backward+Adam over 12.2M params each), oracle/ngp_oracle.c, '
                      'ray generation excluded' % (procs, its)}


PREROLL_MIN, PREROLL_MAX = 272, 1024      # past iteration 256: the reference's schedule samples M/4 + M/4 grid cells from there on


def _respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL),
    exactly as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` line would."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=256)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--n-img', type=int, default=100, help='training cameras of the synthetic scene (SURVEY.md section 8d: 100; the ray table is 28 MB per image)')
    ap.add_argument('--no-preroll', action='store_true', help='time right after --warmup (the occupancy grid is still dense '
                    'and the adaptive batch has not converged: NOT the steady state)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-render', action='store_true')
    ap.add_argument('--no-mip', action='store_true', help='skip the secondary Mip-NeRF (config #3) line')
    ap.add_argument('--no-kilo', action='store_true', help='skip the secondary KiloNeRF (config #5) line')
    ap.add_argument('--no-unbounded', action='store_true', help='skip the secondary unbounded-scene (config #4) line')
    ap.add_argument('--no-f16', action='store_true', help='skip the second line in the reference\'s fp16 MLP precision')
    ap.add_argument('--no-strict', action='store_true', help='skip the secondary line with tcnn\'s default 5 + 5 hidden layers')
    ap.add_argument('--no-extra', action='store_true', help='skip value_all_rows, the real-Lego fixture line and the registry frame time')
    ap.add_argument('--headline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-worker', type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:           # one shard of cpu_baseline(); no GPU, no torch.distributed
        rays, el, it = _cpu_worker(args.cpu_worker, args.cpu_seconds)
        print(json.dumps({'rays': rays, 'seconds': el, 'iterations': it}))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _respawn_under_torchrun(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    # N > 1 over RCCL: keep what RCCL says about the rings / trees, channels and protocols it chose (its INFO log of rank 0, digested into
    # the line's `collective.rccl_log` -- whether the 48.8-MB all-reduce runs as a ring or direct decides the 8-GPU number: dist.comm_model)
    rccl_log = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('XRNERF_DIST_BACKEND', 'nccl') == 'nccl' and os.environ.get('RANK', '0') == '0':
        rccl_log = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'xrnerf_rccl_rank0_%d.log' % os.getpid())
        os.environ.setdefault('NCCL_DEBUG', 'INFO')
        os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING')
        os.environ.setdefault('NCCL_DEBUG_FILE', rccl_log)
    # XRNERF_DIST_BACKEND=gloo + XRNERF_SHARE_GPU=1: protocol test of the N>1 path on a single-GPU box
    rank, local, world = xdist.init_from_env(os.environ.get('XRNERF_DIST_BACKEND', 'nccl'))
    if os.environ.get('XRNERF_SHARE_GPU') == '1':
        local = 0
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    tr = Trainer(dev, n_img=args.n_img, world_size=world, rank=rank)
    sampler = tr.net.sampler
    ops.TIMER = None
    # ---- un-timed pre-roll to the adaptive fixed point, whatever --warmup says: the first iterations march a dense
    # occupancy grid with 4096-ray batches (59 samples/ray); the reference's own schedule re-sizes the batch every 16
    # iterations towards 2^18 samples.  Steady state = at least PREROLL_MIN iterations AND the batch size changed by
    # less than 2 % at two consecutive adaptations.  Every rank takes the same number of iterations (all-reduced).
    preroll, hist = 0, [sampler.n_rays_per_batch]
    if not args.no_preroll:
        while preroll < PREROLL_MAX:
            tr.run(16)
            preroll += 16
            hist.append(sampler.n_rays_per_batch)
            stable = (len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= 0.02 * hist[-2]
                      and abs(hist[-2] - hist[-3]) <= 0.02 * hist[-3])
            flag = torch.tensor([1.0 if (preroll >= PREROLL_MIN and stable) else 0.0], device=dev)
            if world > 1:
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if float(flag) > 0:
                break
    if args.warmup:
        tr.run(args.warmup)
    # the window's phase against the every-16th grid refresh is fixed, not left to --warmup: K timed steps contain
    # round(K / 16) refresh iterations (at least one), the nearest integer to the long-run share; the count is reported
    freq = sampler.update_grid_freq
    want = max(1, int(round(args.steps / float(freq))))
    # the first refresh falls `first_off` steps into the window, the slack split evenly before the first and after the last
    first_off = max(0, min(freq - 1, (args.steps - 1 - (want - 1) * freq) // 2))
    align = ((-tr.iter) % freq - first_off) % freq            # un-timed iterations that put the window at that phase
    if align:
        tr.run(align)
    # which entry point dominates the step?  16 un-timed iterations with events around every training launch (the phase
    # alignment above is redone afterwards); the timed region then carries events around THAT kernel only
    ops.TIMER = ops.KernelTimer(only=set(ALGO), train_only=True)
    for _ in range(freq):
        tr.step()
    torch.cuda.synchronize()
    pre_summ, ops.TIMER = ops.TIMER.summary(), None
    cand = {k: v[1] for k, v in pre_summ.items() if k != 'xr_rays_sampler'}      # (K1 runs once per refresh window, beside the refresh iteration: not a per-step kernel)
    dom_pick = max(cand, key=cand.get) if cand else 'xr_hashgrid_bwd'
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer(only={dom_pick}, train_only=True)
    if ops.LIVE_STATS is not None:
        ops.LIVE_STATS[1:3].zero_()            # running (live rows, valid rows) totals of the backward's row list
    rays0, samples0, it0 = tr.rays_done, tr.samples_done, tr.iter
    step_ev = [ops._CEvent() for _ in range(args.steps + 1)]     # one event per iteration boundary
    def exposure_timers():
        """what measures the compute stream's waits for the step's collectives: the per-iteration path's (grad_sync) and, when the native
        loop carries the exchange, the exchange's own (callbacks: in finish(); RCCL from native code: events recorded in xr_dist.hip)"""
        out = []
        if getattr(tr.net, 'grad_sync', None) is not None:
            out.append(tr.net.grad_sync.exposed)
        if tr._loop is not None and tr._loop.exchange is not None:
            out.append(tr._loop.exchange.exposed)
        return out
    for ex_t in exposure_timers():
        ex_t.summary()                                           # (drop what earlier iterations recorded)
        ex_t.on = True
    barrier()
    t0 = time.perf_counter()
    # (Trainer.run: the iterations between two grid refreshes are enqueued by one native call each, xr_ngp_loop_run -- the events in
    # front of every iteration and around the dominant entry point are recorded by that call)
    tr.run(args.steps, iter_events=step_ev)
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    it1 = tr.iter
    # operands the default MLP arithmetic had to saturate at fp16's range, over everything this process has run so far (pre-roll, warm-up,
    # timed window): waves counted by the forward kernels in the range word (xr_set_mlp_range_word); 0 = the fp16 x 2 split was exact
    mlp_range_events = ops.mlp_range_events(dev)
    step_ms = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(args.steps)]
    is_refresh = [(it0 + k) % sampler.update_grid_freq == 0 for k in range(args.steps)]
    ms_refresh = [m for m, r in zip(step_ms, is_refresh) if r]
    ms_normal = [m for m, r in zip(step_ms, is_refresh) if not r]

    rays = tr.rays_done - rays0
    samples = tr.samples_done - samples0
    stats = torch.tensor([elapsed, float(rays), float(samples)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
        elapsed_max, rays_all, samples_all = float(mx[0]), float(stats[1]), float(stats[2])
    else:
        elapsed_max, rays_all, samples_all = elapsed, float(rays), float(samples)

    if args.headline_only:
        if rank == 0:
            hl = {'value': rays_all / elapsed_max, 'ms_per_step': elapsed_max * 1e3 / args.steps, 'steps': args.steps}
            try:      # the dominant entry point's launches in this window (events around it inside the native loop), for the caller's roofline
                n_l, ms_l, _ = timer.summary()[dom_pick]
                hl.update(dominant=dom_pick, dominant_launches=n_l, dominant_avg_launch_us=ms_l * 1e3 / max(n_l, 1),
                          samples_per_step=samples / args.steps, table_params=int(tr.net.mlp.embedder_pos.params.numel()))
            except Exception:  # noqa: BLE001
                pass
            print(json.dumps(hl))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # The two backward entry points run on the rows whose dL/d(raw) is not exactly zero (ops.live_rows): the units ONE LAUNCH
    # PROCESSES are the live rows, and only those are priced -- the fraction is read from the device-side running totals.
    def live_fraction():
        if ops.LIVE_STATS is None:
            return 1.0
        live, valid = (int(v) & 0xffffffff for v in ops.LIVE_STATS[1:3].tolist())
        return live / valid if valid else 1.0

    LIVE_KERNELS = ('xr_nerf_mlp_bwd', 'xr_hashgrid_bwd')
    live_frac_timed = live_fraction()

    def roof_of(name, launches, total_ms, units, live_frac=1.0, native=True):
        bound, per_unit = ALGO[name]
        halves = world > 1 and name == 'xr_hashgrid_bwd'
        if halves:
            launches = max(1, launches // 2)      # data parallel: fine / coarse halves around the gradient collective = one step's scatter
        if name in LIVE_KERNELS:
            units = units * live_frac
        work = units * per_unit
        fused_adam = native and name == 'xr_hashgrid_bwd' and getattr(tr, 'fuse_adam', False) and world == 1
        if fused_adam:
            # the scatter applies the optimiser's update to the table itself (xr_hashgrid_bwd_adam): per launch and table
            # parameter it reads p, m, v, ema and writes them back (32 B; the gradient never exists in memory)
            work += launches * tr.net.mlp.embedder_pos.params.numel() * 32
        if bound == 'hbm':
            achieved, peak, unit = work / (total_ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
        else:
            achieved, peak, unit = work / (total_ms * 1e-3) / 1e12, MFMA_PEAK[ops.precision()], 'TFLOP/s'
        split = name == 'xr_nerf_mlp_fwd' and ops._mlp_mode() in (2, 3)
        if split:
            peak = MFMA_PEAK['f16']              # the kernel runs on the 16-bit matrix cores (fp16 and bf16: the same dense peak)
        out = {'kernel': name, 'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
               'avg_launch_us': total_ms * 1e3 / max(launches, 1), 'launches': launches,
               'algorithmic_per_sample': per_unit, 'algorithmic_bytes_or_flops_per_launch': work / max(launches, 1)}
        if split and ops._mlp_mode() == 3:
            out['arithmetic'] = ('xr_nerf_mlp_fwd(XR_MLP_F16X2): fp32 operands split into 2 fp16 parts (hi + lo, ~22 bits), 3 v_mfma_f32_32x32x16_f16 per '
                                 'product block, fp32 accumulate (4e-7 relative against float64, the fp32 MFMA path: 2e-7 -- '
                                 'profiles/r06_mlp_fwd_f16x2_split_probe.txt); priced on the ALGORITHMIC flops against the fp16 MFMA peak -- the '
                                 'matrix cores execute 3x these flops (issued_frac)')
            out['issued_frac'] = 3.0 * achieved / peak
        elif split:
            out['arithmetic'] = ('xr_nerf_mlp_fwd(XR_MLP_BF16X3): fp32 operands split exactly into 3 bf16 parts, 6 v_mfma_f32_32x32x16_bf16 per '
                                 'product block, fp32 accumulate (fp32-rounding accuracy); priced on the ALGORITHMIC flops against the '
                                 'bf16 MFMA peak -- the matrix cores execute 6x these flops (issued_frac)')
            out['issued_frac'] = 6.0 * achieved / peak
        if name == 'xr_nerf_mlp_bwd' and ops.precision() == 'f32' and bound == 'mfma':
            # mixed arithmetic (XR_MLP_BWD_DW, default h2f): the forward recompute (18432 of the 59392 flop / sample) with the forward's
            # own arithmetic (fp16 2-way split; b2x: on the fp32 MFMA), the dW products and the dX chain on the bf16 matrix cores with
            # 2-way split operands -- 3 matrix products per algorithmic one each.  The peak quoted is the rate at which the matrix
            # cores could finish exactly this mix.
            arith = os.environ.get('XR_MLP_BWD_DW', 'h2f')
            f_fwd, f_dx, f_dw = 18432.0, 20480.0, 20480.0
            on_f32 = {'f32': f_fwd + f_dx + f_dw, 'b2': f_fwd + f_dx, 'b2x': f_fwd, 'b2f': 0.0, 'h2f': 0.0}.get(arith, f_fwd)
            on_b16 = (f_fwd + f_dx + f_dw) - on_f32
            t_unit = on_f32 / (MFMA_F32_PEAK_TFLOPS * 1e12) + 3.0 * on_b16 / (MFMA_PEAK['f16'] * 1e12)
            mix_peak = (on_f32 + on_b16) / t_unit / 1e12
            if on_f32 == 0.0:
                # everything on the 16-bit matrix cores: `frac` = ALGORITHMIC flops against their dense peak, like the forward's; the mix rate
                # (833 TFLOP/s: three products per algorithmic one) is the issued fraction
                out['peak'], out['frac'], out['issued_frac'] = MFMA_PEAK['f16'], achieved / MFMA_PEAK['f16'], achieved / mix_peak
            else:
                out['peak'], out['frac'] = mix_peak, achieved / mix_peak
            out['arithmetic'] = ('XR_MLP_BWD_DW=%s: %.0f flop/sample on the fp32 MFMA (157.3 TFLOP/s), %.0f on the 16-bit matrix cores as 3 products '
                                 'each (2-way operand split, 2500 TFLOP/s dense); the rate of this mix: %.0f TFLOP/s' % (arith, on_f32, on_b16, mix_peak))
        if name == 'xr_hashgrid_fwd' and launches:
            # The lookup against the roofline that binds it.  It is not HBM: the table slices are L2-resident (one XCD per level) and every
            # (y, z) corner pair of a (sample, level) is one random 16-byte access = one 128-byte line pulled from the L2 into the CU's L1
            # (the two x-neighbours share the line; an odd x issues two 8-byte loads to it).  tools/gather_probe.hip: random 16-byte gathers
            # from L2-resident slices of 0.5 .. 4 MiB retire at 263 G accesses/s chip-wide (profiles/r02_gather_probe.txt; 970 G/s only
            # when the slice fits the 16-KiB L1), whatever their width -- the 64 B/clk/CU L2 -> L1 path moving whole lines.
            n_levels = tr.net.mlp.embedder_pos.meta.n_levels
            req = units / launches * n_levels * 4.0
            rate = req / (total_ms * 1e-3 / launches)
            out['requests_per_launch'] = req
            out['request_rate'] = rate
            out['request_ceiling'] = 263e9
            out['request_frac'] = rate / 263e9
            out['request_note'] = ('requests = samples x %d levels x 4 corner pairs (one 128-B line each from the XCD-local L2); ceiling = random L2-resident '
                                   'gathers, tools/gather_probe.hip (263 G/s).  At that ceiling the launch moves its 1164 algorithmic bytes per sample at %.2f of '
                                   'the HBM peak: the 0.60 of north_star is above what the L2 -> L1 path delivers for this access pattern '
                                   '(profiles/r06_lookup_cell_major_probe.txt: the one structural cut of the request count that was on the table, measured)'
                                   % (n_levels, 263e9 / (n_levels * 4.0) * per_unit / (HBM_PEAK_GBS * 1e9)))
        if halves:
            out['note'] = 'entry point called twice per step (levels 8..15, then 0..7, each handed to the all-reduce): figures are per step'
        if fused_adam:
            out['includes'] = 'Adam + L2 + EMA update of the %d table parameters, applied where each entry\'s gradient completes (32 B per parameter per launch on top of the per-sample figure)' % tr.net.mlp.embedder_pos.params.numel()
        if name in LIVE_KERNELS:
            out['live_row_fraction'] = live_frac
            out['units'] = 'samples with a non-zero output gradient (the rows the launch processes); the others are exact zeros'
        return out

    # ---- roofline of the dominant kernel: HIP events on the launch stream around every TRAINING launch of that entry
    # point inside the timed region (the occupancy-grid density queries use other entry points / are not counted)
    summ = timer.summary()
    launches, total_ms, units = summ[dom_pick]
    roof = roof_of(dom_pick, launches, total_ms, units if units > 0 else samples, live_frac_timed)
    roof['traffic'] = None
    try:   # HBM bytes per launch from separate rocprofv3 --pmc passes of THIS command (tools/pmc_traffic.py)
        # (hardware counters cannot be read from inside this process: the figure is the committed record of the last `rocprofv3 --pmc` passes
        # of this command -- `traffic_from_file` names it -- not a measurement of this run)
        pmc_path = next(p for p in (os.path.join(ROOT, 'profiles', n) for n in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json')) if os.path.exists(p))
        pmc = json.load(open(pmc_path))
        if dom_pick in pmc:
            roof['traffic'] = pmc[dom_pick]['bytes_fetch_x2']
            roof['traffic_from_file'] = 'profiles/' + os.path.basename(pmc_path)
            roof['traffic_unit'] = 'bytes/launch (PMC FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, 2^18-sample batches; read from %s, collected by tools/gpu_call.sh pmc)' % os.path.basename(pmc_path)
    except Exception:  # noqa: BLE001
        pass

    # ---- every hot kernel's roofline: 16 more iterations per entry point with events around THAT entry point only, so that every
    # window runs the product's native step (a timer that wants several stages of the step at once forces the per-entry-point
    # Python sequence, which has neither the fused optimiser update nor the small sums inside the scatter's launches).  Out of the timed region.
    roofs = {}
    s_win = 0
    for k in ALGO:
        if k in ('xr_rays_sampler', 'xr_calc_rgb_forward', 'xr_calc_rgb_backward'):
            continue          # K1 runs once per refresh window as a series (xr_ngp_window_march), not per iteration; K3 / K4 alone are not on the training path
        s0 = tr.samples_done
        ops.TIMER = ops.KernelTimer(only={k}, train_only=True)
        if ops.LIVE_STATS is not None:
            ops.LIVE_STATS[1:3].zero_()
        for _ in range(16):
            tr.step()
        torch.cuda.synchronize()
        timer2, ops.TIMER = ops.TIMER, None
        live_frac_win = live_fraction()
        s_win = tr.samples_done - s0
        summ2 = timer2.summary()
        if k in summ2:
            n_l, ms_l, u_l = summ2[k]
            roofs[k] = roof_of(k, n_l, ms_l, u_l if u_l > 0 else s_win, live_frac_win)      # Adam counts parameters, the rest samples
    r_win = None

    extra = {}
    if ops._mlp_mode() in (2, 3):
        # live evidence that the split forward is an fp32-accurate evaluation: the trained weights, the current batch's
        # marched samples, both forwards (never inside a timed region; a failure here must not cost the line)
        split_kind = ops.f32_forward()
        try:
            mlp = tr.net.mlp
            coords = sampler.coords
            n_chk = int(min(coords.shape[0], int(sampler.n_valid_dev[0]), 1 << 18))      # rows behind the count are stale
            enc_chk = ops.hashgrid_fwd(mlp.embedder_pos.params.detach(), coords[:n_chk, :3], mlp.embedder_pos.meta)
            outs = {}
            for kind in ('mfma', split_kind):
                ops.set_f32_forward(kind)
                outs[kind] = ops.nerf_mlp_fwd(enc_chk, coords[:n_chk, 4:], n_chk, mlp.density_net.params.detach(),
                                              mlp.color_net.params.detach(), 1, 2, mlp.pad_value).clone()
            dev_max = float((outs['mfma'] - outs[split_kind]).abs().max())
            if not (dev_max == dev_max and dev_max < float('inf')):
                raise ValueError('non-finite deviation')
            extra['mlp_forward_check'] = {
                'samples': n_chk, 'max_abs_raw_fp32_mfma': float(outs['mfma'].abs().max()),
                'max_abs_deviation_%s_vs_fp32_mfma' % split_kind: dev_max,
                'note': 'xr_nerf_mlp_fwd in the %s arithmetic (fp32 operands split into %s, fp32 accumulate) against XR_MLP_F32 (fp32 MFMA) on the trained '
                        'weights and the current training batch; parity bar on raw: 1e-4'
                        % (split_kind, '2 fp16 parts, 3 fp16 MFMAs per product block' if split_kind == 'f16x2' else '3 bf16 parts, 6 bf16 MFMAs per product block')}
        except Exception as e:  # noqa: BLE001
            extra['mlp_forward_check'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        finally:
            ops.set_f32_forward(split_kind)
    if not args.no_render:
        H = W = 800
        pose = tr.data.poses[0]
        row0, nrows = xdist.row_band(H, rank, world)
        for _ in range(2):
            rgb, alpha = render_frame(tr.net, pose, H, W, tr.data.focal, row0=row0, nrows=nrows)
            xdist.gather_image(torch.cat([rgb, alpha], -1), H, rank, world)
        barrier()
        t1 = time.perf_counter()
        n_frames = 5
        for f in range(n_frames):
            rgb, alpha = render_frame(tr.net, tr.data.poses[f % tr.data.n_img], H, W, tr.data.focal, row0=row0, nrows=nrows)
            img = xdist.gather_image(torch.cat([rgb, alpha], -1), H, rank, world)
        barrier()
        extra['render_ms_per_800x800_frame'] = (time.perf_counter() - t1) * 1e3 / n_frames
        extra['render_samples_per_ray'] = float(tr.net.sampler.coords.shape[0]) / max(1, nrows * W)
        # the same entry points in their RENDER launches (one launch = every marched sample of the frame / row band, ~14 M):
        # events on the launch stream around each launch of two more frames, outside the frame timer
        ops.TIMER = ops.KernelTimer(only={'xr_hashgrid_fwd', 'xr_nerf_mlp_fwd'}, train_only=False)
        for f in range(2):
            render_frame(tr.net, tr.data.poses[f % tr.data.n_img], H, W, tr.data.focal, row0=row0, nrows=nrows)
        torch.cuda.synchronize()
        timer_r, ops.TIMER = ops.TIMER, None
        extra['roofline_kernels_render'] = {k: roof_of(k, n_l, ms_l, u_l) for k, (n_l, ms_l, u_l) in timer_r.summary().items() if u_l > 0}
        # optional early-terminated rendering (pixels within 1e-4 of the full evaluation, tests/test_gpu_network.py)
        for _ in range(2):
            rgb, alpha = render_frame_ert(tr.net, pose, H, W, tr.data.focal, row0=row0, nrows=nrows)
            xdist.gather_image(torch.cat([rgb, alpha], -1), H, rank, world)
        barrier()
        t1 = time.perf_counter()
        for f in range(n_frames):
            rgb, alpha = render_frame_ert(tr.net, tr.data.poses[f % tr.data.n_img], H, W, tr.data.focal, row0=row0, nrows=nrows)
            img = xdist.gather_image(torch.cat([rgb, alpha], -1), H, rank, world)
        barrier()
        extra['render_ms_per_800x800_frame_early_termination_1e-4'] = (time.perf_counter() - t1) * 1e3 / n_frames
        ev, tot = render_frame_ert.last_evaluated
        extra['render_early_termination_evaluated_fraction'] = ev / max(tot, 1)
        if world == 1 and not args.no_extra:
            try:
                extra.update(registry_frame_ms(tr))
            except Exception as e:  # noqa: BLE001
                extra['render_ms_per_800x800_frame_registry_chunk4096'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    # north_star's named kernel as top-level scalars (the driver's record truncates nested objects)
    if 'xr_hashgrid_fwd' in roofs:
        extra['hash_lookup_frac_of_hbm_training_launches'] = roofs['xr_hashgrid_fwd']['frac']
    if 'xr_hashgrid_fwd' in extra.get('roofline_kernels_render', {}):
        extra['hash_lookup_frac_of_hbm_frame_launches'] = extra['roofline_kernels_render']['xr_hashgrid_fwd']['frac']
    if world > 1:
        sync = getattr(tr.net, 'grad_sync', None)
        # MEASURED: how long this rank's compute stream waited for the step's collectives inside the timed region (events around the
        # waits of grad_sync.finish(); what did not overlap), gathered from every rank; the model stays beside it
        parts = [t.summary() for t in exposure_timers()]
        n_exp = sum(p_['steps'] for p_ in parts)
        exposed = {'steps': n_exp, 'mean_ms': (sum((p_['mean_ms'] or 0.0) * p_['steps'] for p_ in parts) / n_exp) if n_exp else None,
                   'max_ms': max([p_['max_ms'] or 0.0 for p_ in parts] + [0.0]) if n_exp else None}
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, exposed)
        per_rank_iter = [None] * world
        torch.distributed.all_gather_object(per_rank_iter, (sum(ms_normal) / max(len(ms_normal), 1), sum(ms_refresh) / max(len(ms_refresh), 1)))
        extra['collective'] = {'mode': tr.dp_mode, 'exposed_collective_ms_per_rank': [e.get('mean_ms') for e in per_rank],
                               'exposed_collective_ms': max((e.get('mean_ms') or 0.0) for e in per_rank),
                               'exposed_collective_ms_max_step': max((e.get('max_ms') or 0.0) for e in per_rank),
                               'exposed_steps_measured': exposed.get('steps'),
                               'device_ms_normal_iteration_per_rank': [a for a, _ in per_rank_iter],
                               'device_ms_refresh_iteration_per_rank': [b for _, b in per_rank_iter],
                               'native_loop': bool(tr._loop is not None and tr._loop.enqueued > 0),
                               'exchange': type(tr._loop.exchange).__name__ if (tr._loop is not None and tr._loop.exchange is not None) else 'per-iteration path (torch.distributed)',
                               'exchange_fallback_reason': getattr(tr._loop.exchange, 'fallback_reason', None) if (tr._loop is not None and tr._loop.exchange is not None) else None,
                               'host_enqueue_ms_per_iteration_native_loop': (tr._loop.enqueue_s * 1e3 / tr._loop.enqueued) if (tr._loop is not None and tr._loop.enqueued) else None,
                               'model': xdist.comm_model(world, step_ms=elapsed_max * 1e3 / args.steps, wire_bytes_per_float=2.0 if tr.dp_mode == 'allreduce_bf16' else 4.0),
                               'bytes_on_wire_per_rank_total': getattr(sync, 'bytes_on_wire', None),
                               'bytes_reduced_per_rank_total': getattr(sync, 'bytes_reduced', None),
                               'bytes_gathered_per_rank_total': getattr(sync, 'bytes_gathered', None)}
        if rank == 0 and rccl_log is not None:
            try:    # what RCCL chose: the lines of its INFO log that name rings / trees, channel counts, algorithms, protocols, the transport
                keys = ('Ring', 'Tree', 'hannel', 'lgorithm', 'rotocol', 'via', 'xGMI', 'XGMI', 'P2P', 'comm 0x')
                lines = [l.strip()[-200:] for l in open(os.environ.get('NCCL_DEBUG_FILE', rccl_log)) if any(k in l for k in keys)]
                extra['collective']['rccl_log'] = {'lines_kept': len(lines), 'first': lines[:24], 'env': {k: v for k, v in os.environ.items() if k.startswith(('NCCL_', 'RCCL_'))}}
            except Exception as e:  # noqa: BLE001
                extra['collective']['rccl_log'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}

    if rank == 0:
        n_refresh = sum(1 for i in range(it0, it1) if i % sampler.update_grid_freq == 0)
        out = {
            'metric': 'training rays/s, Instant-NGP Lego (configs/instant_ngp/nerf_blender_local01.py), synthetic 800x800 rays',
            'value': rays_all / elapsed_max, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed_max * 1e3 / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'f32 tables, parameters, gradients, accumulation, compositor, optimiser state; the fused MLP multiplies on the fp16 '
                          'matrix cores with every f32 operand split into two fp16 parts (XRNERF_F32_FORWARD=%s: ~4e-7 relative on raw, not '
                          'f32-exact, operands above 65504 saturated and counted -- mlp_range_events); the reference (tiny-cuda-nn) '
                          'computes the MLP in plain fp16' % ops.f32_forward(),
            'library_build': __import__('xrnerf_amd.build', fromlist=['info']).info(),
            'parity': 'partial: sampling / compositing / grid upkeep pinned to the reference kernels (bit-exact indices and '
                      'counts, <=1e-4 fp32); hash grid + SH + fused MLP restate tiny-cuda-nn, which is absent from the '
                      'reference tree (parity unpinned)',
            'config': {'workload': 'Instant-NGP Lego, hash L=16 F=2 T=2^19, 64-wide fused MLP (1+2 hidden), 800x800, %d images; '
                                   'full training iterations %d..%d (batch slice + random bg, K1 march, every-16th grid refresh '
                                   'K6..K11 with its density queries: %d refreshes in the window = round(steps/16), encode, MLP, K3, 5*Huber, K4, '
                                   'MLP backward, table scatter, %sAdam + L2 + EMA over 12.2 M parameters%s) after %d un-timed '
                                   'pre-roll + %d warm-up iterations (adaptive batch at its fixed point: %s rays); the '
                                   'reference\'s dead no-grad MLP pass that only feeds K2\'s dead transmittance loop is skipped; fused-MLP '
                                   'forward: %s; fused-MLP backward: %s'
                                   % (args.n_img, it0, it1 - 1, n_refresh, 'gradient all-reduce, ' if world > 1 else '',
                                      ' (the table\'s update applied inside the scatter: no gradient round trip)' if (world == 1 and tr.fuse_adam) else '',
                                      preroll, args.warmup + align, hist[-1],
                                      'fp32 via exact 3-way bf16 operand split on the bf16 MFMA (xr_nerf_mlp_fwd, XR_MLP_BF16X3)' if ops._mlp_mode() == 2
                                      else 'fp32 results via 2-way fp16 operand split on the fp16 MFMA (xr_nerf_mlp_fwd, XR_MLP_F16X2; 4e-7 relative against float64)'
                                      if ops._mlp_mode() == 3 else 'fp32 MFMA',
                                      {'f32': 'fp32 MFMA throughout', 'b2': 'dW on 2-way-split bf16 operands (2^-16 per product), dX chain on the fp32 MFMA',
                                       'b2x': 'dW and the dX chain on 2-way-split bf16 operands (2^-16 relative per product, fp32 accumulate; '
                                              '1.0e-5 of max against a float64 statement away from ReLU kinks: profiles/r04_mlp_bwd_denc_outlier.txt), '
                                              'forward recompute in fp32 (3-way split)',
                                       'b2f': 'as b2x with the forward recompute on 2-way-split bf16 operands',
                                       'h2f': 'forward recompute with the forward\'s own arithmetic (2-way fp16 split), dW and the dX chain on 2-way-split '
                                              'bf16 operands (2^-16 relative per product), fp32 accumulate'}.get(os.environ.get('XR_MLP_BWD_DW', 'h2f'), 'see XR_MLP_BWD_DW')
                                      if ops._mlp_mode() != 1 else 'fp16 operands, fp32 accumulate'),
                       'rays_per_step': rays_all / args.steps / world, 'samples_per_ray': samples_all / max(rays_all, 1),
                       'samples_per_s': samples_all / elapsed_max, 'n_images': args.n_img,
                       'preroll_iterations': preroll, 'timed_iterations': [it0, it1 - 1], 'grid_refreshes_in_window': n_refresh,
                       'device_ms_normal_iteration': sum(ms_normal) / max(len(ms_normal), 1),
                       'device_ms_refresh_iteration': sum(ms_refresh) / max(len(ms_refresh), 1) if ms_refresh else None,
                       'rays_per_batch_history': hist,
                       'parallelism': 'ray-sharded data parallel x%d, gradient all-reduce (RCCL)' % world if world > 1 else 'single GPU'},
            'roofline': roof,
            'mlp_range_events': mlp_range_events,
            'roofline_kernels': roofs,
            'backward_live_row_fraction': {'timed_window': live_frac_timed, 'kernel_window': live_frac_win,
                                           'note': 'share of the marched samples whose dL/d(raw) is not exactly zero (T == 0 behind opaque '
                                                   'surfaces makes the rest exact zeros); the MLP backward and the table scatter process '
                                                   'these rows only, results are identical to the backward over every row'},
        }
        out.update(extra)
        # secondary measurements must never cost the headline line: a failure is reported in place of the numbers
        def guarded(key, fn):
            try:
                out[key] = fn()
            except Exception as e:  # noqa: BLE001
                out[key] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}

        if world == 1 and not args.no_cpu_baseline:
            guarded('cpu_baseline', cpu_baseline)
            guarded('cpu_baseline_vanilla_nerf_config1', cpu_vanilla_nerf)
        if world == 1 and not (args.no_mip and args.no_kilo and args.no_unbounded and args.no_f16 and args.no_extra and args.no_strict):
            del tr
            torch.cuda.empty_cache()
        if world == 1 and not args.no_f16:
            guarded('ngp_f16_mlp_mode', lambda: ngp_f16_mlp_mode(dev, args.n_img, max(args.steps, 32)))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_extra:
            guarded('value_all_rows', lambda: value_all_rows(args))
            guarded('ngp_real_lego_fixture', lambda: ngp_real_lego_fixture(dev))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_strict:
            guarded('ngp_tcnn_strict_defaults', lambda: ngp_tcnn_strict_defaults(dev, args.n_img, max(args.steps, 32)))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_unbounded:
            guarded('ngp_config4_unbounded', lambda: ngp_config4_unbounded(dev))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_mip:
            guarded('mipnerf_config3', lambda: mipnerf_config3(dev))
        if world == 1 and not args.no_kilo:
            torch.cuda.empty_cache()
            guarded('kilonerf_config5', lambda: kilonerf_config5(dev))
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
