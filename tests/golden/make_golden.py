"""Generates tests/golden/*.npz|json from the REFERENCE ITSELF, in the build container only
(/root/reference is absent on the GPU box, so the fixtures -- not the reference -- travel):

  ref_raymarch.npz  : inputs + outputs of the reference's own K1..K11 kernels
                      (/root/reference/extensions/ngp_raymarch/src/*.cu compiled for the CPU by
                      oracle/build.py through oracle/shim)
  ref_python.npz    : outputs of the reference's pure-Python pieces on the path, imported from
                      /root/reference with an mmcv stub: get_rays_np_hash, poses_nerf2ngp, HuberLoss,
                      img2mse/mse2psnr, get_per_level_scale
  ngp_model_cfg.json: the `model` / optimizer / lr dicts of configs/instant_ngp/nerf_blender_local01.py

Run:  python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import runpy
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
REF = '/root/reference'


def load_ref_module(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_inputs():
    """small deterministic inputs shared by the generator and the tests (no RNG-stream dependence:
    everything that is random is STORED in the fixture)."""
    from xrnerf_amd import synthetic as S
    grid = S.sphere_density_grid(0.3)                      # analytic, deterministic
    return grid


def main():
    import oracle as O
    from xrnerf_amd import synthetic as S
    assert O.have_ref(), 'needs /root/reference (run in the build container)'
    rng = np.random.default_rng(2026)
    out = {}
    grid = golden_inputs()
    # K10/K11 through the reference's update_bitfield_api
    mean16k, bf = O.update_bitfield_ref(grid * np.float32(0.02))
    out['k11_mean'] = mean16k[:1]
    bits_per_level = np.array([int(np.unpackbits(bf[l * S.G3 // 8:(l + 1) * S.G3 // 8]).sum()) for l in range(8)])
    out['k11_bits_per_level'] = bits_per_level
    out['k11_crc'] = np.array([np.bitwise_xor.reduce(bf.view(np.uint32) * (np.arange(bf.size // 4, dtype=np.uint32) | 1))],
                              dtype=np.uint32)
    # K1 on 96 rays (incl. edge cases), two consecutive launches (hidden RNG advances by 2^32)
    poses = S.lego_cameras(6, seed=9)
    o, d, ids = S.training_rays(poses, 84, seed=4)
    eo = np.array([[0.5, 0.5, -1.0], [0.5, 0.5, -1.0], [-1.0, 0.5, 0.5], [0.5, 2.0, 0.5], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5],
                   [3.0, 3.0, 3.0], [0.0, 0.5, -1.0], [1.0, 1.0, -1.0], [0.45, 0.55, -2.0], [0.5, 0.5, -1.0],
                   [0.31, 0.4, 0.35]], np.float32)
    ed = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.577, 0.577, 0.577],
                   [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [0.01, -0.01, 0.9999],
                   [1e-9, 1e-9, 1.0], [-0.6, 0.0, 0.8]], np.float32)
    o, d = np.concatenate([eo, o]), np.concatenate([ed, d])
    out['k1_rays_o'], out['k1_rays_d'] = o, d
    bf_sphere = O.bitfield_given_mean(grid, np.float32(0.5), backend='ref')
    for call in (0, 1):
        c, ri, ns, cnt = O.rays_sampler(o, d, bf_sphere, rng_calls=call, backend='ref')
        s = int(cnt[1])
        out['k1_call%d_numsteps' % call] = ns
        out['k1_call%d_index' % call] = ri
        out['k1_call%d_counter' % call] = cnt
        out['k1_call%d_coords' % call] = c[:s]
    c, ns, s = out['k1_call0_coords'], out['k1_call0_numsteps'], int(out['k1_call0_counter'][1])
    # K2 with clipping
    cap = s // 2
    co, nc, rc, sc = O.compacted_coord(c, ns, cap, backend='ref')
    out['k2_cap'] = np.array([cap]); out['k2_numsteps'] = nc; out['k2_counters'] = np.array([rc[0], sc[0]])
    out['k2_coords'] = co[:cap]
    # K3/K4/K5
    raw = rng.normal(0, 1.5, (s, 4)).astype(np.float32)
    raw[:, 3] = rng.normal(2.0, 3.0, s)
    bg = rng.uniform(0, 1, (ns.shape[0], 3)).astype(np.float32)
    grad = rng.normal(0, 1, (ns.shape[0], 3)).astype(np.float32)
    out['k3_raw'], out['k3_bg'], out['k4_grad'] = raw, bg, grad
    rgb = O.calc_rgb_forward(raw, c, ns, nc, bg, 2, 3, backend='ref')
    out['k3_rgb'] = rgb
    out['k4_draw_mean_small'] = O.calc_rgb_backward(raw, nc, c, grad, rgb, 0.001, 2, 3, backend='ref')
    out['k4_draw_mean_large'] = O.calc_rgb_backward(raw, nc, c, grad, rgb, 0.5, 2, 3, backend='ref')
    r5, a5 = O.calc_rgb_inference(raw, c, ns, [0.2, 0.5, 0.9], 2, 3, backend='ref')
    out['k5_rgb'], out['k5_alpha'] = r5, a5
    # K6 (two launches with different hidden RNG state), on the sphere grid scaled into (0, 0.02]
    g6 = (grid * np.float32(0.02)).astype(np.float32)
    for call, (n, step, casc, thr) in enumerate([(4096, 0, 0, -0.01), (4096, 5, 0, 0.01)]):
        p, i = O.generate_grid_samples(g6, step, n, casc, thr, rng_calls=call, backend='ref')
        out['k6_call%d_pos' % call], out['k6_call%d_idx' % call] = p, i
    # K7: count of untrained cells in the first two cascades + a strided sample of the result
    focal = np.full((6, 2), S.LEGO_FOCAL, np.float32)
    m = O.mark_untrained(focal, poses, 2 * S.G3, (800, 800), backend='ref')
    out['k7_poses'] = poses
    out['k7_count_neg'] = np.array([(m < 0).sum()])
    out['k7_sample'] = m[::4099].copy()
    # K8/K9
    idx = rng.integers(0, S.G3, 5000).astype(np.int32); idx[:200] = idx[0]
    mlp = rng.normal(0, 2, (5000, 1)).astype(np.float32)
    tmp = O.splat(mlp, idx, np.zeros(8 * S.G3, np.float32), backend='ref')
    out['k8_idx'], out['k8_mlp'] = idx, mlp
    out['k8_vals'] = tmp[idx]
    g9 = g6.copy(); g9[::7] = -1.0
    e = O.ema(tmp, g9, backend='ref')
    out['k9_vals'] = e[idx]; out['k9_sum'] = np.array([e.astype(np.float64).sum()])
    np.savez_compressed(os.path.join(HERE, 'ref_raymarch.npz'), **out)
    print('ref_raymarch.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')

    # ---------------- reference Python pieces
    py = {}
    gr = load_ref_module('xrnerf/datasets/load_data/get_rays.py', 'ref_get_rays')
    hu = load_ref_module('xrnerf/datasets/utils/hashnerf.py', 'ref_hashnerf_utils')
    import torch
    K = np.array([[S.LEGO_FOCAL, 0, 400.0], [0, S.LEGO_FOCAL, 400.0], [0, 0, 1]])
    p44 = np.stack([S.blender_pose(30.0, -30.0, 4.0), S.blender_pose(-110.0, -55.0, 4.0)])
    ngp = hu.poses_nerf2ngp(p44.copy(), [1, -1, -1], 0.33, [0.5, 0.5, 0.5])
    py['poses44'] = p44; py['poses_ngp'] = ngp
    ro, rd = gr.get_rays_np_hash(800, 800, K, ngp[0])
    sel = rng.integers(0, 800, (64, 2))
    py['rays_sel'] = sel
    py['rays_o'] = np.asarray(ro)[sel[:, 0], sel[:, 1]].astype(np.float64)
    py['rays_d'] = np.asarray(rd)[sel[:, 0], sel[:, 1]].astype(np.float64)
    me = load_ref_module('xrnerf/models/networks/utils/metrics.py', 'ref_metrics')
    x = rng.uniform(0, 1, (200, 3)).astype(np.float32); y = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    py['loss_x'], py['loss_y'] = x, y
    py['huber_sum'] = np.array([float(me.HuberLoss(torch.tensor(x), torch.tensor(y), 0.1, 'sum'))])
    py['mse'] = np.array([float(me.img2mse(torch.tensor(x), torch.tensor(y)))])
    py['psnr'] = np.array([float(me.mse2psnr(me.img2mse(torch.tensor(x), torch.tensor(y))))])
    # get_per_level_scale lives in a module that imports tinycudann at import time (guarded by try/except)
    sys.modules.setdefault('xrnerf', types.ModuleType('xrnerf'))
    src = open(os.path.join(REF, 'xrnerf/models/mlps/hashnerf_mlp.py')).read()
    ns_ = {}
    exec(src[src.index('def get_per_level_scale'):src.index('@MLPS.register_module()')], {'np': np}, ns_)
    py['per_level_scale'] = np.array([ns_['get_per_level_scale'](1)])
    np.savez_compressed(os.path.join(HERE, 'ref_python.npz'), **py)
    print('ref_python.npz ok')

    # ---------------- vanilla NeRF (BASELINE config #1) through the reference's OWN modules, small widths
    sys.path.insert(0, HERE)
    import ref_import
    R = ref_import.load()
    torch.manual_seed(0)
    mcfg = dict(skips=[2], netdepth=4, netwidth=32, output_ch=5, use_viewdirs=True, netchunk=1024 * 32,
                embedder=dict(type='BaseEmbedder', i_embed=0, multires=10, multires_dirs=4))
    ref_mlp, ref_fine = R.NerfMLP(**mcfg), R.NerfMLP(**mcfg)
    ref_render = R.NerfRender(white_bkgd=True, raw_noise_std=0)
    van = {}
    n_rays, n_s = 48, 16
    g = torch.Generator().manual_seed(1)
    rays_o = torch.randn(n_rays, 3, generator=g) * 0.1 + torch.tensor([0., 0., 4.])
    rays_d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g) * 0.2 - torch.tensor([0., 0., 1.]), dim=-1) * 1.3
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    t = torch.linspace(0., 1., n_s)
    z = (2. * (1 - t) + 6. * t).expand(n_rays, n_s)
    t_rand = torch.rand(n_rays, n_s, generator=g)
    mids = .5 * (z[..., 1:] + z[..., :-1])
    z = torch.cat([z[..., :1], mids], -1) + (torch.cat([mids, z[..., -1:]], -1) - torch.cat([z[..., :1], mids], -1)) * t_rand
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    data = {'pts': pts, 'viewdirs': viewdirs, 'z_vals': z, 'rays_o': rays_o, 'rays_d': rays_d}
    with torch.no_grad():
        data = ref_mlp(data)
        van['coarse_raw'] = data['raw'].numpy().copy()
        data, ret = ref_render(data, False)
        van['coarse_rgb'], van['coarse_disp'], van['coarse_acc'] = ret['rgb'].numpy(), ret['disp'].numpy(), ret['acc'].numpy()
        van['coarse_weights'] = data['weights'].numpy().copy()
        data = R.sample_pdf(data, 24, False, True)                  # deterministic u
        van['fine_z'] = data['z_vals'].numpy().copy()
        data = ref_fine(data)
        _, fret = ref_render(data, True)
        van['fine_rgb'] = fret['rgb'].numpy()
    for k, v in ref_mlp.state_dict().items():
        van['sd_coarse.' + k] = v.numpy()
    for k, v in ref_fine.state_dict().items():
        van['sd_fine.' + k] = v.numpy()
    van.update(rays_o=rays_o.numpy(), rays_d=rays_d.numpy(), viewdirs=viewdirs.numpy(), z_vals=z.numpy(), t_rand=t_rand.numpy())
    np.savez_compressed(os.path.join(HERE, 'ref_vanilla_nerf.npz'), **van)
    print('ref_vanilla_nerf.npz ok', sum(v.nbytes for v in van.values()) // 1024, 'KiB raw')

    cfg = runpy.run_path(os.path.join(REF, 'configs/instant_ngp/nerf_blender_local01.py'))
    keep = {k: cfg[k] for k in ('model', 'optimizer', 'lr_config', 'custom_hooks', 'max_iters', 'N_rand_per_sampler')}
    keep['vanilla_model'] = runpy.run_path(os.path.join(REF, 'configs/nerf/nerf_blender_base01.py'))['model']
    json.dump(keep, open(os.path.join(HERE, 'ngp_model_cfg.json'), 'w'), indent=1, sort_keys=True)
    print('ngp_model_cfg.json ok')

    # the pybind surface of the extension module: function -> parameter names (pybind_api.h:4-95)
    import re
    src = re.sub(r'//.*', '', open(os.path.join(REF, 'extensions/ngp_raymarch/include/pybind_api.h')).read())
    api = {}
    for m in re.finditer(r'void\s+(\w+_api)\s*\((.*?)\)\s*;', src, flags=re.S):
        params = [q.strip() for q in m.group(2).split(',') if q.strip()]
        api[m.group(1)] = [re.split(r'[\s&\*]+', q)[-1] for q in params]
    json.dump(api, open(os.path.join(HERE, 'raymarch_cuda_api.json'), 'w'), indent=1, sort_keys=True)
    print('raymarch_cuda_api.json ok')


if __name__ == '__main__':
    main()
