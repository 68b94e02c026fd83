"""Generates tests/golden/ref_mipnerf.npz from the REFERENCE'S OWN Mip-NeRF code (BASELINE config #3),
in the build container only:  python tests/golden/make_golden_mip.py

Every array is an input or an output of an unmodified reference function imported from /root/reference
through tests/golden/ref_import.py::load_mip():
  load_rays_multiscale (datasets/load_data/get_rays.py:100-152)  -> rays_o/rays_d/viewdirs/radii of a 2-scale camera
  GetZvals            (datasets/pipelines/create.py:486-531)     -> z_vals (randomized, draws stored) / z_det
  cast_rays           (networks/utils/mip.py:134-148)            -> means, covs   (cone and cylinder)
  MipNerfEmbedder.forward (embedders/mipnerf_embedder.py:85-99)  -> embedded
  MipNerfRender.forward   (renders/nerf_render.py:45-98 + mipnerf_render.py) -> rgb, disp, acc, weights, d(sum(G*rgb))/d raw
  resample_along_rays (networks/utils/mip.py:151-176)            -> new z_vals (randomized with stored draws, and not)
Random draws the reference takes from torch's global RNG are reproduced by re-seeding and stored in the fixture,
so the tests do not depend on torch's RNG stream.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

S = 32            # intervals per ray in the fixture (config: 128)


def multiscale_rays(ns, rng):
    """two cameras (full and half resolution) on the Blender hemisphere, like the multiscale dataset's meta"""
    def pose(theta, phi, radius=4.0):
        c, s = np.cos, np.sin
        cam = np.array([radius * c(phi) * c(theta), radius * c(phi) * s(theta), radius * s(phi)])
        fwd = -cam / np.linalg.norm(cam)
        right = np.cross(fwd, [0, 0, 1.]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, cam
        return m.astype(np.float32)

    widths, heights = [16, 8], [12, 6]
    focal = [18.0, 9.0]
    pix2cam = [np.array([[1. / f, 0, -.5 * w / f], [0, -1. / f, .5 * h / f], [0, 0, -1.]], np.float32)
               for f, w, h in zip(focal, widths, heights)]
    meta = dict(pix2cam=pix2cam, cam2world=[pose(0.3, 0.5), pose(2.1, 0.9)], width=widths, height=heights,
                lossmult=[1.0, 4.0], near=[2.0, 2.0], far=[6.0, 6.0])
    rays = ns.load_rays_multiscale(meta, 2)
    flat = {k: np.concatenate([np.asarray(v[i], np.float32).reshape(-1, v[i].shape[-1]) for i in range(2)], 0)
            for k, v in rays.items()}
    pick = rng.choice(flat['rays_o'].shape[0], 40, replace=False)
    return {k: v[pick] for k, v in flat.items()}


def main():
    assert ref_import.available(), 'needs /root/reference (run in the build container)'
    ns = ref_import.load_mip()
    rng = np.random.default_rng(31)
    out = {}
    rays = multiscale_rays(ns, rng)
    R = rays['rays_o'].shape[0]
    for k, v in rays.items():
        out['ray_' + k] = v
    T = {k: torch.tensor(v) for k, v in rays.items()}

    # GetZvals, randomized: reproduce its torch.rand draw
    torch.manual_seed(5)
    z_rand = torch.rand([R, S + 1])
    torch.manual_seed(5)
    res = ns.GetZvals(lindisp=False, N_samples=S + 1, randomized=True)(dict(T))
    out['z_rand'], out['z_vals'] = z_rand.numpy(), res['z_vals'].numpy()
    out['z_det'] = ns.GetZvals(lindisp=False, N_samples=S + 1, randomized=False)(dict(T))['z_vals'].numpy()
    out['z_lindisp'] = ns.GetZvals(lindisp=True, N_samples=S + 1, randomized=False)(dict(T))['z_vals'].numpy()
    z = res['z_vals']

    for shape in ('cone', 'cylinder'):
        means, covs = ns.mip.cast_rays(z, T['rays_o'], T['rays_d'], T['radii'], shape)
        out['means_' + shape], out['covs_' + shape] = means.numpy(), covs.numpy()

    emb = ns.MipNerfEmbedder(min_deg_point=0, max_deg_point=16, min_deg_view=0, max_deg_view=4, use_viewdirs=True,
                             append_identity=True)
    data = dict(T); data['z_vals'] = z
    data = ns.mip.sample_along_rays(data, 'cone')
    out['embedded'] = emb(data)['embedded'].numpy()
    emb2 = ns.MipNerfEmbedder(min_deg_point=2, max_deg_point=7, min_deg_view=1, max_deg_view=3, use_viewdirs=True,
                              append_identity=False)
    data2 = dict(T); data2['z_vals'] = z
    data2 = ns.mip.sample_along_rays(data2, 'cylinder')
    out['embedded_cyl_2_7_1_3_noid'] = emb2(data2)['embedded'].numpy()

    # render + its gradient
    raw = torch.tensor(rng.normal(0, 2.5, (R, S, 4)).astype(np.float32))
    raw[3, :, 3] = -40.0                       # an empty ray (acc -> 0: the nan_to_num / clamp branch of get_disp_map)
    raw[4, :, 3] = 30.0                        # softplus' linear branch, saturating ray
    G = torch.tensor(rng.normal(0, 1, (R, 3)).astype(np.float32))
    out['raw'], out['grad_rgb'] = raw.numpy(), G.numpy()
    for tag, kw in (('', dict(white_bkgd=True, density_bias=-1., rgb_padding=0.001, density_activation='softplus')),
                    ('_relu_black', dict(white_bkgd=False, density_bias=0., rgb_padding=0., density_activation='relu'))):
        render = ns.MipNerfRender(raw_noise_std=0, **kw)
        r = raw.clone().requires_grad_(True)
        d = {'raw': r, 'z_vals': z, 'rays_d': T['rays_d']}
        d, ret = render(d, False)
        (ret['rgb'] * G).sum().backward()
        for k in ('rgb', 'disp', 'acc'):
            out['render%s_%s' % (tag, k)] = ret[k].detach().numpy()
        out['render%s_weights' % tag] = d['weights'].detach().numpy()
        out['render%s_grad_raw' % tag] = r.grad.numpy()

    # resample_along_rays from the white/softplus weights
    w = torch.tensor(out['render_weights'])
    torch.manual_seed(9)
    u_rand = torch.rand([R, S + 1])
    torch.manual_seed(9)
    d = dict(T); d['z_vals'] = z; d['weights'] = w.clone()
    d = ns.mip.resample_along_rays(d, True, 'cone', 0.01)
    out['resample_rand'], out['resample_z_rand'] = u_rand.numpy(), d['z_vals'].numpy()
    out['resample_means'], out['resample_covs'] = d['samples'][0].numpy(), d['samples'][1].numpy()
    d = dict(T); d['z_vals'] = z; d['weights'] = w.clone()
    out['resample_z_det'] = ns.mip.resample_along_rays(d, False, 'cone', 0.01)['z_vals'].numpy()
    # zero padding with all-zero / tiny weights: the eps-padding branch of sorted_piecewise_constant_pdf
    wz = w.clone(); wz[0] = 0.0; wz[1] = 1e-9; wz[2, : S // 2] = 0.0
    out['resample_w_zero'] = wz.numpy()
    d = dict(T); d['z_vals'] = z; d['weights'] = wz.clone()
    out['resample_z_det_pad0'] = ns.mip.resample_along_rays(d, False, 'cone', 0.0)['z_vals'].numpy()

    # ---- network level: the reference's MipNerfNetwork (config's model dict, MLP shrunk to 4 x 64 for the fixture)
    import copy, json, runpy
    cfg = runpy.run_path('/root/reference/configs/mipnerf/mipnerf_multiscale.py')
    json.dump({'model': cfg['model'], 'num_samples': cfg['num_samples'], 'N_rand_per_sampler': cfg['N_rand_per_sampler'],
               'optimizer': cfg['optimizer'], 'lr_config': cfg['lr_config']},
              open(os.path.join(HERE, 'mip_model_cfg.json'), 'w'), indent=1, sort_keys=True)
    model = copy.deepcopy(cfg['model'])
    model['mlp'].update(netdepth=4, netwidth=64, skips=[2])

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    torch.manual_seed(21)
    net = ns.MipNerfNetwork(Cfg(model['cfg']), mlp=model['mlp'], render=model['render'])
    for k, v in net.state_dict().items():
        out['net_sd.' + k] = v.numpy().copy()
    target = torch.tensor(rng.uniform(0, 1, (R, 3)).astype(np.float32))
    out['net_target'] = target.numpy()
    base = dict(T); base['z_vals'] = torch.tensor(out['z_det'])
    with torch.no_grad():
        ret = net.forward({k: v.clone() for k, v in base.items()}, is_test=True)
    for k in ('rgb', 'coarse_rgb', 'disp', 'coarse_disp', 'acc', 'coarse_acc'):
        out['net_test_' + k] = ret[k].numpy()
    # one training step: the only torch.rand call inside is the resampler's [R, S+1] draw
    torch.manual_seed(33)
    out['net_train_rand'] = torch.rand([R, S + 1]).numpy()
    torch.manual_seed(33)
    data = {k: v.clone()[None] for k, v in T.items()}       # the DataLoader's batch dimension of 1 (unfold_batching)
    data['z_vals'] = z.clone()[None]; data['target_s'] = target[None]
    res = net.train_step(data, None)
    res['loss'].backward()
    for k in ('loss', 'loss_fine', 'loss_coarse', 'psnr'):
        out['net_train_' + k] = np.float32(res['log_vars'][k])
    for name in ('mlp.rgb_linear.weight', 'mlp.pts_linears.0.weight', 'mlp.alpha_linear.bias'):
        out['net_grad.' + name] = dict(net.named_parameters())[name].grad.numpy().copy()

    np.savez_compressed(os.path.join(HERE, 'ref_mipnerf.npz'), **out)
    print('ref_mipnerf.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
