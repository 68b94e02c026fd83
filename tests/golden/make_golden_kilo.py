"""Generates tests/golden/ref_kilonerf.npz from the REFERENCE'S OWN in-tree KiloNeRF code (BASELINE config #5), in the
build container only:  python tests/golden/make_golden_kilo.py

The reference's fast path needs the external `kilonerf_cuda` library; its in-tree PyTorch statements of the same
operations (used by its distillation phase) are what runs here, unmodified, through tests/golden/ref_import.py:
  reorder_points_and_dirs       (networks/utils/transforms.py:57-151)  -> active samples, grouping, counts
  convert_to_local_coords_multi (transforms.py:35-45)                  -> local coordinates
  MultiNetworkFourierEmbedding  (embedders/kilonerf_fourier_embedder.py:12-57, 'pytorch')
  MultiNetwork / extract_single_network (mlps/multi_modules.py:405-707, 'bmm')   -> per-network tiny MLP
  KiloNerfMLP.forward's scatter-back (mlps/kilonerf_mlp.py:176-189), re-stated line by line below
  NerfRender.forward            (renders/nerf_render.py:45-98)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

FIXED_RES = [3, 4, 2]
RES = [48, 64, 32]
GMIN = np.float32([-1.2, -1.5, -0.8])
GMAX = np.float32([1.3, 1.4, 0.9])


def scene(rng, R=48, S=40):
    """rays that cross the global domain, partly from outside; uniform samples"""
    cam = rng.normal(0, 1, (R, 3)); cam = 3.2 * cam / np.linalg.norm(cam, axis=-1, keepdims=True)
    tgt = rng.uniform(-0.9, 0.9, (R, 3))
    d = tgt - cam; d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, (R, 1))
    z = np.linspace(1.2, 5.2, S)[None, :] + rng.uniform(0, 0.1, (R, S)).cumsum(-1) * 0.2
    occ = np.zeros(RES, bool)
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, r) for r in RES], indexing='ij'), -1)
    occ[(g ** 2).sum(-1) < 0.55] = True
    occ &= rng.uniform(0, 1, RES) < 0.9
    return cam.astype(np.float32), d.astype(np.float32), np.sort(z, -1).astype(np.float32), occ


def main():
    assert ref_import.available()
    ns = ref_import.load_kilo()
    rng = np.random.default_rng(51)
    out = {}
    o, d, z, occ = scene(rng)
    R, S = z.shape
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    N = int(np.prod(FIXED_RES))
    out.update(rays_o=o, rays_d=d, viewdirs=vd, z_vals=z, occupancy=occ.reshape(-1), gmin=GMIN, gmax=GMAX,
               fixed_res=np.int64(FIXED_RES), res=np.int64(RES))
    # nodes exactly as get_nodes_fixed_resolution builds them (datasets/kilonerf_node_dataset.py:108-135)
    import itertools
    voxel = (GMAX.astype(np.float64) - GMIN.astype(np.float64)) / np.array(FIXED_RES)
    dmins, dmaxs = [], []
    for vi in itertools.product(*[range(r) for r in FIXED_RES]):
        dmins.append((GMIN.astype(np.float64) + np.array(vi) * voxel).tolist())
        dmaxs.append((GMIN.astype(np.float64) + (np.array(vi) + 1) * voxel).tolist())
    dmins, dmaxs = torch.tensor(dmins), torch.tensor(dmaxs)          # float32, like KiloNerfMLP.init_mlp
    out['domain_mins'], out['domain_maxs'] = dmins.numpy(), dmaxs.numpy()

    torch.manual_seed(7)
    multi = ns.MultiNetwork(N, 63, 27, 4, 32, 2, None, True, 32, 'relu', linear_implementation='bmm')
    with torch.no_grad():                      # distilled networks have O(1) activations; default init is too flat
        for p in multi.parameters():
            p.mul_(2.5)
    names = ['pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear']
    mods = dict(multi.named_modules())
    for nm in names:                            # stored in the fast path's layout: weight [N, in, out]
        out['w.' + nm] = mods[nm].weight.detach().permute(0, 2, 1).contiguous().numpy()
        out['b.' + nm] = mods[nm].bias.detach().numpy().copy()

    T = torch.tensor
    pts = T(o)[:, None, :] + T(d)[:, None, :] * T(z)[:, :, None]           # GetPts, create.py:588-597
    out['pts'] = pts.numpy()
    data = {'pts': pts, 'viewdirs': T(vd), 'global_domain_min': T(GMIN), 'global_domain_max': T(GMAX)}
    data = ns.reorder_points_and_dirs(data, FIXED_RES, RES, T(occ.reshape(-1)), N)
    out['active_samples'] = data['active_samples_mask'].numpy()
    out['batch_size_per_network'] = data['batch_size_per_network'].numpy()
    # network of every active sample (the reference's sort is unstable: compare as sets per network)
    pr, dr = data['points_reordered'], data['directions_reordered']
    emb_p = ns.MultiNetworkFourierEmbedding(1, 3, 10)
    emb_d = ns.MultiNetworkFourierEmbedding(1, 3, 4)
    raw_sorted = torch.empty(pr.shape[0], 4)
    start = 0
    with torch.no_grad():
        for n in range(N):
            c = int(data['batch_size_per_network'][n])
            if c == 0:
                continue
            local = ns.convert_to_local_coords_multi(pr[start:start + c][None], dmins[n:n + 1], dmaxs[n:n + 1])
            e = torch.cat([emb_p(local), emb_d(dr[start:start + c][None].contiguous())], 2)
            raw_sorted[start:start + c] = multi.extract_single_network(n)(e)[0]
            if n == int(np.argmax(out['batch_size_per_network'])):
                out['probe_net'], out['probe_local'], out['probe_embedded'] = np.int64(n), local[0].numpy(), e[0].numpy()
                out['probe_raw'] = raw_sorted[start:start + c].numpy().copy()
            start += c
    # kilonerf_mlp.py:176-189
    back = torch.empty_like(raw_sorted)
    back[data['reorder_indices']] = raw_sorted
    full = torch.zeros(R * S, 4)
    full[data['active_samples_mask']] = back
    raw = full.view(R, S, 4)
    out['raw'] = raw.numpy()
    render = ns.NerfRender(white_bkgd=True, raw_noise_std=0)
    dd, ret = render({'raw': raw, 'z_vals': T(z), 'rays_d': T(d)}, True)
    out['rgb'], out['disp'], out['acc'] = ret['rgb'].numpy(), ret['disp'].numpy(), ret['acc'].numpy()
    out['weights'] = dd['weights'].numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_kilonerf.npz'), **out)
    print('active %d of %d samples, %d networks used' % (out['active_samples'].size, R * S, int((out['batch_size_per_network'] > 0).sum())))
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
