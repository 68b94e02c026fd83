"""Host side of the Mip-NeRF path without a GPU: registry contract on the reference's config #3 model dict, channel
arithmetic, C-ABI exports, and loud failure (no CPU fallback) when handed host tensors."""
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_reference_config3_builds_unchanged():
    import xrnerf_amd
    from xrnerf_amd import mip, vanilla
    cfg = json.load(open(os.path.join(G, 'mip_model_cfg.json')))
    p = '/root/reference/configs/mipnerf/mipnerf_multiscale.py'
    if os.path.exists(p):
        import runpy
        assert json.loads(json.dumps(runpy.run_path(p)['model'])) == cfg['model']
    net = xrnerf_amd.build_network(cfg['model'])
    assert isinstance(net, mip.MipNerfNetwork) and isinstance(net.mlp, vanilla.NerfMLP)
    assert isinstance(net.mlp.embedder, mip.MipNerfEmbedder) and isinstance(net.render, mip.MipNerfRender)
    assert net.mlp.embedder.get_embed_ch() == (96, 27)
    assert (net.num_levels, net.ray_shape, net.resample_padding, net.coarse_loss_mult) == (2, 'cone', 0.01, 0.1)
    assert net.mlp.pts_linears[0].in_features == 96 and net.mlp.pts_linears[5].in_features == 256 + 96
    assert net.mlp.views_linears[0].in_features == 256 + 27
    # parameter names are the reference's (its checkpoints load strictly): see tests/test_gpu_mip.py for the values
    keys = set(net.state_dict().keys())
    assert {'mlp.pts_linears.0.weight', 'mlp.views_linears.0.bias', 'mlp.feature_linear.weight', 'mlp.alpha_linear.weight',
            'mlp.rgb_linear.bias'} <= keys


def test_exports_and_channels():
    from xrnerf_amd import _lib, ops
    L = _lib.load()
    for name in ('xr_mip_zvals', 'xr_mip_encode', 'xr_mip_encode_gaussians', 'xr_mip_render_forward',
                 'xr_mip_render_backward', 'xr_mip_resample', 'xr_mip_encode_channels'):
        assert hasattr(L, name)
    assert ops.mip_encode_channels(0, 16, 0, 4, True) == 123
    assert ops.mip_encode_channels(2, 7, 1, 3, False) == 42
    hdr = open(os.path.join(os.path.dirname(G), '..', 'include', 'xrnerf_mi355.h')).read()
    for name in _lib.SIGNATURES:
        assert name + '(' in hdr, name          # every bound symbol is declared in the public header


def test_no_cpu_fallback():
    from xrnerf_amd import _lib, mip
    data = {'near': torch.full((4, 1), 2.0), 'far': torch.full((4, 1), 6.0)}
    with pytest.raises(_lib.XrError):
        mip.get_z_vals(data, 9)
    emb = mip.MipNerfEmbedder(0, 16, 0, 4, use_viewdirs=True)
    z = torch.linspace(2, 6, 9).expand(4, 9)
    d = {'z_vals': z, 'rays_o': torch.zeros(4, 3), 'rays_d': torch.ones(4, 3), 'radii': torch.ones(4, 1) * 1e-3,
         'viewdirs': torch.ones(4, 3)}
    with pytest.raises(_lib.XrError):
        emb(mip.sample_along_rays(d, 'cone'))


def test_lazy_gaussians_match_the_reference_fixture():
    """MipSamples unpacks to the reference's (means, covs) for callers that want the gaussians themselves"""
    import numpy as np
    from xrnerf_amd import mip
    g = np.load(os.path.join(G, 'ref_mipnerf.npz'))
    t = lambda k: torch.tensor(g[k])
    for shape in ('cone', 'cylinder'):
        means, covs = mip.MipSamples(t('z_vals'), t('ray_rays_o'), t('ray_rays_d'), t('ray_radii'), shape)
        assert np.abs(means.numpy() - g['means_' + shape]).max() <= 2e-6
        assert np.abs(covs.numpy() - g['covs_' + shape]).max() <= 1e-7 + 1e-5 * np.abs(g['covs_' + shape]).max()
