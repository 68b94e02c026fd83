"""Import shim for the reference's pure-PyTorch modules (build container only): a ~40-line `mmcv` stub and
empty package shells whose __path__ points into /root/reference, so that the leaf files import unmodified
without dragging in every optional dependency of xrnerf/models/__init__.py."""
import importlib
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'xrnerf'))


def _stub_mmcv():
    if 'mmcv' in sys.modules:
        return

    class Registry:
        def __init__(self, name, parent=None, **kw):
            self.name, self._m = name, {}

        def register_module(self, name=None, force=False, module=None):
            def reg(cls):
                self._m[name or cls.__name__] = cls
                return cls
            return reg(module) if module is not None else reg

        def build(self, cfg, **kw):
            cfg = dict(cfg)
            return self._m[cfg.pop('type')](**cfg)

    mmcv = types.ModuleType('mmcv')
    cnn = types.ModuleType('mmcv.cnn'); cnn.MODELS = Registry('model')
    utils = types.ModuleType('mmcv.utils'); utils.Registry = Registry
    runner = types.ModuleType('mmcv.runner'); runner.get_dist_info = lambda: (0, 1)
    mmcv.cnn, mmcv.utils, mmcv.runner = cnn, utils, runner
    sys.modules.update({'mmcv': mmcv, 'mmcv.cnn': cnn, 'mmcv.utils': utils, 'mmcv.runner': runner})


def load():
    """-> namespace with the reference's BaseEmbedder, NerfMLP, NerfRender, sample_pdf"""
    assert available()
    _stub_mmcv()
    for pkg in ('xrnerf', 'xrnerf.models', 'xrnerf.models.embedders', 'xrnerf.models.mlps', 'xrnerf.models.renders',
                'xrnerf.models.networks', 'xrnerf.models.networks.utils'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, *pkg.split('.'))]
            sys.modules[pkg] = m
    sys.modules['xrnerf.models'].builder = importlib.import_module('xrnerf.models.builder')
    ns = types.SimpleNamespace()
    ns.builder = sys.modules['xrnerf.models'].builder
    ns.BaseEmbedder = importlib.import_module('xrnerf.models.embedders.base').BaseEmbedder
    ns.NerfMLP = importlib.import_module('xrnerf.models.mlps.nerf_mlp').NerfMLP
    ns.NerfRender = importlib.import_module('xrnerf.models.renders.nerf_render').NerfRender
    ns.sample_pdf = importlib.import_module('xrnerf.models.networks.utils.hierarchical_sample').sample_pdf
    return ns


def load_mip():
    """-> namespace with the reference's Mip-NeRF pieces (BASELINE config #3): mip.py's functions,
    MipNerfEmbedder, MipNerfRender, NerfMLP, GetZvals and load_rays_multiscale, imported unmodified.
    Extra stubs: `turtle` (mipnerf_embedder.py:2 imports it by accident; needs tkinter), `cv2` / `imageio`
    (module-level imports of datasets/pipelines/create.py, unused by GetZvals), mmcv.parallel."""
    ns = load()
    for name in ('turtle', 'cv2', 'imageio'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.forward = None
            sys.modules[name] = m
    mmcv = sys.modules['mmcv']
    if not hasattr(mmcv, 'parallel'):
        par = types.ModuleType('mmcv.parallel'); par.collate = None
        mmcv.parallel = par
        sys.modules['mmcv.parallel'] = par
        mmcv.utils.build_from_cfg = lambda cfg, reg, default_args=None: reg.build(cfg)
        mmcv.utils.digit_version = lambda v: tuple(int(x) for x in v.split('+')[0].split('.')[:3])
    for pkg in ('xrnerf.datasets', 'xrnerf.datasets.pipelines', 'xrnerf.datasets.load_data'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, *pkg.split('.'))]
            sys.modules[pkg] = m
    ns.mip = importlib.import_module('xrnerf.models.networks.utils.mip')
    ns.MipNerfEmbedder = importlib.import_module('xrnerf.models.embedders.mipnerf_embedder').MipNerfEmbedder
    ns.MipNerfRender = importlib.import_module('xrnerf.models.renders.mipnerf_render').MipNerfRender
    sys.modules['xrnerf.datasets'].builder = importlib.import_module('xrnerf.datasets.builder')
    ns.GetZvals = importlib.import_module('xrnerf.datasets.pipelines.create').GetZvals
    ns.load_rays_multiscale = importlib.import_module('xrnerf.datasets.load_data.get_rays').load_rays_multiscale
    # networks/mipnerf.py does `from .utils import (merge_ret, mse2psnr, ...)`; utils/__init__.py would pull in every
    # model family's dependencies, so the names are placed on the package shell from their leaf modules instead
    utils = sys.modules['xrnerf.models.networks.utils']
    for leaf, names in (('transforms', ('merge_ret', 'recover_shape')), ('metrics', ('mse2psnr', 'img2mse', 'HuberLoss')),
                        ('batching', ('unfold_batching',)), ('mip', ('resample_along_rays', 'sample_along_rays')),
                        ('hierarchical_sample', ('sample_pdf',))):
        mod = importlib.import_module('xrnerf.models.networks.utils.' + leaf)
        for n in names:
            setattr(utils, n, getattr(mod, n))
    utils.__all__ = [n for n in vars(utils) if not n.startswith('_')]
    sys.modules['mmcv.runner'].load_checkpoint = None
    if 'tqdm' not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = types.ModuleType('tqdm'); t.tqdm = lambda x, **k: x
            sys.modules['tqdm'] = t
    ns.MipNerfNetwork = importlib.import_module('xrnerf.models.networks.mipnerf').MipNerfNetwork
    return ns


def load_kilo():
    """-> namespace with the reference's in-tree KiloNeRF pieces (BASELINE config #5): reorder_points_and_dirs,
    convert_to_local_coords_multi, MultiNetworkFourierEmbedding, MultiNetwork, NerfRender.  `kilonerf_cuda` (external,
    absent) is only imported inside try/except by these files."""
    ns = load()
    tr = importlib.import_module('xrnerf.models.networks.utils.transforms')
    ns.reorder_points_and_dirs = tr.reorder_points_and_dirs
    ns.convert_to_local_coords_multi = tr.convert_to_local_coords_multi
    ns.multi_modules = importlib.import_module('xrnerf.models.mlps.multi_modules')
    ns.MultiNetwork = ns.multi_modules.MultiNetwork
    emb = importlib.import_module('xrnerf.models.embedders.kilonerf_fourier_embedder')
    ns.MultiNetworkFourierEmbedding = emb.MultiNetworkFourierEmbedding
    ns.KiloNerfFourierEmbedder = emb.KiloNerfFourierEmbedder
    return ns


def load_ngp(raymarch_module, tcnn_module):
    """-> namespace with the reference's Instant-NGP Python, imported UNMODIFIED from /root/reference with the two
    extension modules it binds replaced by this repository's drop-ins:
        sys.modules['raymarch_cuda'] = raymarch_module   (xrnerf_amd.raymarch_cuda)
        sys.modules['tinycudann']    = tcnn_module       (xrnerf_amd.tcnn, or a test double with the same surface)
    Loaded: samplers/utils/*.py (the ten autograd-Function wrappers), samplers/ngp_grid_sampler.py (NGPGridSampler),
    mlps/hashnerf_mlp.py (HashNerfMLP), renders/hashnerf_render.py (HashNerfRender), networks/hashnerf.py
    (HashNerfNetwork) and the loss helpers they use.  cfg dicts must be `Cfg` (attribute access like mmcv's ConfigDict)."""
    assert available()
    _stub_mmcv()
    sys.modules['raymarch_cuda'] = raymarch_module
    sys.modules['tinycudann'] = tcnn_module
    ns = load()
    for pkg in ('xrnerf.models.samplers',):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, *pkg.split('.'))]
            sys.modules[pkg] = m
    # a module imported earlier against other extension modules would keep them: (re)import the binding files
    for name in [n for n in sys.modules if n.startswith('xrnerf.models.samplers.') or n in (
            'xrnerf.models.renders.hashnerf_render', 'xrnerf.models.mlps.hashnerf_mlp', 'xrnerf.models.networks.hashnerf',
            'xrnerf.models.networks.nerf')]:
        del sys.modules[name]
    ns.sampler_utils = importlib.import_module('xrnerf.models.samplers.utils')
    ns.NGPGridSampler = importlib.import_module('xrnerf.models.samplers.ngp_grid_sampler').NGPGridSampler
    ns.HashNerfMLP = importlib.import_module('xrnerf.models.mlps.hashnerf_mlp').HashNerfMLP
    ns.HashNerfRender = importlib.import_module('xrnerf.models.renders.hashnerf_render').HashNerfRender
    utils = sys.modules['xrnerf.models.networks.utils']
    for leaf, names in (('transforms', ('merge_ret', 'recover_shape')), ('metrics', ('mse2psnr', 'img2mse', 'HuberLoss')),
                        ('batching', ('unfold_batching',)), ('hierarchical_sample', ('sample_pdf',))):
        mod = importlib.import_module('xrnerf.models.networks.utils.' + leaf)
        for n in names:
            setattr(utils, n, getattr(mod, n))
    utils.__all__ = [n for n in vars(utils) if not n.startswith('_')]
    if 'tqdm' not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = types.ModuleType('tqdm'); t.tqdm = lambda x, **k: x
            sys.modules['tqdm'] = t
    ns.HashNerfNetwork = importlib.import_module('xrnerf.models.networks.hashnerf').HashNerfNetwork
    return ns


class Cfg(dict):
    """attribute-style dict (what mmcv's ConfigDict gives the reference's constructors: `cfg.chunk`, `'chunk' in cfg`)"""
    __getattr__ = dict.__getitem__
