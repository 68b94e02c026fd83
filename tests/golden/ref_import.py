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
