"""Module registry with the reference's type-name contract.

Mirrors /root/reference/xrnerf/models/builder.py:7-36 (one mmcv `Registry('models')` aliased five
ways; classes self-register with `@X.register_module()`, configs pick them with `type=`).  mmcv is
not a dependency here: this is the small part of its Registry that contract needs.
"""
import inspect


class ConfigDict(dict):
    """dict with attribute access (mmcv.utils.ConfigDict behaviour that NerfNetwork relies on:
    `cfg.get('phase')`, `'chunk' in cfg`, `cfg.chunk`; networks/nerf.py:23-31)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(ConfigDict.wrap(v) for v in obj)
        return obj


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError('%s is already registered in %s' % (key, self.name))
            self._modules[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args=None):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise KeyError('cfg must be a dict with a "type" key, got %r' % (cfg,))
        args = dict(cfg)
        t = args.pop('type')
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError('%s is not in the %s registry' % (t, self.name))
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        if not inspect.isclass(cls):
            raise TypeError('type must be a str or class')
        return cls(**{k: ConfigDict.wrap(v) for k, v in args.items()})


MODELS = Registry('models')
MLPS = MODELS
RENDERS = MODELS
EMBEDDERS = MODELS
NETWORKS = MODELS
SAMPLERS = MODELS


def build_mlp(cfg):
    return MLPS.build(cfg)


def build_render(cfg):
    return RENDERS.build(cfg)


def build_embedder(cfg):
    return EMBEDDERS.build(cfg)


def build_network(cfg):
    return NETWORKS.build(cfg)


def build_sampler(cfg):
    return SAMPLERS.build(cfg)
