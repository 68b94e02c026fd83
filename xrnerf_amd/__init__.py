"""xrnerf_amd: MI355X-native Instant-NGP hot path behind openxrlab/xrnerf's module registry.

Importing the package registers the reference's type names (`HashNerfNetwork`, `NGPGridSampler`,
`HashNerfMLP`, `HashNerfRender`) in `xrnerf_amd.builder.MODELS`, so
`build_network(cfg.model)` works on the reference's configs/instant_ngp/*.py unchanged; likewise the
vanilla-NeRF (config #1, `vanilla.py`), Mip-NeRF (config #3, `mip.py`) and KiloNeRF (config #5, `kilo.py`) type names.
"""
from . import builder  # noqa: F401
from .builder import build_embedder, build_mlp, build_network, build_render, build_sampler  # noqa: F401
from . import mlps, networks, renders, samplers, vanilla, mip, kilo  # noqa: F401,E402

__version__ = '0.1.0'
