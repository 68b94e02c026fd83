"""Registry-level modules of the widened rows on the host: MipNerfNetwork and KiloNerfNetwork (forward, autograd nodes,
train_step) driven exactly like their GPU tests (tests/test_gpu_mip.py, tests/test_gpu_kilo.py), with the kernels running
through the HIP-on-CPU shim (tests/hip_emu) -- against the reference fixtures.  The 8x256 MLP of the Mip-NeRF path runs as its one
autograd node over the emulated linear kernels (vanilla._NerfMlpFn: strided operands, no concatenations); the narrow layers of the
vanilla config take torch's own linear here (host tensors)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hip_emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def edev():
    import emulib
    ctx = emulib.emulated_ops()
    dev = ctx.__enter__()
    yield dev
    ctx.__exit__(None, None, None)


def test_mipnerf_network_against_the_reference_fixture_on_the_host(edev):
    import test_gpu_mip as T
    gold = np.load(os.path.join(G, 'ref_mipnerf.npz'))
    T.test_network_against_reference_fixture(edev, gold)
    T.test_render_fixture(edev, gold, '', dict(density_bias=-1., rgb_padding=0.001, white_bkgd=True, density_activation='softplus'))


def test_whole_mlp_as_one_autograd_node_on_the_host(edev):
    """vanilla._NerfMlpFn (strided linear kernels, the skip and view buffers written in place) against the layer-by-layer float64 graph"""
    import test_gpu_linear as T
    T.test_whole_mlp_as_one_autograd_node_equals_the_layer_by_layer_graph(edev, 130, True)
    T.test_whole_mlp_as_one_autograd_node_equals_the_layer_by_layer_graph(edev, 33, False)


def test_kilonerf_network_and_gradients_on_the_host(edev, tmp_path):
    import kilo_oracle as K
    import test_gpu_kilo as T
    gold = np.load(os.path.join(G, 'ref_kilonerf.npz'))
    T.test_network_behind_the_registry(edev, gold, tmp_path)          # render through the registry + one fine-tuning step
    T.test_parameter_gradients_on_the_reference_fixture(edev, K, gold)
