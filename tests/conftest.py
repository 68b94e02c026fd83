import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def O():
    """the CPU oracle (test infrastructure only)"""
    import oracle
    return oracle


@pytest.fixture(scope='session')
def lego():
    """synthetic Lego-shaped scene: density grid, its bitfield (oracle K11), cameras"""
    import oracle
    from xrnerf_amd import synthetic as S
    grid = S.lego_density_grid()
    mean = oracle.density_mean(grid)
    bf = oracle.bitfield_given_mean(grid, mean)
    poses = S.lego_cameras(20)
    return dict(grid=grid, mean=mean, bitfield=bf, poses=poses)


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from xrnerf_amd import _lib
    _lib.load()   # raises (does not skip) when the HIP library is missing on a GPU box
    return torch.device('cuda:0')


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
