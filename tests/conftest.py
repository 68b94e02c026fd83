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


def grad_close(got, ref, what='', kinks=None):
    """Gradient parity against the oracle.  The bar is max|got - ref| <= 1e-3 * max(1, max|ref|).

    The default backward recomputes the hidden activations with the forward's arithmetic (fp16 2-way operand split,
    ~4e-7 relative -- see profiles/r05_mlp_fwd_f16x2_split_probe.txt) while the oracle recomputes them in float64: a hidden
    unit whose pre-activation lies within that distance of zero can land on the other side of the ReLU, and then ONE sample's
    contribution to a weight entry appears or disappears.  That is a property of ReLU under any two summation orders, not of the
    kernel (tests/test_gpu_tcnn.py::test_nerf_mlp_bwd_split_recompute_differs_by_relu_kinks_only shows these are the only
    differences), so where the recompute is the split one (`kinks`, default: XR_MLP_BWD_DW unset or 'h2f') a SMALL number of
    entries may exceed the bar, bounded three ways: at most 5 % of the entries, none beyond 3e-2 * max (one sample's whole contribution to
    a table entry few samples touch), and the whole difference
    within 1e-2 of the reference in the 2-norm (ONE sample of a 4 100-sample batch that changes sides in a 10-layer network moved the table
    gradient by 5.3e-3 of its norm).  With the fp32 recompute (f32 / b2 / b2x) the plain bar applies."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    if kinks is None:
        kinks = os.environ.get('XR_MLP_BWD_DW', 'h2f') == 'h2f'
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    err = np.abs(got - ref)
    worst = float(err.max()) if err.size else 0.0
    if worst <= 1e-3 * scale:
        return
    assert kinks, (what, worst, scale)
    over = float((err > 1e-3 * scale).mean())
    rel2 = float(np.linalg.norm(got - ref)) / max(float(np.linalg.norm(ref)), 1e-30)
    assert over <= 0.05 and worst <= 3e-2 * scale and rel2 <= 1e-2, (what, worst, scale, over, rel2)
