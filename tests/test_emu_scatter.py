"""The third-generation table scatter (xrnerf_amd/csrc/xr_scatter.hip: binned hashed + dense levels, run-length kernel for
the small dense levels, overflow lists, overwrite mode) executed from its real source on the host (tests/hip_emu) against
oracle/ngp_oracle.c, at sizes the emulation finishes in seconds: the row threshold of the path is lowered with XR_SC_TEST=min_n=256.
Each case is its own process because the switches are read once per process (tests/scatter_emu_case.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, 'tests', 'scatter_emu_case.py')


def run(n, mode, test=''):
    e = dict(os.environ, XR_SC_TEST='min_n=256' + (',' + test if test else ''))
    r = subprocess.run([sys.executable, CASE, str(n), mode], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('n,mode', [(3000, 'rand'), (9000, 'rays'), (5000, 'cluster'), (3000, 'faces')])
def test_scatter_generation_3_against_the_oracle(n, mode):
    out = run(n, mode)
    assert out.count('levels out of tolerance: []') == 9, out
    assert 'identical to scatter + optimiser launch: True' in out, out


@pytest.mark.parametrize('n,mode,test', [(5000, 'cluster', 'block=1024'), (5000, 'rays', 'block=4096'), (3000, 'faces', 'rl=0'),
                                         (5000, 'rays', 'rl_chunks=3')])
def test_scatter_layout_parameters_give_the_same_gradients(n, mode, test):
    """XR_SC_TEST: samples per binning workgroup, the small dense levels through the bins, row chunks of the run-length kernel"""
    run(n, mode, test)

