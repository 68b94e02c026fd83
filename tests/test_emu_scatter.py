"""The third-generation table scatter (xrnerf_amd/csrc/xr_scatter.hip: binned hashed + dense levels, run-length kernel for
the small dense levels, overflow lists, overwrite mode) executed from its real source on the host (tests/hip_emu) against
oracle/ngp_oracle.c, at sizes the emulation finishes in seconds: the row threshold of the path is lowered with XR_SC_MIN_N.
Each case is its own process because the switches are read once per process (tests/scatter_emu_case.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, 'tests', 'scatter_emu_case.py')


def run(n, mode, **env):
    e = dict(os.environ, XR_SC_MIN_N='256', **env)
    r = subprocess.run([sys.executable, CASE, str(n), mode], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('n,mode', [(3000, 'rand'), (9000, 'rays'), (5000, 'cluster'), (3000, 'faces')])
def test_scatter_generation_3_against_the_oracle(n, mode):
    out = run(n, mode)
    assert out.count('levels out of tolerance: []') == 7, out
    assert 'identical to scatter + optimiser launch: True' in out, out


@pytest.mark.parametrize('n,mode,env', [(5000, 'cluster', dict(XR_SC_BLOCK='1024')), (5000, 'rays', dict(XR_SC_BLOCK='4096')),
                                        (3000, 'faces', dict(XR_SC_RL='0')), (5000, 'rays', dict(XR_SC_RL_CHUNKS='3')),
                                        (5000, 'cluster', dict(XR_SC_MODE='1')), (5000, 'rays', dict(XR_SC_DENSE_ATOMIC='1'))])
def test_scatter_switches_give_the_same_gradients(n, mode, env):
    run(n, mode, **env)

