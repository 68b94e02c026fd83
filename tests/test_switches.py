"""The host side's runtime switches (xrnerf_amd/switches.py): parsing, defaults, loud failures -- and that tools/README.md's table names
exactly the switches the sources read (an environment variable that is read but not listed, or listed but gone, fails here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_and_frame_modes(monkeypatch):
    from xrnerf_amd import switches
    monkeypatch.delenv('XRNERF_STEP', raising=False)
    monkeypatch.delenv('XRNERF_FRAME', raising=False)
    assert switches.step_mode() == 'fused' and switches.frame_mode() == 'one_launch'
    for m in ('fused', 'py', 'modular'):
        monkeypatch.setenv('XRNERF_STEP', m)
        assert switches.step_mode() == m
    for m in ('one_launch', 'async', 'sync', 'ert'):
        monkeypatch.setenv('XRNERF_FRAME', m)
        assert switches.frame_mode() == m
    monkeypatch.setenv('XRNERF_STEP', '1')
    with pytest.raises(ValueError):
        switches.step_mode()
    monkeypatch.setenv('XRNERF_FRAME', 'loop')
    with pytest.raises(ValueError):
        switches.frame_mode()


def test_trainer_overrides(monkeypatch):
    from xrnerf_amd import switches
    monkeypatch.delenv('XRNERF_TRAINER', raising=False)
    assert switches.trainer_overrides() == {}
    monkeypatch.setenv('XRNERF_TRAINER', 'fuse_adam=0,native_loop=1,march_window=main')
    assert switches.trainer_overrides() == {'fuse_adam': False, 'native_loop': True, 'march_window': 'main'}
    monkeypatch.setenv('XRNERF_TRAINER', 'fused_adam=0')
    with pytest.raises(ValueError):
        switches.trainer_overrides()


def test_readme_lists_exactly_the_switches_the_sources_read():
    listed = set(re.findall(r'^\| `(X[A-Z0-9_]+)` \|', open(os.path.join(ROOT, 'tools', 'README.md')).read(), flags=re.M))
    read = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'xrnerf_amd')):
        if os.sep + 'build' in dirpath:
            continue
        for f in files:
            if not f.endswith(('.py', '.hip', '.h')):
                continue
            src = open(os.path.join(dirpath, f), errors='replace').read()
            read |= set(re.findall(r'getenv\("(X[A-Z0-9_]+)"\)', src))
            if f.endswith('.py'):
                read |= set(re.findall(r'''os\.environ(?:\.get)?[\[(]['"](X[A-Z0-9_]+)['"]''', src))
    read -= {'XR_EXTRA_HIPCC_FLAGS'}                      # build time (xrnerf_amd/build.py), listed under the table
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    assert len(listed) <= 13
