"""TEST INFRASTRUCTURE: drives the REFERENCE's own Instant-NGP Python (samplers/utils/*.py, ngp_grid_sampler.py,
hashnerf_mlp.py, hashnerf_render.py, networks/hashnerf.py, imported unmodified through tests/golden/ref_import.py with
`raymarch_cuda` / `tinycudann` replaced by xrnerf_amd.raymarch_cuda / xrnerf_amd.tcnn) side by side with this package's
classes, from identical state and rays, and records / compares the state trajectory.

Used by tests/test_reference_python_on_shims.py (CPU: the kernels run through tests/hip_emu) and by
tests/golden/make_golden_ngp_trajectory.py (fixture replayed on the GPU by tests/test_gpu_trajectory.py)."""
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

from xrnerf_amd import synthetic as S  # noqa: E402

H = W = 800
N_IMG = 6


def scene(n_img=N_IMG, aabb_scale=1):
    """what PassDatasetHook hands to the sampler (hashnerf_dataset.py:55-86), for synthetic Lego cameras; aabb_scale > 1:
    the unbounded setting of BASELINE config #4 (log2(aabb_scale) + 1 occupancy cascades)"""
    poses = S.lego_cameras(n_img, seed=1)
    alldata = {'aabb_scale': aabb_scale, 'aabb_range': (0.5 - aabb_scale / 2, 0.5 + aabb_scale / 2), 'poses': poses,
               'focal': np.ones((n_img, 2), dtype=float) * float(S.LEGO_FOCAL), 'metadata': S.metadata_rows(n_img, float(S.LEGO_FOCAL))}
    return poses, alldata, {'H': H, 'W': W}


def batch(poses, n_rays, it, dev):
    """deterministic training batch of iteration `it`: rays, random targets / alpha / background"""
    o, d, ids = S.training_rays(poses, n_rays, seed=1000 + it)
    rng = np.random.default_rng(5000 + it)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return {'rays_o': t(o), 'rays_d': t(d), 'img_ids': t(ids.astype(np.int32)),
            'target_s': t(rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)),
            'alpha': t((rng.uniform(0, 1, (n_rays, 1)) > 0.3).astype(np.float32)),
            'bg_color': t(rng.uniform(0, 1, (n_rays, 3)).astype(np.float32))}


class OracleMlp(nn.Module):
    """Stand-in for HashNerfMLP with the two entry points a sampler uses -- run_density(pts) and __call__(data)['raw'] --
    computed by the plain-C oracle (OpenMP) so that a 2 M-point occupancy-grid query takes seconds on the host.  The SAME
    object is handed to the reference's sampler and to ours: what is compared is the two samplers."""

    def __init__(self, seed=0):
        super().__init__()
        import oracle as O
        self.O, self.meta = O, O.GridMeta()
        self.table = S.hash_table(self.meta.n_params, seed=3 + seed, scale=0.5)        # large values: a structured density field
        self.wd = S.mlp_weights(32, 64, 1, 16, 4 + seed)
        self.wc = S.mlp_weights(32, 64, 2, 16, 5 + seed)
        self.O.set_threads(max(1, min(8, os.cpu_count() or 1)))

    def run_density(self, pts):
        x = pts.detach().cpu().numpy().astype(np.float32).reshape(-1, 3)
        raw = self.O.nerf_mlp_fwd(self.table, self.wd, self.wc, x, None, self.meta)
        return torch.from_numpy(raw[:, 3:4].copy()).to(pts.device)

    def forward(self, data):
        x = data['pts'].detach().cpu().numpy().astype(np.float32).reshape(-1, 3)
        d = data['viewdirs'].detach().cpu().numpy().astype(np.float32).reshape(-1, 3)
        raw = self.O.nerf_mlp_fwd(self.table, self.wd, self.wc, x, d, self.meta) if x.shape[0] else np.zeros((0, 4), np.float32)
        data['raw'] = torch.from_numpy(raw).to(data['pts'].device)
        return data


def sampler_state(s, n_valid=None):
    """everything the render / the hooks / the next iteration read from a sampler after sample()"""
    cnt = s.rays_numsteps_compacted.cpu().numpy()
    total = int(cnt[:, 0].sum())
    out = {'n_rays_per_batch': int(s.n_rays_per_batch), 'numsteps': s.rays_numsteps.cpu().numpy().copy(), 'numsteps_compacted': cnt.copy(),
           'coords': s.coords[:total].cpu().numpy().copy(), 'bitfield': s.density_grid_bitfield.cpu().numpy().copy(),
           'grid_mean': s.density_grid_mean.cpu().numpy().copy(), 'grid': s.density_grid.cpu().numpy().copy(),
           'ema_step': int(s.density_grid_ema_step)}
    return out


def compare_states(a, b, it, exact=True):
    """reference-side state `a` vs this package's `b` after iteration `it`"""
    assert a['n_rays_per_batch'] == b['n_rays_per_batch'], (it, a['n_rays_per_batch'], b['n_rays_per_batch'])
    assert a['ema_step'] == b['ema_step'], it
    assert np.array_equal(a['bitfield'], b['bitfield']), 'bitfield differs after iteration %d' % it
    assert np.array_equal(a['grid'], b['grid']), 'density grid differs after iteration %d' % it
    assert np.array_equal(a['grid_mean'][:1], b['grid_mean'][:1]), it
    # per-ray sample counts: the reference assigns sample ranges with atomicAdd (any order); (count) must agree ray by
    # ray, (count, base) as a multiset -- here the bases agree too because both run the same kernels
    assert np.array_equal(a['numsteps'][:, 0], b['numsteps'][:, 0]), 'per-ray sample counts differ at iteration %d' % it
    assert np.array_equal(a['numsteps_compacted'], b['numsteps_compacted']), it
    assert a['coords'].shape == b['coords'].shape and np.array_equal(a['coords'], b['coords']), 'sample rows differ at iteration %d' % it
