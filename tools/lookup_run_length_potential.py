"""How many table-corner loads would a run-length lookup save?  CPU only (the oracle marches the rays): per level, the number of RUNS --
maximal stretches of consecutive samples of a lane's R-row chunk that fall into one grid cell -- per sample, on the synthetic Lego
scene's training rays.  A thread that walks R consecutive rows and re-loads its 8 corners only on a cell change issues `runs` loads
instead of `samples`; weighted with the measured per-level cost of the lookup (profiles/r03_microbench_fwd3.txt) this bounds what such
a kernel could gain -- IF its loads were compacted across the lanes of a wave (an issued vector-memory instruction costs its
texture-address cycles whatever its exec mask: with 64 lanes on 64 different rays some lane changes its cell at nearly every step).
usage: python tools/lookup_run_length_potential.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, ROOT)
import oracle as O
from xrnerf_amd import synthetic as S

COST = [23.7, 22.8, 23.6, 23.5, 23.8, 27.6, 32.0, 38.9, 47.3, 50.5, 51.4, 51.6, 51.6, 51.7, 52.0, 52.0]
grid = S.lego_density_grid(); bf = O.bitfield_given_mean(grid, O.density_mean(grid))
o, d, _ = S.training_rays(S.lego_cameras(20), 6000, seed=3)
rc, ri, rn, rcnt = O.rays_sampler(o, d, bf)
n = int(rcnt[1]); x = rc[:n, :3]
hit = int((rn[:, 0] > 0).sum())
print('%d samples on %d rays that meet the occupied cells (%.1f per ray)' % (n, hit, n / max(hit, 1)))
om = O.GridMeta()
for R in (4, 8, 16):
    runs = []
    for l in range(16):
        g = np.floor(x * om.scale[l] + 0.5).astype(np.int64)
        key = (g[:, 0] * 4099 + g[:, 1]) * 4099 + g[:, 2]
        new = np.ones(n, bool); new[1:] = key[1:] != key[:-1]
        new[::R] = True                       # a lane's chunk of R rows starts a run
        runs.append(float(new.mean()))
    saved = sum(c * (1 - r) for c, r in zip(COST, runs))
    print('R = %2d  runs per sample, levels 0..15: %s' % (R, ' '.join('%.2f' % v for v in runs)))
    print('         upper bound of the gain (every level\'s whole cost proportional to its loads, compaction and staging free): '
          '%.0f of %.0f cost units = %.1f %%' % (saved, sum(COST), 100 * saved / sum(COST)))
