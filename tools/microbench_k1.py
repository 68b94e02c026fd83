"""K1 (occupancy-grid ray march) on 12.5 K training rays of the synthetic Lego grid: total time of one call"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import oracle as O
from xrnerf_amd import ops, synthetic as S
dev = torch.device('cuda:0')
grid = S.lego_density_grid(); bf = O.bitfield_given_mean(grid, O.density_mean(grid))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12544
o, d, _ = S.training_rays(S.lego_cameras(20), n, seed=3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
to, td, tb = t(o), t(d), t(bf)
def run(): return ops.rays_sampler(to, td, tb, (0., 1.), 0.05, 1 / 256, n * 64, 0)
c, ri, ns, cnt = run(); torch.cuda.synchronize()
for _ in range(3): run()
torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
nsn = ns[:, 0].cpu().numpy()
print('rays %d samples %d  per ray mean %.1f max %d  rays > 64: %d  > 128: %d   K1 %.1f us' % (
    n, int(cnt[1]), nsn.mean(), nsn.max(), (nsn > 64).sum(), (nsn > 128).sum(), a.elapsed_time(b) / 20 * 1e3))
