"""per-level cost of the hash-grid scatter: single-level grids at different resolutions"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import oracle as O
from xrnerf_amd import ops, synthetic as S
dev = torch.device('cuda:0')
grid = S.lego_density_grid(); bf = O.bitfield_given_mean(grid, O.density_mean(grid))
o, d, _ = S.training_rays(S.lego_cameras(20), 18000, seed=3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
c, _, ns, cnt = ops.rays_sampler(t(o), t(d), t(bf), (0., 1.), 0.05, 1 / 256, 18000 * 64, 0)
n = int(cnt[1]); c = c[:n]
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
for res in (16, 32, 64, 128, 256, 512, 1024, 2048):
    meta = ops.GridMeta(n_levels=1, base_resolution=res + 1, per_level_scale=1.0)
    g = torch.zeros(meta.n_params, device=dev)
    denc = torch.randn((2, (n + 63) // 64 * 64), device=dev)
    tb = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g))
    table = torch.randn(meta.n_params, device=dev)
    tf = timeit(lambda: ops.hashgrid_fwd(table, c[:, :3], meta))
    print('res %5d entries %7d  bwd %.4f ms  fwd %.4f ms' % (res, meta.n_params // 2, tb, tf))
