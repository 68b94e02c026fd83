"""micro-benchmark of the hash-grid kernels on realistic (ray-ordered) samples; env XR_HG_BWD_MODE / XR_HG_ORDER
select experimental variants.  usage: python tools/microbench_encode.py [n_rays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import oracle as O
from xrnerf_amd import ops, synthetic as S

dev = torch.device('cuda:0')
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 18000
grid = S.lego_density_grid(); bf = O.bitfield_given_mean(grid, O.density_mean(grid))
poses = S.lego_cameras(20)
o, d, _ = S.training_rays(poses, n_rays, seed=3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
c, _, ns, cnt = ops.rays_sampler(t(o), t(d), t(bf), (0., 1.), 0.05, 1 / 256, n_rays * 64, 0)
n = int(cnt[1]); c = c[:n]
meta = ops.GridMeta()
table = t(S.hash_table(meta.n_params))
enc = ops.hashgrid_fwd(table, c[:, :3], meta)
denc = torch.randn_like(enc)
g = torch.zeros(meta.n_params, device=dev)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
tf = timeit(lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=enc.shape[1]))
tb = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g))
tz = timeit(lambda: g.zero_())
print('mode=%s order=%s n=%d fwd %.3f ms (%.0f GB/s algo)  bwd %.3f ms (%.0f GB/s algo)  zero %.3f ms' % (
    os.environ.get('XR_HG_BWD_MODE', '0'), os.environ.get('XR_HG_ORDER', '0'), n, tf, n * 1164 / tf / 1e6, tb, n * 2188 / tb / 1e6, tz))
# correctness of the variant vs mode 0 reference result computed by the oracle on a subset
sub = 20000
x = c[:sub, :3].contiguous().cpu().numpy(); dy = denc[:, :sub].t().contiguous().cpu().numpy()
ref = O.hashgrid_bwd(x, dy, O.GridMeta())
g.zero_(); dsub = torch.zeros((32, (sub + 63) // 64 * 64), device=dev); dsub[:, :sub] = denc[:, :sub]
ops.hashgrid_bwd(c[:sub, :3], dsub, meta, g)
print('   bwd max err vs oracle: %.3e (ref max %.3e)' % (np.abs(g.cpu().numpy() - ref).max(), np.abs(ref).max()))
