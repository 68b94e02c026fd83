"""fraction of marched samples whose dL/draw row is exactly zero in steady-state training steps (those rows are skipped by the
scatter's binning pass; a compaction in front of the MLP backward would skip them there too)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['XRNERF_STEP'] = 'py'
import torch
from xrnerf_amd.train import Trainer
from xrnerf_amd import ops
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=20)
for _ in range(330): tr.step()
seen = []
orig = ops.nerf_mlp_bwd
def spy(enc_t, dirs, n, wd, wc, nhd, nhc, draw, *a, **k):
    nv = int(k['n_dev'][0]) if k.get('n_dev') is not None else n
    d = draw[:nv]
    z = (d == 0).all(1)
    tiles = z[: nv // 32 * 32].reshape(-1, 32).all(1)
    small = (d.abs().amax(1) < 1e-25)
    seen.append((nv, float(z.float().mean()), float(tiles.float().mean()), float(small.float().mean())))
    return orig(enc_t, dirs, n, wd, wc, nhd, nhc, draw, *a, **k)
ops.nerf_mlp_bwd = spy
for _ in range(8): tr.step()
try:
    smp = tr.sampler if hasattr(tr, 'sampler') else tr.net.sampler
    c = smp.rays_numsteps_compacted[:, 0].float()
    qs = torch.tensor([0.5, 0.9, 0.99, 1.0], device=c.device)
    print('samples per ray: mean %.1f  rays with 0: %.2f  > 64: %.3f  quantiles 0.5/0.9/0.99/max %s' % (
        float(c.mean()), float((c == 0).float().mean()), float((c > 64).float().mean()), [int(v) for v in torch.quantile(c, qs)]))
except Exception as e:
    print('ray-length statistics unavailable:', e)
for s in seen: print('valid rows %d  zero-gradient rows %.3f  all-zero 32-row tiles %.3f  rows with |g| < 1e-25 %.3f' % s)
