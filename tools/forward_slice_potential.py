"""How many of the FORWARD's rows (lookup + fused MLP) does a training step evaluate for nothing?  Rows behind an opaque surface have an
exactly-zero transmittance: their rgb weight and their dL/d(raw) are exact zeros (the backward already skips them: the live-row list).  A
forward in depth slices -- the first k samples of every ray, then the rest of the rays whose transmittance is still positive -- would skip
them too.  Measured on the bench scene after 320 iterations: per ray the number of marched and of live rows, and the rows a two- or
three-slice forward would evaluate.  usage: python tools/forward_slice_potential.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
for name, kw in (('synthetic Lego-shaped scene, 100 cameras', dict(n_img=100)),):
    tr = Trainer(dev, native_loop=False, **kw)
    for _ in range(int(os.environ.get('ITERS', '320'))):
        tr.step()
    torch.cuda.synchronize()
    s = tr.net.sampler
    ns = s.rays_numsteps_compacted.cpu().numpy().astype(np.int64)            # (count, base) per ray, clipped to the batch
    draw = tr.net._last['raw'].new_empty(0)
    b = tr.net._step_bufs[tr.net._step_turn]
    draw = b.draw.cpu().numpy()
    n_valid = int(s.n_valid_dev.cpu()[0])
    cnt, base = ns[:, 0], ns[:, 1]
    live = np.zeros_like(cnt)
    for r in np.nonzero(cnt)[0]:
        rows = draw[base[r]:base[r] + cnt[r]]
        nz = np.nonzero(np.any(rows != 0, axis=1))[0]
        live[r] = (nz[-1] + 1) if len(nz) else 0                              # rows up to the last live one must be evaluated
    total = int(cnt.sum())
    print('%s: %d rays (%d with samples), %d rows, valid %d; rows up to each ray\'s last live row: %d (%.3f)' %
          (name, len(cnt), int((cnt > 0).sum()), total, n_valid, int(live.sum()), live.sum() / max(total, 1)))
    for ks in ((4,), (8,), (12,), (16,), (4, 12), (8, 24)):
        ev = np.zeros_like(cnt)
        alive = cnt > 0
        lo = 0
        for k in ks + (1 << 30,):
            seg = np.clip(cnt - lo, 0, k - lo) * alive                        # rows [lo, k) of the rays still alive
            ev += seg
            alive = alive & (live > k)                                        # still transparent behind row k - 1
            lo = k
        print('   slices at %-10s -> %7d rows evaluated (%.3f of all), %d launches of lookup + MLP' % (ks, int(ev.sum()), ev.sum() / max(total, 1), len(ks) + 1))
