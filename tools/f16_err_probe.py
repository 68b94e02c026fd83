import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import f16_reference as R
from xrnerf_amd import ops
dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for n in (5000, 8193, 32768, 33000, 40001, 100000):
    rng = np.random.default_rng(n)
    enc = rng.normal(0, 0.5, (n, 32)).astype(np.float32); dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    wd = rng.uniform(-0.4, 0.4, 3072).astype(np.float32); wc = rng.uniform(-0.3, 0.3, 7168).astype(np.float32)
    draw = rng.normal(0, 1e-2, (n, 4)).astype(np.float32)
    ld = (n + 63) // 64 * 64
    enc_t = torch.zeros((32, ld), device=dev); enc_t[:, :n] = T(enc).t()
    ops.set_precision('f16')
    gwd = torch.zeros(3072, device=dev); gwc = torch.zeros(7168, device=dev)
    denc_t = ops.nerf_mlp_bwd(enc_t, T(dirs), n, T(wd), T(wc), 1, 2, T(draw), gwd, gwc).cpu().numpy()
    ops.set_precision('f32')
    g32d = torch.zeros(3072, device=dev); g32c = torch.zeros(7168, device=dev)
    ops.nerf_mlp_bwd(enc_t, T(dirs), n, T(wd), T(wc), 1, 2, T(draw), g32d, g32c)
    de, rwd, rwc = R.backward(enc, dirs, wd, wc, draw)
    e = lambda a, b: '%.2e/%.2e' % (np.abs(a - b).max(), np.abs(b).max())
    print(n, 'f16 vs ref: denc', e(denc_t[:, :n].T, de), 'wd', e(gwd.cpu().numpy(), rwd), 'wc', e(gwc.cpu().numpy(), rwc),
          '| f32 kernel vs f16 ref: wd', e(g32d.cpu().numpy(), rwd), 'wc', e(g32c.cpu().numpy(), rwc), flush=True)
    # per-layer blocks of wc
    d = np.abs(gwc.cpu().numpy() - rwc); print('     wc blocks c0 %.2e c1 %.2e c2 %.2e' % (d[:2048].max(), d[2048:6144].max(), d[6144:].max()))
