"""fused MLP backward with the live-row compaction: dense and compositor-like sparse dL/d(raw) (runs of exactly-zero rows,
`dead` of the rows), both precisions; run once with XR_MLP_LIVE=0 for the backward over every row"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
def sparse(n, dead):
    d = rng.normal(0, 1e-2, (n, 4)).astype(np.float32)
    pos = 0
    while pos < n:
        run = int(rng.integers(4, 60))
        if rng.uniform() < dead: d[pos:pos + run] = 0
        pos += run
    return torch.from_numpy(d).to(dev)
print('XR_MLP_LIVE=%s' % os.environ.get('XR_MLP_LIVE', '1'))
for n in (1 << 16, 1 << 17, 1 << 18):
    enc_t = torch.randn((32, n), device=dev) * 0.5
    coords = torch.rand((n, 7), device=dev)
    wd = (torch.rand(3072, device=dev) - 0.5) * 0.8; wc = (torch.rand(7168, device=dev) - 0.5) * 0.6
    denc = torch.empty_like(enc_t)
    gwd, gwc = torch.zeros(3072, device=dev), torch.zeros(7168, device=dev)
    ndev = torch.tensor([n], dtype=torch.int32, device=dev)
    for dead in (0.0, 0.55, 0.9):
        draw = sparse(n, dead)
        frac = float((draw == 0).all(1).float().mean())
        out = []
        for mode in ('f32', 'f16'):
            ops.set_precision(mode)
            out.append('%s %.1f us' % (mode, timeit(lambda: ops.nerf_mlp_bwd(enc_t, coords[:, 4:], n, wd, wc, 1, 2, draw, gwd, gwc, denc_t=denc, n_dev=ndev))))
        print('n %7d  dead rows %.2f:  %s' % (n, frac, '   '.join(out)), flush=True)
