"""micro-benchmark of the fused MLP kernels at 2^18 samples: python tools/microbench_mlp.py [nhd nhc [live_fraction]]
(1, 2): the register-resident kernels; any other depth: the streamed kernels (weights through LDS layer by layer)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops, synthetic as S
nhd, nhc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 2)
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device('cuda:0'); n = 1 << 18
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
enc = torch.randn((32, n), device=dev) * 0.1
dirs = torch.rand((n, 3), device=dev)
wd, wc = t(S.mlp_weights(32, 64, nhd, 16, 4)), t(S.mlp_weights(32, 64, nhc, 16, 5))
draw = torch.randn((n, 4), device=dev)
if frac < 1.0:
    draw[torch.rand((n,), device=dev) >= frac] = 0.0
gwd, gwc = torch.zeros_like(wd), torch.zeros_like(wc)
raw = torch.empty((n, 4), device=dev); denc = torch.empty_like(enc)
mac = lambda nh: 32 * 64 + (nh - 1) * 64 * 64 + 64 * 16
ff = 2 * (mac(nhd) + mac(nhc))
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
tf = timeit(lambda: ops.nerf_mlp_fwd(enc, dirs, n, wd, wc, nhd, nhc, raw=raw))
td = timeit(lambda: ops.nerf_mlp_fwd(enc, None, n, wd, None, nhd, nhc, raw=raw))
live = ops.live_rows(draw, n) if frac < 1.0 else None
tb = timeit(lambda: ops.nerf_mlp_bwd(enc, dirs, n, wd, wc, nhd, nhc, draw, gwd, gwc, denc_t=denc, live=live))
nl = int(live[1][0]) if live is not None else n
print('(%d, %d) n=%d  fwd %.3f ms (%.1f TFLOP/s)  density-only %.3f ms  bwd over %d rows %.3f ms (%.1f TFLOP/s)' % (
    nhd, nhc, n, tf, n * ff / tf / 1e9, td, nl, tb, nl * (3 * ff - 2048) / tb / 1e9))
