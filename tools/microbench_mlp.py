"""micro-benchmark of the fused MLP kernels at 2^18 samples: python tools/microbench_mlp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops, synthetic as S
dev = torch.device('cuda:0'); n = 1 << 18
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
enc = torch.randn((32, n), device=dev) * 0.1
dirs = torch.rand((n, 3), device=dev)
wd, wc = t(S.mlp_weights(32, 64, 1, 16, 4)), t(S.mlp_weights(32, 64, 2, 16, 5))
draw = torch.randn((n, 4), device=dev)
gwd, gwc = torch.zeros_like(wd), torch.zeros_like(wc)
raw = torch.empty((n, 4), device=dev); denc = torch.empty_like(enc)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
tf = timeit(lambda: ops.nerf_mlp_fwd(enc, dirs, n, wd, wc, 1, 2, raw=raw))
td = timeit(lambda: ops.nerf_mlp_fwd(enc, None, n, wd, None, 1, 2, raw=raw))
tb = timeit(lambda: ops.nerf_mlp_bwd(enc, dirs, n, wd, wc, 1, 2, draw, gwd, gwc, denc_t=denc))
print('n=%d  fwd %.3f ms (%.1f TFLOP/s)  density-only %.3f ms  bwd %.3f ms (%.1f TFLOP/s)' % (
    n, tf, n * 20480 / tf / 1e9, td, tb, n * 59392 / tb / 1e9))
