"""fused MLP forward / backward in both arithmetic modes at 2^18 samples (events on the launch stream)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
n = 1 << 18
rng = np.random.default_rng(0)
enc_t = torch.randn((32, n), device=dev) * 0.5
coords = torch.rand((n, 7), device=dev)
wd = (torch.rand(3072, device=dev) - 0.5) * 0.8; wc = (torch.rand(7168, device=dev) - 0.5) * 0.6
draw = torch.randn((n, 4), device=dev) * 1e-2
raw = torch.empty((n, 4), device=dev); denc = torch.empty_like(enc_t)
gwd, gwc = torch.zeros(3072, device=dev), torch.zeros(7168, device=dev)
ndev = torch.tensor([n], dtype=torch.int32, device=dev)
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
for mode, peak in (('f32', 157.3), ('f16', 2500.0)):
    ops.set_precision(mode)
    tf = timeit(lambda: ops.nerf_mlp_fwd(enc_t, coords[:, 4:], n, wd, wc, 1, 2, raw=raw, n_dev=ndev))
    td = timeit(lambda: ops.nerf_mlp_fwd(enc_t, None, n, wd, None, 1, 2, raw=raw))
    tb = timeit(lambda: ops.nerf_mlp_bwd(enc_t, coords[:, 4:], n, wd, wc, 1, 2, draw, gwd, gwc, denc_t=denc, n_dev=ndev))
    print('%s: fwd %.1f us (%.0f TFLOP/s, %.3f of %g)  density-only fwd %.1f us  bwd %.1f us (%.0f TFLOP/s, %.3f)' % (
        mode, tf, n * 20480 / tf / 1e6, n * 20480 / tf / 1e6 / peak, peak, td, tb, n * 59392 / tb / 1e6, n * 59392 / tb / 1e6 / peak), flush=True)
