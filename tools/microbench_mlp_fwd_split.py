"""fp32 forward of the fused MLP at 2^18 samples: fp32-MFMA kernel vs the 3-way bf16 operand split (events on the launch
stream), their largest deviation from each other, and the 14.3 M-sample size of a rendered frame"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
wd = (torch.rand(3072, device=dev) - 0.5) * 0.8; wc = (torch.rand(7168, device=dev) - 0.5) * 0.6
for n, reps in ((1 << 18, 30), (14300000, 5)):
    enc_t = torch.randn((32, n), device=dev) * 0.5
    coords = torch.rand((n, 7), device=dev)
    ndev = torch.tensor([n], dtype=torch.int32, device=dev)
    out = {}
    for kind in ('mfma', 'bf16x3'):
        ops.set_f32_forward(kind)
        raw = torch.empty((n, 4), device=dev)
        tf = timeit(lambda: ops.nerf_mlp_fwd(enc_t, coords[:, 4:], n, wd, wc, 1, 2, raw=raw, n_dev=ndev), reps)
        td = timeit(lambda: ops.nerf_mlp_fwd(enc_t, None, n, wd, None, 1, 2, raw=raw), reps)
        ops.nerf_mlp_fwd(enc_t, coords[:, 4:], n, wd, wc, 1, 2, raw=raw, n_dev=ndev)
        out[kind] = raw.clone()
        print('n %9d  %-7s fwd %8.1f us (%.0f TFLOP/s algorithmic)   density-only fwd %8.1f us' % (n, kind, tf, n * 20480 / tf / 1e6, td), flush=True)
    d = (out['mfma'] - out['bf16x3']).abs().max().item()
    print('n %9d  max |mfma - bf16x3| = %.3e at max |raw| = %.3e' % (n, d, out['mfma'].abs().max().item()), flush=True)
