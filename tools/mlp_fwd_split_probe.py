"""forward of the fused MLP (1, 2) by operand arithmetic: fp32 MFMA (reference of this probe, against a float64 statement), 3-way bf16 split,
2-way fp16 split (XR_MLP_FWD_SPLIT=h2, read once per process: run once per setting) -- error against float64 at three input scales, time at 2^18"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops, synthetic as S
dev = torch.device('cuda:0'); n = 1 << 18
wd, wc = S.mlp_weights(32, 64, 1, 16, 4), S.mlp_weights(32, 64, 2, 16, 5)
twd, twc = torch.from_numpy(wd).to(dev), torch.from_numpy(wc).to(dev)
g = torch.Generator().manual_seed(0)
dirs = torch.rand((n, 3), generator=g).to(dev)
def f64(enc, dirs):
    e = enc.double().t(); W = [w.double() for w in ops._net_layers(twd, 1)]
    h = torch.relu(e @ W[0].t()); dout = h @ W[1].t()
    cin = torch.cat([dout[:, 1:16], ops.sh4(dirs).double(), torch.ones((e.shape[0], 1), dtype=torch.float64, device=dev)], 1)
    C = [w.double() for w in ops._net_layers(twc, 2)]
    h = torch.relu(cin @ C[0].t()); h = torch.relu(h @ C[1].t()); cout = h @ C[2].t()
    return torch.cat([cout[:, :3], dout[:, :1]], 1)
kind = os.environ.get('XR_MLP_FWD_SPLIT', 'b3')
for scale in (1e-4, 1e-2, 0.5, 20.0):
    enc = (torch.randn((32, n), generator=g) * scale).to(dev)
    ref = f64(enc[:, :20000], dirs[:20000])
    out = {}
    for mode in ('mfma', 'bf16x3'):
        ops.set_f32_forward(mode)
        raw = ops.nerf_mlp_fwd(enc, dirs, n, twd, twc, 1, 2)
        out[mode] = float((raw[:20000].double() - ref).abs().max())
    print('%s input scale %-7g |raw| max %.3g   max abs error vs float64: fp32 MFMA %.3g   split forward (%s) %.3g' % (
        kind, scale, float(ref.abs().max()), out['mfma'], kind, out['bf16x3']), flush=True)
ops.set_f32_forward('bf16x3')
enc = (torch.randn((32, n), generator=g) * 0.1).to(dev); raw = torch.empty((n, 4), device=dev)
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
print('%s  forward at 2^18 rows: %.1f us   density only: %.1f us' % (kind, 1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, dirs, n, twd, twc, 1, 2, raw=raw)),
      1e3 * timeit(lambda: ops.nerf_mlp_fwd(enc, None, n, twd, None, 1, 2, raw=raw))))
