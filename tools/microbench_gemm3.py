"""linear layers on split operands (csrc/xr_gemm3.hip) against the fp32-MFMA / in-kernel-split kernels of xr_gemm.hip on the shapes of
the 8x256 NeRF MLP: python tools/microbench_gemm3.py [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3


# correctness at a ragged size
torch.manual_seed(0)
m, n, k = 300, 136, 96
x, w, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.3, torch.randn(n, device=dev)
ref = x.double() @ w.double().t() + b.double()
xp, wp = ops.p3_split(x), ops.p3_split(w)
assert torch.equal(xp.float().sum(0)[:, :k], x), 'split is not exact'
yp = ops.p3_gemm_nt(xp, wp, n, bias=b, relu=True)
yf = torch.empty(m, n, device=dev)
ops.p3_gemm_nt(xp, wp, n, bias=b, relu=True, out_f32=yf)
err = (yf.double() - ref.clamp_min(0)).abs().max() / ref.abs().max()
errp = (yp.float().sum(0)[:, :n].double() - ref.clamp_min(0)).abs().max() / ref.abs().max()
print('ragged %dx%dx%d: fp32 out err %.2e of max, planes out err %.2e of max, planes == fp32 out: %s' %
      (m, n, k, float(err), float(errp), bool(torch.equal(yp.float().sum(0)[:, :n], yf))))
mask = (torch.rand(m, n, device=dev) > 0.5).float()
mp = ops.p3_split(mask)
ops.p3_gemm_nt(xp, wp, n, mask_hi=mp[0], out_f32=yf)
zp = ops.p3_gemm_nt(xp, wp, n, mask_hi=mp[0])
errm = (yf.double() - (ref - b.double()) * mask.double()).abs().max() / ref.abs().max()
print('masked: err %.2e, planes == fp32: %s' % (float(errm), bool(torch.equal(zp.float().sum(0)[:, :n], yf))))

for (N, K) in ((256, 256), (256, 96), (256, 352), (128, 288)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.randn(N, device=dev)
    xp, wp = ops.p3_split(x), ops.p3_split(w)
    out = ops.p3_planes(M, N, dev)
    yf = torch.empty(M, N, device=dev)
    flops = 2.0 * M * N * K
    t_split = timeit(lambda: ops.p3_split(x, out=xp))
    t_p = timeit(lambda: ops.p3_gemm_nt(xp, wp, N, bias=b, relu=True, out_planes=out))
    t_f = timeit(lambda: ops.p3_gemm_nt(xp, wp, N, bias=b, relu=True, out_f32=yf))
    t_m = timeit(lambda: ops.p3_gemm_nt(xp, wp, N, mask_hi=out[0], out_planes=out))
    t_old = timeit(lambda: ops.linear_forward(x, w, b, True)) if K % 4 == 0 else float('nan')
    ref = torch.relu(torch.nn.functional.linear(x[:4096].double(), w.double(), b.double()))
    err = (yf[:4096].double() - ref).abs().max() / ref.abs().max()
    print('M=%d N=%3d K=%3d  planes->planes %7.1f us (%6.1f TF)  planes->fp32 %7.1f us  masked planes->planes %7.1f us | in-kernel split %7.1f us (%6.1f TF) | '
          'split of x %6.1f us | err %.1e' % (M, N, K, t_p, flops / t_p / 1e6, t_f, t_m, t_old, flops / t_old / 1e6, t_split, float(err)))
