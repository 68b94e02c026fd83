"""the three linear products of the 8x256 NeRF MLP at M = 131072: fp32-MFMA kernel vs the 3-way bf16 operand split
(XR_GEMM_F32 is read per launch), and their deviation from the fp64 product"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
for N, K in ((256, 256), (256, 352)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    ref = (x[:4096].double() @ w.double().t() + b.double()).clamp_min(0)
    for kind in ('mfma', 'bf16x3', 'bf16x3all'):
        os.environ['XR_GEMM_F32'] = kind
        y = ops.linear_forward(x, w, b, True)
        err = float((y[:4096].double() - ref).abs().max() / ref.abs().max())
        t = [timeit(lambda: ops.linear_forward(x, w, b, True)), timeit(lambda: ops.linear_backward_input(dy, y, w)),
             timeit(lambda: ops.linear_backward_weight(dy, y, x))]
        print('M=%d N=%3d K=%3d %-9s fwd %7.1f us (%5.1f TF)  dX %7.1f us (%5.1f TF)  dW %7.1f us (%5.1f TF)   fwd err vs fp64 %.1e of max' % (
            M, N, K, kind, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[2], fl / t[2] / 1e6, err), flush=True)
