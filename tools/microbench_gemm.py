"""fp32-MFMA linear kernels vs torch (hipBLASLt) on the shapes of the 8x256 NeRF MLP; python tools/microbench_gemm.py [M]"""
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


for N, K in ((256, 256), (256, 96), (256, 352), (128, 284), (260, 256), (4, 128)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y = ops.linear_forward(x, w, b, True)
    fl = 2.0 * M * N * K
    rows = [('fwd', lambda: ops.linear_forward(x, w, b, True), lambda: torch.relu(torch.nn.functional.linear(x, w, b))),
            ('dX', lambda: ops.linear_backward_input(dy, y, w), lambda: (dy * (y > 0)) @ w),
            ('dW', lambda: ops.linear_backward_weight(dy, y, x), lambda: (dy * (y > 0)).t() @ x),
            ('db', lambda: ops.linear_backward_bias(dy, y), lambda: (dy * (y > 0)).sum(0))]
    for name, f, t in rows:
        us, ut = timeit(f), timeit(t)
        print('M=%d N=%3d K=%3d %-3s  mfma kernel %8.1f us (%6.1f TFLOP/s)   torch %8.1f us (%6.1f TFLOP/s)' % (
            M, N, K, name, us, 0 if name == 'db' else fl / us / 1e6, ut, 0 if name == 'db' else fl / ut / 1e6))
