"""micro-benchmark of the Mip-NeRF stage kernels (xrnerf_amd/csrc/xr_mip.hip) at the training batch (1024 x 128) and at
an 800x800 frame's worth of rays in 65536-ray chunks; prints per-kernel time and algorithmic GB/s.
usage: python tools/microbench_mip.py [n_rays ...]      (profile: rocprofv3 --kernel-trace --stats -- python tools/microbench_mip.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import mip, ops

dev = torch.device('cuda:0')
S = 128


def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3   # us


for R in ([int(a) for a in sys.argv[1:]] or [1024, 65536]):
    rays = mip.synthetic_multiscale_rays(R, dev, seed=2)
    zr = torch.rand(R, S + 1, device=dev)
    z = ops.mip_zvals(rays['near'], rays['far'], S + 1, False, zr)
    out = torch.empty((R * S, 123), device=dev)
    raw = torch.randn(R, S, 4, device=dev)
    g = torch.randn(R, 3, device=dev)
    w = ops.mip_render_forward(raw, z, rays['rays_d'], -1., 0.001, True)[3]
    rows = [
        ('zvals', lambda: ops.mip_zvals(rays['near'], rays['far'], S + 1, False, zr), R * (S + 1) * 8 + R * 8),
        ('encode', lambda: ops.mip_encode(rays['rays_o'], rays['rays_d'], rays['viewdirs'], rays['radii'], z, 0, 16, 0, 4,
                                          True, 'cone', out=out), R * S * 123 * 4 + R * (44 + (S + 1) * 4)),
        ('render_fwd', lambda: ops.mip_render_forward(raw, z, rays['rays_d'], -1., 0.001, True), R * S * 20 + R * ((S + 1) * 4 + 32)),
        ('render_bwd', lambda: ops.mip_render_backward(raw, z, rays['rays_d'], g, -1., 0.001, True), R * S * 32 + R * ((S + 1) * 4 + 24)),
        ('resample', lambda: ops.mip_resample(z, w, 0.01, zr), R * ((S + 1) * 12 + S * 4)),
    ]
    for name, f, nbytes in rows:
        us = timeit(f)
        print('R=%6d S=%d %-10s %9.1f us  %8.1f GB/s algorithmic (%.1f %% of 8 TB/s)  %.2f G samples/s' % (
            R, S, name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80.0, R * S / us / 1e3))
