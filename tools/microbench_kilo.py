"""KiloNeRF frame time on the synthetic Lego-shaped scene (1440 networks, 144x256x160 occupancy, 384 samples per ray):
python tools/microbench_kilo.py [H W]   (profile: rocprofv3 --kernel-trace --stats -- python tools/microbench_kilo.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import kilo, synthetic as S

dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
mlp, gmin, gmax = kilo.synthetic_scene(dev, seed=1)
focal = 1111.111 * W / 800
poses = kilo.orbit_poses(6)
for fused in (True, False):
    for it in range(2):
        rgb, disp, acc = kilo.render_frame(mlp, gmin, gmax, poses[0], H, W, focal, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for f in range(n):
        rgb, disp, acc = kilo.render_frame(mlp, gmin, gmax, poses[f % len(poses)], H, W, focal, fused=fused)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print('%dx%d x 384 samples, %s: %.2f ms/frame (ray generation included), mean acc %.3f, peak memory %.2f GB' % (
        H, W, 'fused frame call' if fused else 'module path (z_vals, dense raw, NerfRender)', ms, float(acc.mean()),
        torch.cuda.max_memory_allocated() / 2**30))
