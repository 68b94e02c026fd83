"""hash-grid scatter: atomic kernel vs LDS-partition scan kernel on marched Lego samples, per level group.
XR_SCATTER / XR_SCAN_LEVELS are read once per process -> this script re-executes itself per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def child():
    import numpy as np, torch
    import oracle as O
    from xrnerf_amd import ops, synthetic as S
    dev = torch.device('cuda:0')
    grid = S.lego_density_grid(); bf = O.bitfield_given_mean(grid, O.density_mean(grid))
    o, d, _ = S.training_rays(S.lego_cameras(20), 18000, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    c, _, ns, cnt = ops.rays_sampler(t(o), t(d), t(bf), (0., 1.), 0.05, 1 / 256, 18000 * 64, 0)
    n = min(int(cnt[1]), 1 << 18); c = c[:n].contiguous()
    meta = ops.GridMeta()
    ld = (n + 63) // 64 * 64
    denc = torch.randn((32, ld), device=dev)
    g = torch.zeros(meta.n_params, device=dev)

    def timeit(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
    out = ['n=%d mode=%s cap=%s' % (n, os.environ.get('XR_SCATTER_GROUPS'), os.environ.get('XR_SCAN_LEVELS'))]
    for lv in ((0, 16), (5, 16), (8, 16)):
        out.append('levels %-8s %.4f ms' % (lv, timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, levels=lv))))
    # correctness of the scan kernel against the atomic kernel is in tests/; here a quick checksum
    g.zero_(); ops.hashgrid_bwd(c[:, :3], denc, meta, g); out.append('sum|g| %.6e' % float(g.abs().sum()))
    print('\n'.join(out), flush=True)


if __name__ == '__main__':
    if os.environ.get('XR_CHILD') == '1':
        child()
    else:
        for mode, cap in (('1', '16'), ('2', '16'), ('3', '16')):
            env = dict(os.environ, XR_CHILD='1', XR_SCATTER_GROUPS=mode, XR_SCAN_LEVELS=cap)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
