"""Scatter micro-benchmark (xr_scatter.hip) on marched Lego samples (ray-ordered, ~2.6e5 rows), over every row and on a live-row list
shaped like the training step's (per ray: the leading ~47 % of its samples), by level range, with and without XR_SCATTER_OVERWRITE,
and the agreement with the atomic kernel.  XR_SC_TEST (layout parameters) is read once per process, so the script re-executes
itself per setting; XRNERF_LIB selects another build.  usage: python tools/microbench_scatter3.py [quick]"""
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
    # live list: per ray the leading 47 % of its samples (what stays in front of the surface)
    nsn = ns.cpu().numpy().astype(np.int64)
    live = []
    for cntr, base in nsn:
        if base + cntr <= n and cntr > 0:
            k = max(1, int(round(0.47 * cntr)))
            live.append(np.arange(base, base + k))
    live = np.concatenate(live).astype(np.int32)
    rows = torch.zeros(n, dtype=torch.int32, device=dev); rows[:len(live)] = t(live)
    nl = torch.tensor([len(live), 0, 0, 0], dtype=torch.int32, device=dev)

    def timeit(f, reps=30):
        # ONE call between two events, the device idle before it (median): what the entry costs inside a training step.  Timing a
        # train of calls (XR_B2B=1) lets the next call's helper-stream work start under the previous call's tail and reads
        # ~25 us lower -- the figures of profiles/r03_scatter3_*variants.txt before this note were taken that way.
        for _ in range(5): f()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        if os.environ.get('XR_B2B'):
            a.record()
            for _ in range(reps): f()
            b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
        ts = []
        for _ in range(reps):
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))
    tag = ' '.join('%s=%s' % (k[3:], os.environ[k]) for k in sorted(os.environ) if k.startswith('XR_SC'))
    out = []
    if os.environ.get('XR_ONLY'):          # one configuration only (clean rocprofv3 kernel averages): levels 0-16, live list, overwrite
        us = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, live=(rows, nl), overwrite=True), 50)
        print('%-34s ONLY levels (0, 16) live list (%d): %.1f us' % (tag, len(live), us), flush=True)
        return
    for lv in ((0, 16), (5, 16), (3, 16), (0, 3), (0, 5), (8, 16), (0, 8), (5, 9), (12, 16), (3, 5)):
        if os.environ.get('XR_QUICK') and lv not in ((0, 16), (5, 16), (0, 5), (5, 9), (12, 16), (3, 5)): continue
        us_all = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, levels=lv, overwrite=True))
        us_live = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, levels=lv, live=(rows, nl), overwrite=True))
        out.append('%-34s levels %-8s  all rows (%d) %.1f us   live list (%d) %.1f us' % (tag, lv, n, us_all, len(live), us_live))
        if lv == (0, 16):
            out[-1] += '   frac(live) %.3f' % (len(live) * 2188 / us_live / 1e3 / 8000)
    us_add = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, live=(rows, nl), overwrite=False))
    out.append('%-34s add mode (no overwrite), live list: %.1f us' % (tag, us_add))
    # agreement with the atomic scatter on the live list
    g.zero_(); ops.hashgrid_bwd(c[:, :3], denc, meta, g, live=(rows, nl), overwrite=True)
    g0 = torch.zeros_like(g); ops.hashgrid_bwd(c[:, :3], denc, meta, g0, live=(rows, nl), use_workspace=False)
    out.append('   max |binned - atomic| %.3e (max |g| %.3e)' % (float((g - g0).abs().max()), float(g0.abs().max())))
    print('\n'.join(out), flush=True)


if __name__ == '__main__':
    if os.environ.get('XR_CHILD') == '1':
        child()
    else:
        quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
        runs = [dict(), dict(XR_SC_TEST='rl_chunks=8'), dict(XR_SC_TEST='block=1024')]
        for env in runs:
            e = dict(os.environ, XR_CHILD='1', **env)
            if quick: e['XR_QUICK'] = '1'
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=False)
