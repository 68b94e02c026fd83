"""Round-2 hash-grid micro-benchmark: forward gather variants (XR_HG_FWD_MODE / XR_HG_WSH) and scatter generations
(XR_SC_MODE) on marched Lego samples (ray-ordered, ~2^18).  The switches are read once per process, so the script
re-executes itself per setting.  usage: python tools/microbench_hash.py [fwd|bwd|all]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def child(which):
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
    table = t(S.hash_table(meta.n_params))
    ld = (n + 63) // 64 * 64
    enc = torch.empty((32, ld), device=dev)
    ndev = torch.tensor([n], dtype=torch.int32, device=dev)

    def timeit(f, reps=30):
        for _ in range(5): f()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
    tag = 'FWD_MODE=%s WSH=%s SC_MODE=%s' % (os.environ.get('XR_HG_FWD_MODE'), os.environ.get('XR_HG_WSH'), os.environ.get('XR_SC_MODE'))
    if which == 'fwd':
        us = timeit(lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=ld, n_dev=ndev))
        # the 14 M-sample render-sized launch too (n repeated 16x along fresh rays is not needed: reuse rows)
        print('%-44s n=%d fwd %.1f us  %.0f GB/s algo  frac %.3f' % (tag, n, us, n * 1164 / us / 1e3, n * 1164 / us / 1e3 / 8000), flush=True)
        if os.environ.get('XR_FWD_ABLATE'):
            x3 = c[:, :3].contiguous()
            print('   compact [n,3] positions: %.1f us' % timeit(lambda: ops.hashgrid_fwd(table, x3, meta, enc_t=enc, ld=ld, n_dev=ndev)), flush=True)
            for lv in ((0, 5), (5, 16), (5, 8), (8, 13), (13, 16), (15, 16), (0, 1), (4, 5)):
                print('   levels %-8s %.1f us' % (lv, timeit(lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=ld, n_dev=ndev, levels=lv))), flush=True)
        ref = O.hashgrid_fwd(S.hash_table(meta.n_params), c[:4096, :3].cpu().numpy(), O.GridMeta())
        got = enc[:, :4096].t().cpu().numpy()
        print('   bit-exact vs oracle on 4096 rows:', bool((got == ref).all()), flush=True)
    else:
        denc = torch.randn((32, ld), device=dev)
        g = torch.zeros(meta.n_params, device=dev)
        for lv in ((0, 16), (5, 16), (0, 5)):
            us = timeit(lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, n_dev=ndev, levels=lv))
            print('%-44s n=%d bwd levels %-8s %.1f us%s' % (tag, n, lv, us, '  %.0f GB/s algo frac %.3f' % (n * 2188 / us / 1e3, n * 2188 / us / 1e3 / 8000) if lv == (0, 16) else ''), flush=True)
        sub = 30000
        x = c[:sub, :3].contiguous().cpu().numpy(); dy = denc[:, :sub].t().contiguous().cpu().numpy()
        ref = O.hashgrid_bwd(x, dy, O.GridMeta())
        g.zero_(); dsub = torch.zeros((32, (sub + 63) // 64 * 64), device=dev); dsub[:, :sub] = denc[:, :sub]
        ops.hashgrid_bwd(c[:sub, :3], dsub, meta, g)
        print('   bwd max err vs oracle: %.3e (ref max %.3e)' % (np.abs(g.cpu().numpy() - ref).max(), np.abs(ref).max()), flush=True)


if __name__ == '__main__':
    if os.environ.get('XR_CHILD'):
        child(os.environ['XR_CHILD'])
    else:
        what = sys.argv[1] if len(sys.argv) > 1 else 'all'
        runs = []
        if what in ('fwd', 'all'):
            # 8 = the round-1 kernel (level-major mapping, divergent 16-B pair gathers); 0 = same mapping, plain 8-B gathers;
            # 16 = two-list balanced mapping (default); 24 = balanced + pair gathers; 20 = balanced + levels 0-1 from LDS;
            # 18 = balanced + non-temporal loads
            modes = os.environ.get('XR_FWD_MODES', '8,0,16,24,20,18').split(',')
            runs += [('fwd', dict(XR_HG_FWD_MODE=m, XR_HG_WSH='0,1,2')) for m in modes]
        if what in ('bwd', 'all'):
            runs += [('bwd', dict(XR_SC_MODE=m)) for m in ('0', '1')]
        for which, env in runs:
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, XR_CHILD=which, **env), check=False)
