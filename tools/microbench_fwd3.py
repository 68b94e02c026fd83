"""Forward-gather micro-benchmark on marched Lego samples (ray-ordered, ~2.6e5 rows, and a 16x larger frame-sized launch): every level
alone (the per-level cost line the cost-balanced XCD map in xr_encode.hip is built from), then the whole lookup with 28-byte
coordinate rows and with positions as three planes, checked bit-exactly against the oracle.  XRNERF_LIB selects another build
(tools/build_variant.sh) for A/Bs.  usage: python tools/microbench_fwd3.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def setup():
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
    return np, torch, O, ops, S, dev, c, n, meta, table


def timeit(torch, f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3


def child(which):
    np, torch, O, ops, S, dev, c, n, meta, table = setup()
    ld = (n + 63) // 64 * 64
    enc = torch.empty((32, ld), device=dev)
    ndev = torch.tensor([n], dtype=torch.int32, device=dev)
    if which == 'costs':
        cs = [timeit(torch, lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=ld, n_dev=ndev, levels=(l, l + 1)), 20) for l in range(16)]
        print('COST ' + ','.join('%.1f' % v for v in cs), flush=True)
        return
    tag = 'lib=%s' % os.path.basename(os.environ.get('XRNERF_LIB', 'default'))
    us = timeit(torch, lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=ld, n_dev=ndev))
    soa = c[:, :3].t().contiguous()                              # planes [3, n]
    e2 = torch.empty_like(enc)
    us2 = timeit(torch, lambda: ops.hashgrid_fwd(table, soa, meta, enc_t=e2, ld=ld, n_dev=ndev))
    same = bool(torch.equal(enc[:, :n], e2[:, :n]))
    ref = O.hashgrid_fwd(S.hash_table(meta.n_params), c[:4096, :3].cpu().numpy(), O.GridMeta())
    exact = bool((enc[:, :4096].t().cpu().numpy() == ref).all())
    # frame-sized launch: the same rows 16 times over (coherent like a frame's rays; 4.1 M samples)
    big = c.repeat(16, 1)[:, :3].contiguous(); nb = big.shape[0]; ldb = (nb + 63) // 64 * 64
    encb = torch.empty((32, ldb), device=dev)
    usb = timeit(torch, lambda: ops.hashgrid_fwd(table, big, meta, enc_t=encb, ld=ldb), 5)
    bigs = big.t().contiguous()
    usbs = timeit(torch, lambda: ops.hashgrid_fwd(table, bigs, meta, enc_t=encb, ld=ldb), 5)
    f = lambda u, m: m * 1164 / u / 1e3 / 8000
    print('%-60s n=%d rows %.1f us (frac %.3f) planes %.1f us (frac %.3f) | n=%d rows %.0f us (%.3f) planes %.0f us (%.3f) | planes==rows %s, bit-exact vs oracle %s'
          % (tag, n, us, f(us, n), us2, f(us2, n), nb, usb, f(usb, nb), usbs, f(usbs, nb), same, exact), flush=True)


if __name__ == '__main__':
    if os.environ.get('XR_CHILD'):
        child(os.environ['XR_CHILD'])
    else:
        me = [sys.executable, os.path.abspath(__file__)]
        r = subprocess.run(me, env=dict(os.environ, XR_CHILD='costs'), capture_output=True, text=True)
        print(r.stdout, r.stderr[-400:])
        subprocess.run(me, env=dict(os.environ, XR_CHILD='run'), check=False)
