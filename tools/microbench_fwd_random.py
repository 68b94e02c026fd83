"""The occupancy-grid refresh's 2^20-point density query on UNIFORMLY RANDOM positions (K6's points: no ray coherence, every lane
its own cell at every level): per-level cost of xr_hashgrid_fwd2 on planes, then the whole launch with the training map's level
costs against a map built from the costs measured here (XR_HG_COST is read once per process: the script re-executes itself).
usage: python tools/microbench_fwd_random.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3


def child(which):
    import torch
    from xrnerf_amd import ops, synthetic as S
    dev = torch.device('cuda:0')
    n = 1 << 20
    g = torch.Generator(device='cpu'); g.manual_seed(5)
    planes = torch.rand((3, n), generator=g).to(dev).contiguous()
    meta = ops.GridMeta()
    table = torch.from_numpy(S.hash_table(meta.n_params)).to(dev)
    enc = torch.empty((32, n), device=dev)
    if which == 'costs':
        cs = [timeit(torch, lambda: ops.hashgrid_fwd(table, planes, meta, enc_t=enc, ld=n, levels=(l, l + 1)), 5) for l in range(16)]
        print('COST ' + ','.join('%.1f' % v for v in cs), flush=True)
        return
    us = timeit(torch, lambda: ops.hashgrid_fwd(table, planes, meta, enc_t=enc, ld=n))
    srt = planes[:, torch.argsort((planes[0] * 8).floor() + 8 * (planes[1] * 8).floor() + 64 * (planes[2] * 8).floor())].contiguous()
    us_s = timeit(torch, lambda: ops.hashgrid_fwd(table, srt, meta, enc_t=enc, ld=n))
    print('MODE=%s COST=%-50s random %.1f us (%.3f of HBM)   bucketed into 8^3 blocks %.1f us' % (
        os.environ.get('XR_HG_FWD_MODE'), (os.environ.get('XR_HG_COST') or 'default')[:50], us, n * 1164 / us / 1e3 / 8000, us_s), flush=True)


if __name__ == '__main__':
    if os.environ.get('XR_CHILD'):
        child(os.environ['XR_CHILD'])
    else:
        me = [sys.executable, os.path.abspath(__file__)]
        r = subprocess.run(me, env=dict(os.environ, XR_CHILD='costs'), capture_output=True, text=True)
        print(r.stdout, r.stderr[-400:])
        cost = [ln[5:] for ln in r.stdout.splitlines() if ln.startswith('COST ')]
        cost = cost[0] if cost else ''
        for env in (dict(), dict(XR_HG_COST=cost), dict(XR_HG_FWD_MODE='8'), dict(XR_HG_FWD_MODE='32', XR_HG_COST=cost)):
            subprocess.run(me, env=dict(os.environ, XR_CHILD='run', **env), check=False)
