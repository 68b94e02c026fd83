"""Do the hash-grid kernels run slower inside the training loop because the optimiser's 439-MB stream has flushed the table /
workspace out of the L2s and the Infinity Cache?  Times xr_hashgrid_fwd and xr_hashgrid_bwd (a) back to back, (b) each call
behind a 600-MB streaming pass (events around every single call, median).  usage: python tools/microbench_cold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
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
enc = torch.empty((32, ld), device=dev); denc = torch.randn((32, ld), device=dev)
g = torch.zeros(meta.n_params, device=dev)
planes = c[:, :3].t().contiguous()
nsn = ns.cpu().numpy().astype(np.int64)
live = np.concatenate([np.arange(b, b + max(1, int(round(0.47 * k)))) for k, b in nsn if b + k <= n and k > 0]).astype(np.int32)
rows = torch.zeros(n, dtype=torch.int32, device=dev); rows[:len(live)] = t(live)
nl = torch.tensor([len(live), 0, 0, 0], dtype=torch.int32, device=dev)
big = torch.empty(150_000_000, device=dev)       # 600 MB


def med(f, thrash, reps=20):
    ts = []
    for _ in range(reps + 3):
        if thrash:
            big.add_(1.0)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts[3:]))


for name, f in (('fwd rows', lambda: ops.hashgrid_fwd(table, c[:, :3], meta, enc_t=enc, ld=ld)),
                ('fwd planes', lambda: ops.hashgrid_fwd(table, planes, meta, enc_t=enc, ld=ld)),
                ('bwd live overwrite', lambda: ops.hashgrid_bwd(c[:, :3], denc, meta, g, live=(rows, nl), overwrite=True))):
    print('%-20s hot %.1f us   behind a 600-MB stream %.1f us' % (name, med(f, False), med(f, True)), flush=True)
