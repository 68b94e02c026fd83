import sys, time, torch
sys.path.insert(0, '/root/repo')
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=20)
for _ in range(64): tr.step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(64):
    a = time.perf_counter(); tr.step(); ts.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
ts = sorted(ts)
print('host enqueue per step: median %.3f ms, min %.3f, p90 %.3f; loop %.3f ms/step; drain after loop %.3f ms' % (ts[32]*1e3, ts[0]*1e3, ts[57]*1e3, (t1-t0)/64*1e3, (t2-t1)*1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(32): tr.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
