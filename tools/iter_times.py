import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=20)
PRE = int(os.environ.get('PREROLL', '64'))
for _ in range(PRE): tr.step()
evs = []
for i in range(128):
    a = torch.cuda.Event(enable_timing=True); a.record(); tr.step(); evs.append(a)
b = torch.cuda.Event(enable_timing=True); b.record(); torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(127)]
by = {}
for i, t in enumerate(ts): by.setdefault((PRE + i) % 16, []).append(t)
print('mean %.3f ms' % (sum(ts) / len(ts)))
for k in sorted(by): print('iter%%16=%2d  mean %.3f ms' % (k, sum(by[k]) / len(by[k])))
