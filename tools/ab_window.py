"""A/B of the window march's placement on one box: per-iteration device times (events on the compute stream) at the adaptive
fixed point for march_window = side / main / off, the in-place march with 8 lanes per ray or one -> normal / refresh / mean ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd import ops
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
variants = [v.split(':') for v in (sys.argv[1:] or ['side:1', 'side:0', 'main:1', 'off:1'])]
for rep in range(2):
    for mode, wide in variants:
        tr = Trainer(dev, n_img=int(os.environ.get('N_IMG', '100')), march_window=mode)
        tr.net.sampler.wide_in_place = wide == '1'
        tr.run(288)
        torch.cuda.synchronize()
        K = 64
        ev = [ops._CEvent() for _ in range(K + 1)]
        tr.run(K, iter_events=ev)
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
        ref = [m for i, m in enumerate(ms) if (288 + i) % 16 == 0]
        nor = [m for i, m in enumerate(ms) if (288 + i) % 16 != 0]
        first = [m for i, m in enumerate(ms) if (288 + i) % 16 == 1]
        print('march_window=%-4s wide=%s  normal %.4f  refresh %.4f  first-after-refresh %.4f  mean %.4f ms  rays %d' % (
            mode, wide, sum(nor) / len(nor), sum(ref) / len(ref), sum(first) / len(first), sum(ms) / K, tr.net.sampler.n_rays_per_batch), flush=True)
        del tr
