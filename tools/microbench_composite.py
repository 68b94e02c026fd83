"""xr_composite_train alone on the bench's steady-state batch: the captured call replayed as is, with every ray cut to at
most 16 / 64 samples, and with the ray list sorted by length (what bounds the launch: the longest rays or the launch itself)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['XRNERF_STEP'] = 'py'
import torch
from xrnerf_amd.train import Trainer
from xrnerf_amd import ops
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=20)
for _ in range(330): tr.step()
cap = {}
orig = ops.composite_train
def spy(*a, **k):
    cap['a'] = [x.clone() if torch.is_tensor(x) else x for x in a]; cap['k'] = dict(k)
    return orig(*a, **k)
ops.composite_train = spy
tr.step()
ops.composite_train = orig
torch.cuda.synchronize()
a = cap['a']
raw, coords, ns, nsc, bg, tgt, alpha, gmean, ra, da, loss_mse, draw = a[:12]
def timeit(nsc_, reps=50):
    f = lambda: orig(raw, coords, ns, nsc_, bg, tgt, alpha, gmean, ra, da, loss_mse, draw, **cap['k'])
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
c = nsc[:, 0].float()
qs = torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)
print('rays %d  samples per ray: mean %.1f  none %.2f  > 64: %.3f  quantiles 0.5/0.9/0.99/max %s' % (
    c.numel(), float(c.mean()), float((c == 0).float().mean()), float((c > 64).float().mean()), [int(v) for v in torch.quantile(c, qs)]))
print('as captured            %.1f us' % timeit(nsc))
for cut in (() if os.environ.get('MB_ONLY_CAPTURED') else (64, 16, 4, 0)):
    n2 = nsc.clone(); n2[:, 0] = torch.clamp(n2[:, 0], max=cut)
    print('rays cut to <= %-3d     %.1f us' % (cut, timeit(n2)))
