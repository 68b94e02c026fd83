import sys, time, torch
sys.path.insert(0, '/root/repo')
from xrnerf_amd import ops
from xrnerf_amd.train import Trainer, render_frame
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=20)
for _ in range(192): tr.step()
torch.cuda.synchronize()
for _ in range(2): render_frame(tr.net, tr.data.poses[0], 800, 800, tr.data.focal)
ops.TIMER = ops.KernelTimer()
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in range(5): rgb, a = render_frame(tr.net, tr.data.poses[f], 800, 800, tr.data.focal)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5 * 1e3
s = ops.TIMER.summary(); ops.TIMER = None
print('frame %.2f ms; samples/ray %.1f; alpha mean %.3f' % (el, tr.net.sampler.coords.shape[0] / 640000, float(a.mean())))
for k, v in sorted(s.items(), key=lambda kv: -kv[1][1]): print('  %-22s %.3f ms/frame' % (k, v[1] / 5))
from xrnerf_amd.train import render_frame_ert
for _ in range(2): render_frame_ert(tr.net, tr.data.poses[0], 800, 800, tr.data.focal)
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in range(5): rgb2, a2 = render_frame_ert(tr.net, tr.data.poses[f], 800, 800, tr.data.focal)
torch.cuda.synchronize(); print('ERT frame %.2f ms, evaluated %d of %d samples' % ((time.perf_counter()-t0)/5*1e3, *render_frame_ert.last_evaluated))
# how many samples could early termination skip?  fraction of samples after T < 1e-4
net = tr.net
from xrnerf_amd import ops as O2
o, d = O2.gen_rays(tr.data.poses[0], 800, 800, tr.data.focal, tr.data.focal, 400., 400., device=dev)
data = {'rays_o': o, 'rays_d': d, 'img_ids': torch.zeros((o.shape[0], 1), dtype=torch.int32, device=dev)}
with torch.no_grad():
    data = net.sampler.sample(data, net.mlp, True); data = net.mlp(data)
raw, coords, ns = data['raw'], net.sampler.coords, net.sampler.rays_numsteps
dt = coords[:, 3] * (1.73205080757/1024*128 - 1.73205080757/1024) + 1.73205080757/1024
alpha = 1 - torch.exp(-torch.exp(raw[:, 3]) * dt)
ray = torch.repeat_interleave(torch.arange(ns.shape[0], device=dev), ns[:, 0].long())
logT = torch.log1p(-alpha.clamp(max=1 - 1e-7))
cs = torch.cumsum(logT, 0); base = ns[:, 1].long(); start = torch.where(base > 0, cs[(base - 1).clamp(min=0)], torch.zeros_like(cs[:1]))
Tbefore = torch.exp(cs - logT - start[ray])
print('fraction of samples with T_before < 1e-4: %.3f; < 1e-2: %.3f' % (float((Tbefore < 1e-4).float().mean()), float((Tbefore < 1e-2).float().mean())))
