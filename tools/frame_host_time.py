"""Where the registry frame's time goes (HashNerfNetwork.val_step, chunk = 4096, one launch per kernel): host stamps around the frame's
phases -- ray generation (pipeline), batchify_forward (enqueue incl. the sampler's one read-back), the wait for the device, the
device-to-host copy of the image -- against the device time of the same frame (events).  usage: python tools/frame_host_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=int(os.environ.get('N_IMG', '100')))
tr.run(320)
net, data = tr.net, tr.data
H = W = 800
focal = data.focal
rows = []
for f in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o, d = ops.gen_rays(data.poses[f % data.n_img], H, W, focal, focal, 0.5 * W, 0.5 * H, device=dev)
    frame = {'rays_o': o, 'rays_d': d, 'img_ids': torch.zeros((o.shape[0], 1), dtype=torch.float32, device=dev), 'src_shape': np.array([H, W, 3])}
    t1 = time.perf_counter()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    with torch.no_grad():
        ret = net._render_rows(frame)
    b.record()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rgb = ret['rgb'].reshape(H, W, 3).cpu().numpy()
    t4 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, a.elapsed_time(b)))
print('frame  rays-gen  render-enqueue(+read-back)  wait  image D2H  total | device ms between the events')
for i, r in enumerate(rows):
    print('%5d  %8.2f  %25.2f  %5.2f  %9.2f  %6.2f | %.2f' % ((i,) + r))
m = np.median(np.array(rows[2:]), 0)
print('median %7.2f  %25.2f  %5.2f  %9.2f  %6.2f | %.2f' % tuple(m))
