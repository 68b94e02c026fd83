"""sanity: train on the synthetic Lego-shaped scene and report held-out-style PSNR of rendered frames"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from xrnerf_amd.train import Trainer, render_frame, _render_boxes
from xrnerf_amd import ops
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tr = Trainer(dev, n_img=30, H=R, W=R)
def psnr_of(k):
    rgb, alpha = render_frame(tr.net, tr.data.poses[k], R, R, tr.data.focal)
    o, d = ops.gen_rays(tr.data.poses[k], R, R, tr.data.focal, tr.data.focal, R / 2, R / 2, device=dev)
    gt = _render_boxes(o, d, tr.data.boxes.to(dev))
    pred = rgb.reshape(-1, 3)      # bg black: gt rgb already premultiplied by hit mask
    mse = ((pred - gt[:, :3]) ** 2).mean()
    return float(-10 * torch.log10(mse))
t0 = time.time()
for it in range(1, 3001):
    out = tr.step()
    if it in (100, 300, 1000, 2000, 3000):
        torch.cuda.synchronize()
        print('iter %5d  %.1f s  train-psnr %.2f  rays/batch %d  render PSNR %.2f dB' % (it, time.time() - t0, float(out['log_vars']['psnr']), tr.net.sampler.n_rays_per_batch, sum(psnr_of(k) for k in range(3)) / 3))
