"""Real-pixels sanity record (GPU box): trains configs/instant_ngp/nerf_blender_local01.py's model on the reference's own
5-image Blender-Lego fixture through datasets.HashNerfDataset (staged by tools/stage_ref_lego.py) and prints PSNR at
100 / 1000 / 5000 iterations: on the 4 training views (val + train images, the reference's image order; masked by alpha as
networks/hashnerf.py:83-90 does) and on the held-out test view.  Four views cannot reach the 35.1 dB the reference
publishes for the full 100-view scene (docs/en/benchmark.md:231-233) on novel views; the training-view number shows the
pipeline fits real images.   usage: python tools/train_real_lego.py [datadir] [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import datasets, synthetic
from xrnerf_amd.train import Trainer, render_frame

datadir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'oracle', '_ref', 'data', 'lego')
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device('cuda:0')
cfg = dict(datadir=datadir, half_res=False, testskip=1, white_bkgd=False, load_alpha=True, N_rand_per_sampler=4096, mode='train', val_n=2)
ds = datasets.HashNerfDataset(cfg, device=dev)
imgs, poses, _, hwf, i_split = datasets.load_blender_data(datadir, False, 1)
test_img = torch.from_numpy(imgs[i_split[2][0]]).to(dev)
test_pose = synthetic.poses_nerf2ngp(poses[i_split[2][0]][None])[0]
tr = Trainer(dev, dataset=ds)
H, W, focal = ds.H, ds.W, ds.focal


def psnr(rgb, rgba):
    a = rgba[..., 3:]
    mse = ((rgb * a - rgba[..., :3] * a) ** 2).mean()
    return float(-10 * torch.log10(mse))


rec = {'scene': 'reference test fixture nerf_synthetic/lego: %d training views (val + train), 1 held-out test view, %dx%d' % (ds.n_img, H, W), 'points': []}
t0 = time.time()
for it in range(1, 5001):
    out = tr.step()
    if it in (100, 1000, 5000):
        torch.cuda.synchronize()
        tv = [psnr(render_frame(tr.net, ds.poses[k], H, W, focal)[0], torch.from_numpy(ds.images[k]).to(dev)) for k in range(ds.n_img)]
        te = psnr(render_frame(tr.net, test_pose, H, W, focal)[0], test_img)
        p = {'iteration': it, 'seconds': time.time() - t0, 'batch_psnr': float(out['log_vars']['psnr']), 'rays_per_batch': int(tr.net.sampler.n_rays_per_batch),
             'train_view_psnr_mean': float(np.mean(tv)), 'train_view_psnr': tv, 'held_out_test_view_psnr': te}
        rec['points'].append(p)
        print(json.dumps(p), flush=True)
if out_path:
    json.dump(rec, open(out_path, 'w'), indent=1)
