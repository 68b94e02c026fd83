"""config #3 training steps only (for rocprofv3 --kernel-trace --stats): python tools/profile_mip_step.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xrnerf_amd
from xrnerf_amd import mip
dev = torch.device('cuda:0')
cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'mip_model_cfg.json')))
R, S = cfg['N_rand_per_sampler'], cfg['num_samples']
torch.manual_seed(0)
net = xrnerf_amd.build_network(cfg['model']).to(dev)
from xrnerf_amd.train import FusedAdam
opt = FusedAdam(list(net.parameters()), lr=cfg['optimizer']['lr'], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_momentum=None)
rays = mip.synthetic_multiscale_rays(R, dev, seed=1)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    data = mip.get_z_vals(dict(rays), S + 1, randomized=True)
    out = net.train_step({k: v[None] for k, v in data.items()}, opt)
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()
    opt.step()
torch.cuda.synchronize()
print('loss', float(out['log_vars']['loss']))
