for lib in ${MIP_AB_LIBS:-unset}; do
  if [ "$lib" = unset ]; then unset XRNERF_LIB; else export XRNERF_LIB=$lib; fi
  python - <<'PY'
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import xrnerf_amd
from xrnerf_amd import mip
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
dev = torch.device('cuda:0')
cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'mip_model_cfg.json')))
R, S = cfg['N_rand_per_sampler'], cfg['num_samples']
torch.manual_seed(0)
net = xrnerf_amd.build_network(cfg['model']).to(dev)
from xrnerf_amd.train import FusedAdam
opt = (FusedAdam(list(net.parameters()), lr=cfg['optimizer']['lr'], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_momentum=None)
       if os.environ.get('MIP_OPT', 'fused') == 'fused' else torch.optim.Adam(net.parameters(), lr=cfg['optimizer']['lr']))
rays = mip.synthetic_multiscale_rays(R, dev, seed=1)
def step():
    data = mip.get_z_vals(dict(rays), S + 1, randomized=True)
    out = net.train_step({k: v[None] for k, v in data.items()}, opt)
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()
    opt.step()
    return out
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): out = step()
torch.cuda.synchronize()
print(os.environ.get('XRNERF_LIB', 'default').split('_')[-1], os.environ.get('MIP_OPT', 'fused'), '%.3f ms/step' % ((time.perf_counter() - t0) * 1e3 / 20), 'loss %.6f' % float(out['log_vars']['loss']))
PY
done
