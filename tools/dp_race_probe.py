"""Two ranks sharing cuda:0 (gloo), the worker of tests/test_gpu_dist.py run K times through the native loop and K times through the
per-iteration path: prints the distinct signatures each path produced (a data race shows as more than one) and, per checkpoint, which
parameter tensors differ.  usage: python tools/dp_race_probe.py [K] [dp_mode]"""
import os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else 'allreduce'
WORKER = r'''
import hashlib, os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from xrnerf_amd import dist as xd
from xrnerf_amd.train import Trainer
rank, local, world = xd.init_from_env('gloo')
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
native = os.environ.get('DP_NATIVE', '1') == '1'
kw = eval(os.environ.get('DP_KW', '{}'))
tr = Trainer(dev, n_img=3, H=128, W=128, world_size=world, rank=rank, ema=False, native_loop=native, **kw)
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:8]
snaps = []
levels = os.environ.get('DP_LEVELS', '0') == '1'
sync_at = eval(os.environ.get('DP_SYNC', '(0, 5, 15, 16, 17, 18, 19)'))
it = 0
while it < 20:
    if it == 9 and native and os.environ.get('DP_RUN4', '1') == '1':
        tr.run(4); it += 4; continue
    tr.step(); it += 1
    if levels:
        snaps.append([q.detach().clone() for q in tr.net.parameters() if q.numel() > 0])
    if it - 1 not in sync_at:
        continue
    it -= 1
    torch.cuda.synchronize()
    ps = [q for q in tr.net.parameters() if q.numel() > 0]
    print('SIG', rank, it, h(tr.net.sampler.density_grid_bitfield), ' '.join(h(q) for q in ps), flush=True)
    it += 1
if levels:
    torch.cuda.synchronize()
    off = [2 * int(o) for o in tr.net.mlp.embedder_pos.meta.offset]
    for i, ps in enumerate(snaps):
        tab = max(ps, key=lambda q: q.numel()).reshape(-1)
        print('SIG', rank, 'L', i, ' '.join(h(tab[off[l]:off[l + 1]]) for l in range(len(off) - 1)), '|', ' '.join(h(q) for q in ps if q.numel() < tab.numel()), flush=True)
dist.barrier(); dist.destroy_process_group()
''' % ROOT
d = tempfile.mkdtemp()
open(os.path.join(d, 'w.py'), 'w').write(WORKER)
seen = {}
for native in eval(os.environ.get('DP_PATHS', '(1, 0)')):
    for k in range(K):
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0', XRNERF_DP=MODE,
                   DP_NATIVE=str(native))
        procs = [subprocess.Popen([sys.executable, os.path.join(d, 'w.py')], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        if any(p.returncode for p in procs):
            print('run failed', native, k, outs[0][-800:]); continue
        sig = tuple(l for l in outs[0].splitlines() if l.startswith('SIG'))
        seen.setdefault(sig, []).append((native, k))
print('%d distinct signature lists over %d runs (%s)' % (len(seen), 2 * K, MODE))
ref = max(seen, key=lambda s: len(seen[s]))
for s, runs in seen.items():
    print('runs', runs, 'majority' if s is ref else 'DEVIATES')
    if s is not ref:
        for a, b in zip(ref, s):
            if a != b:
                print('   majority', a); print('   this    ', b)
