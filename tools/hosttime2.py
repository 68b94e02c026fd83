"""host enqueue time vs device time per training step (steady state): is the step host-bound?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xrnerf_amd.train import Trainer
dev = torch.device('cuda:0')
tr = Trainer(dev, n_img=int(os.environ.get("N_IMG", "100")))
for _ in range(320): tr.step()
torch.cuda.synchronize()
for rep in range(3):
    N = 15                      # between two refreshes
    while tr.iter % 16 != 1: tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get('HT_STEP') == '1':
        for _ in range(N): tr.step()        # one call per iteration (still the native loop unless XRNERF_TRAINER=native_loop=0)
    else:
        tr.run(N)                           # one native window
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host (incl. the window-end counter drain, which waits for the side stream) %.3f ms/step   total (host + drain) %.3f ms/step' % ((t1 - t0) * 1e3 / N, (t2 - t0) * 1e3 / N), flush=True)
    if tr._loop is not None and tr._loop.enqueued:
        print('   inside xr_ngp_loop_run: %.4f ms/iteration over %d iterations' % (tr._loop.enqueue_s * 1e3 / tr._loop.enqueued, tr._loop.enqueued), flush=True)
        tr._loop.enqueue_s, tr._loop.enqueued = 0.0, 0
if os.environ.get('HT_PROFILE') != '1':
    sys.exit(0)
import cProfile, pstats
while tr.iter % 16 != 1: tr.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(15): tr.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr).stats
rows = sorted(((tt, ct, nc, f) for f, (cc, nc, tt, ct, callers) in st.items()), reverse=True)
print('self us/step  cum us/step  calls/step  function')
for tt, ct, nc, f in rows[:40]:
    print('%10.1f  %10.1f  %8.1f   %s:%d %s' % (tt * 1e6 / 15, ct * 1e6 / 15, nc / 15.0, os.path.basename(f[0]), f[1], f[2]))
