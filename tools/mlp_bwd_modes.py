"""fused MLP backward by arithmetic mode (XR_MLP_BWD_DW = f32 | b2 | b2x, one process per mode): time on 2^17 rows and the
deviation of every gradient from a float64 statement of the same network (numpy), as a fraction of the gradient's max"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from xrnerf_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
n = 1 << 17
enc_t = (rng.normal(0, 0.5, (32, n))).astype(np.float32)
dirs = rng.uniform(0, 1, (n, 3)).astype(np.float32)
wd = ((rng.uniform(size=3072) - 0.5) * 0.8).astype(np.float32); wc = ((rng.uniform(size=7168) - 0.5) * 0.6).astype(np.float32)
draw = rng.normal(0, 1e-2, (n, 4)).astype(np.float32)
t = lambda a: torch.from_numpy(a).to(dev)
enc_d, dirs_d, wd_d, wc_d, draw_d = t(enc_t), t(dirs), t(wd), t(wc), t(draw)
denc = torch.empty_like(enc_d); gwd = torch.zeros(3072, device=dev); gwc = torch.zeros(7168, device=dev)
ndev = torch.tensor([n], dtype=torch.int32, device=dev)
f = lambda: ops.nerf_mlp_bwd(enc_d, dirs_d, n, wd_d, wc_d, 1, 2, draw_d, gwd, gwc, denc_t=denc, n_dev=ndev)
gwd.zero_(); gwc.zero_(); f(); torch.cuda.synchronize()
got = [gwd.cpu().numpy().astype(np.float64), gwc.cpu().numpy().astype(np.float64), denc.cpu().numpy().astype(np.float64)]
# float64 statement of the same network with torch autograd (topology (1, 2); colour input = [density out 1..15, SH-4, pad = 1];
# the SH values come from the fp32 kernel and are a constant of the graph)
D = torch.float64
x = enc_d.t().to(D).requires_grad_(True)
W0 = wd_d[:2048].view(64, 32).to(D).requires_grad_(True); W1 = wd_d[2048:].view(16, 64).to(D).requires_grad_(True)
C0 = wc_d[:2048].view(64, 32).to(D).requires_grad_(True); C1 = wc_d[2048:6144].view(64, 64).to(D).requires_grad_(True)
C2 = wc_d[6144:].view(16, 64).to(D).requires_grad_(True)
dout = torch.relu(x @ W0.t()) @ W1.t()
cin = torch.cat([dout[:, 1:16], ops.sh4(dirs_d).to(D), torch.ones((n, 1), dtype=D, device=dev)], 1)
cout = torch.relu(torch.relu(cin @ C0.t()) @ C1.t()) @ C2.t()
raw = torch.cat([cout[:, :3], dout[:, :1]], 1)
(raw * draw_d.to(D)).sum().backward()
ref = [torch.cat([W0.grad.reshape(-1), W1.grad.reshape(-1)]).cpu().numpy(),
       torch.cat([C0.grad.reshape(-1), C1.grad.reshape(-1), C2.grad.reshape(-1)]).cpu().numpy(), x.grad.t().cpu().numpy()]
names = ['dW density', 'dW color', 'dL/d enc']
line = ['%s %.2e' % (nm, np.abs(g - r).max() / np.abs(r).max()) for nm, g, r in zip(names, got, ref)]
def timeit(reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
print('XR_MLP_BWD_DW=%-4s  %.1f us on %d rows   max |err| / max |grad|:  %s' % (os.environ.get('XR_MLP_BWD_DW', 'b2x'), timeit(), n, '   '.join(line)), flush=True)
# where does the largest dL/d enc deviation come from?  A hidden unit whose pre-activation is within rounding of 0 has its ReLU on in
# one arithmetic and off in the other: that row's gradient then differs by a whole weight column, whatever the precision of the products.
with torch.no_grad():
    z0 = x @ W0.t(); zc0 = cin @ C0.t(); zc1 = torch.relu(zc0) @ C1.t()
    zmin = torch.cat([z0.abs(), zc0.abs(), zc1.abs()], 1).min(1).values.cpu().numpy()        # per row: the pre-activation closest to its kink
err_row = np.abs(got[2] - ref[2]).max(0) / np.abs(ref[2]).max()
bad = np.nonzero(err_row > 1e-4)[0]
print('   dL/d enc: %d of %d rows deviate by more than 1e-4 of max; their closest |pre-activation| (float64): max %.2e (median over all rows %.2e); '
      'max deviation over the rows whose closest |pre-activation| exceeds 1e-5: %.2e of max'
      % (len(bad), n, zmin[bad].max() if len(bad) else 0.0, np.median(zmin), err_row[zmin > 1e-5].max()), flush=True)
