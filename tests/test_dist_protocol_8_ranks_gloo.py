"""The data-parallel protocol at the node's size -- EIGHT ranks over gloo on the host (no kernels run here: what is checked is the
arithmetic every rank must agree on before an 8-GPU node exists).  SURVEY.md section 8e; /root/reference/xrnerf/core/apis/train.py:28-38
is the reference's DDP wrapping this replaces.
  * image-space sharding: row bands of 800 rows (8 x 100) and of 756 rows (four bands of 95, four of 94 -- BASELINE config #4) tile
    the frame exactly, and dist.gather_image puts the ragged bands back in order on every rank;
  * zero1: the table (12 196 240 floats) padded to 8 shards of a multiple of 4 floats, every rank owning its slice of the SAME padded
    storage; reduce-scatter (over gloo: all-reduce + slice) then all-gather leaves every rank with every shard's update;
  * the native loop's exchange hooks (dist.CallbackExchange) on slices of registered buffers with 8 ranks: all-reduce of the fine /
    coarse halves, reduce-scatter into shard `rank`, all-gather from it;
  * the bucketed all-reduce (three buckets), the collective path decision of the trainer (Trainer._agreed_span's MIN over ranks);
  * dist.comm_model at 8 ranks (ring against direct exchange: the figures DESIGN.md section 6 quotes)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from xrnerf_amd import dist as xd
rank, local, world = xd.init_from_env('gloo')
assert world == 8
# ---- row bands (image-space ray sharding) and the tile gather
for H, W in ((800, 8), (756, 6), (13, 3)):
    bands = [xd.row_band(H, r, world) for r in range(world)]
    assert bands[0][0] == 0 and all(bands[r][0] + bands[r][1] == bands[r + 1][0] for r in range(world - 1)) and bands[-1][0] + bands[-1][1] == H
    assert max(b[1] for b in bands) - min(b[1] for b in bands) <= 1
    row0, nrows = bands[rank]
    img = (torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4) * 0.5)
    full = xd.gather_image(img[row0:row0 + nrows].clone(), H, rank, world)
    assert full.shape == img.shape and torch.equal(full, img)
assert [xd.row_band(756, r, 8)[1] for r in range(8)] == [95, 95, 95, 95, 94, 94, 94, 94]
assert [xd.row_band(800, r, 8)[1] for r in range(8)] == [100] * 8
# ---- zero1 at world 8: padding, shard ownership, reduce-scatter -> update -> all-gather
N = 12196240 // 64 + 3                                    # the table's size scaled down, NOT a multiple of 8 (nor of 4)
z = xd.Zero1GradSync(world, rank)
p = torch.nn.Parameter(torch.arange(N, dtype=torch.float32) * 1e-3)
sh = z.attach(p)
assert z.shard %% 4 == 0 and z.shard * world >= N and z.shard * world - N < 4 * world and sh.numel() == z.shard
assert p.data_ptr() == z.param_padded.data_ptr() and sh.data_ptr() == z.param_padded.data_ptr() + 4 * rank * z.shard
full = 12196240
assert (-(-full // 8) + 3) // 4 * 4 == 1524532 and 1524532 * 8 - full == 16          # the real table: 16 floats of padding over 8 ranks
padded, view = z.pad_grad(torch.device('cpu'))
view.copy_(torch.arange(N, dtype=torch.float32) * (rank + 1))                        # this rank's gradient
mlp_g = torch.full((640,), float(rank + 1))
z.ready(mlp_g); z.ready(view)
scale = z.finish()
assert scale == 1.0 / 8 and torch.equal(mlp_g, torch.full((640,), 36.0))
lo, hi = rank * z.shard, min((rank + 1) * z.shard, N)
want = torch.arange(N, dtype=torch.float32) * 36.0
assert torch.equal(z.shard_param.grad[:max(hi - lo, 0)], want[lo:hi]) and float(z.shard_param.grad[max(hi - lo, 0):].abs().sum()) == 0.0
with torch.no_grad():
    sh.add_(100.0 * (rank + 1))                                                      # "the optimiser" on this rank's shard
z.gather_params()
for r in range(world):
    a, b = r * z.shard, min((r + 1) * z.shard, N)
    assert torch.equal(p.data[a:b], torch.arange(N, dtype=torch.float32)[a:b] * 1e-3 + 100.0 * (r + 1)), r
# ---- the native loop's exchange hooks with 8 ranks (callback form), on slices of registered buffers
ex = xd.native_exchange(world, rank)
assert type(ex).__name__ == 'CallbackExchange' and ex.c.world_size == 8 and ex.c.rank == rank
shard = 250
grad = torch.arange(world * shard, dtype=torch.float32) * (rank + 1)
shard_grad, gathered = torch.zeros(shard), torch.zeros(world * shard)
ex.register(grad, shard_grad, gathered)
f = ex.c
cut = 1234                                                                           # fine half [cut, end), then coarse half [0, cut)
assert f.all_reduce(f.ctx, grad.data_ptr() + 4 * cut, world * shard - cut, None) == 0 and f.all_reduce(f.ctx, grad.data_ptr(), cut, None) == 0
assert f.finish(f.ctx, None) == 0
assert torch.equal(grad, torch.arange(world * shard, dtype=torch.float32) * 36.0)
g2 = torch.arange(world * shard, dtype=torch.float32) * (rank + 1)
ex.register(g2)
assert f.reduce_scatter(f.ctx, g2.data_ptr(), shard_grad.data_ptr(), shard, None) == 0 and f.finish(f.ctx, None) == 0
assert torch.equal(shard_grad, torch.arange(world * shard, dtype=torch.float32)[rank * shard:(rank + 1) * shard] * 36.0)
gathered[rank * shard:(rank + 1) * shard] = float(rank)
assert f.all_gather(f.ctx, gathered.data_ptr() + 4 * rank * shard, gathered.data_ptr(), shard, None) == 0 and f.finish(f.ctx, None) == 0
assert torch.equal(gathered, torch.arange(world, dtype=torch.float32).repeat_interleave(shard))
# ---- bucketed all-reduce: three buckets in flight, one finish
bs = xd.BucketedGradSync(world)
b1, b2, b3 = torch.full((10,), 1.0 + rank), torch.full((1 << 16,), 2.0), torch.full((1 << 15,), float(rank))
for b in (b1, b2, b3):
    bs.ready(b)
assert bs.finish() == 1.0 / 8
assert torch.equal(b1, torch.full((10,), 36.0)) and torch.equal(b2, torch.full((1 << 16,), 16.0)) and torch.equal(b3, torch.full((1 << 15,), 28.0))
# ---- the path decision of the training loop is a MIN over the ranks (Trainer._agreed_span): one rank without a native span decides
from xrnerf_amd.train import Trainer
class _S:                     # (only what _agreed_span reads)
    update_grid_freq = 16
class _N:
    sampler = _S()
t = Trainer.__new__(Trainer)
t._ctrl_group, t.iter, t.net = dist.group.WORLD, 5, _N()
assert t._agreed_span(0 if rank == 3 else 11) == 0
assert t._agreed_span(11 - rank) == 4
t.iter = 16
assert t._agreed_span(7) == 7                               # a refresh iteration: per-iteration on every rank by construction, no message
# ---- the communication model at the node's size
m = xd.comm_model(8, step_ms=0.40)
gb = 4.0 * 12196240 + 4.0 * 10240
assert abs(m['gradient_bytes_per_rank'] - gb) < 1 and abs(m['ring_ms'] - 2 * 7 / 8 * gb / 153e9 * 1e3) < 1e-9
assert abs(m['direct_ms'] - 2 * gb / 8 / 153e9 * 1e3) < 1e-9 and m['direct_ms'] < m['ring_ms'] / 6
dist.barrier(); dist.destroy_process_group()
print('ok')
'''


def test_protocol_at_eight_ranks(tmp_path):
    script = tmp_path / 'w8.py'
    script.write_text(WORKER % ROOT)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='8', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(8)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-1500:] for o in outs if 'ok' not in o]
    assert all('ok' in o for o in outs)
