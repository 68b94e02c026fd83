"""Data-parallel training path on the GPU box: two ranks (gloo transport, both on cuda:0 -- the driver's test box has one
GPU; RCCL over xGMI is the same code path with backend 'nccl') run the fused training step with the bucketed gradient
all-reduce for 18 iterations (grid refreshes at 0 and 16).  The replicas must stay identical -- parameters after every
Adam step, occupancy grid / bitfield after every refresh (the refresh is a deterministic function of the parameters, the
shared cameras and the explicit K6 RNG counters: no exchange) -- while marching DIFFERENT rays.  Also: `bench.py --gpus 2`
started by plain python (it spawns its own ranks)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from xrnerf_amd import dist as xd
from xrnerf_amd.train import Trainer
rank, local, world = xd.init_from_env('gloo')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
native = os.environ.get('DP_NATIVE', '1') == '1'
tr = Trainer(dev, n_img=3, H=128, W=128, world_size=world, rank=rank, ema=False, native_loop=native)
assert tr.net._fused_ok() and tr.net.grad_sync is not None
sig = []
bufs_seen, mem = set(), []
render_mid = os.environ.get('DP_RENDER_MID') == '1'
for it in range(18):
    if render_mid and it == 6 and rank == 0:
        # a frame on ONE rank in the middle of a window: its sampler takes back the marches issued ahead, so this rank has no native
        # span for the next iteration -- the path choice is collective (Trainer._agreed_span), the other rank follows it
        from xrnerf_amd.train import render_frame
        render_frame(tr.net, tr.data.poses[0], 32, 32, tr.data.focal * 32.0 / 128.0)
    if it == 9 and native:
        tr.run(4); continue_from = 13            # a window of several iterations in one native call
    elif it in (10, 11, 12) and native:
        continue
    else:
        tr.step()
    bufs_seen |= {id(b) for b in tr.net._step_bufs}
    if it in (5, 17):
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated())
    if os.environ.get('XRNERF_DP') == 'zero1':
        assert tr.net.mlp.embedder_pos.params.grad is None           # the shard's .grad is the optimiser's, the full table has none
    if it in (0, 15, 16, 17):
        torch.cuda.synchronize()
        p = torch.cat([q.detach().reshape(-1)[:4096] for q in tr.net.parameters() if q.numel() > 0])
        sig.append((hashlib.sha1(tr.net.sampler.density_grid_bitfield.cpu().numpy().tobytes()).hexdigest(),
                    hashlib.sha1(tr.net.sampler.density_grid.cpu().numpy().tobytes()).hexdigest(),
                    hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest(),
                    float(tr.data.rays_rgb[:64].sum())))
# the step alternates between its two buffer sets and allocates nothing per iteration (a gradient left on a parameter no optimiser
# clears used to pin a set per step: 48.8 MB of growth per iteration under zero1)
assert len(bufs_seen) == 2, len(bufs_seen)
assert mem[1] - mem[0] < (8 << 20) or (render_mid and rank == 0), mem          # (the frame's own buffers on the rank that rendered one)
assert tr.iter == 18
if native:
    # the iterations between the refreshes went through xr_ngp_loop_run with the exchange hooks (callbacks into torch.distributed here)
    assert tr._loop is not None and tr._loop.exchange is not None
    assert (1 <= tr._loop.enqueued < 16) if render_mid else tr._loop.enqueued == 16, tr._loop.enqueued
else:
    assert tr._loop is None
print('SIG', rank, ' '.join(a + b + c for a, b, c, _ in sig), flush=True)
objs = [None] * world
dist.all_gather_object(objs, sig)
if rank == 0:
    a, b = objs
    for (bf0, g0, p0, r0), (bf1, g1, p1, r1) in zip(a, b):
        assert bf0 == bf1, 'bitfields differ across ranks'
        assert g0 == g1, 'density grids differ across ranks'
        assert p0 == p1, 'parameters differ across ranks'
        assert r0 != r1, 'both ranks march the same rays'
    print('replicas identical over', len(a), 'checkpoints')
dist.barrier(); dist.destroy_process_group()
'''


def _two_ranks(tmp_path, dp_mode, native, render_mid=False):
    import socket
    script = tmp_path / ('w%d.py' % native)
    script.write_text(WORKER % ROOT)
    outs = None
    for attempt in range(2):          # (one retry: the rendezvous port is picked by bind-and-release, which another process can win)
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0', XRNERF_DP=dp_mode,
                   DP_NATIVE='1' if native else '0', DP_RENDER_MID='1' if render_mid else '0')
        # each rank on its own half of the chip's 256 compute units: two processes whose waves share compute units do not give a
        # bit-for-bit repeatable scatter (profiles/r06_two_processes_one_gpu_scatter_probe.txt); one process per GPU -- the product -- does
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), HSA_CU_MASK=('0:0-127', '0:128-255')[r]),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        if all(p.returncode == 0 for p in procs):
            break
        print('attempt %d failed:\n%s' % (attempt, '\n'.join(o[-1500:] for o in outs)))
    assert all(p.returncode == 0 for p in procs), [o[-1500:] for o in outs]
    assert 'replicas identical' in outs[0]
    return [[l for l in o.splitlines() if l.startswith('SIG')][-1].split()[2:] for o in outs]


@pytest.mark.parametrize('dp_mode', ['allreduce', 'zero1'])
def test_two_ranks_stay_identical_replicas(tmp_path, dp_mode):
    """dp_mode zero1: reduce-scatter -> Adam on the rank's shard -> all-gather (dist.Zero1GradSync; over gloo the reduce-scatter
    is an all-reduce + slice: the protocol, the padded storage and the sharded optimiser are what runs here).  Twice: through the
    NATIVE loop (xr_ngp_loop_run with the gradient-exchange hooks: the buckets go to the collectives from C++, one optimiser launch
    on the summed gradients per iteration) and through the per-iteration path -- same grids, bitfields and parameters at every
    checkpoint, bit for bit.
    Bit for bit ACROSS two jobs needs a bit-for-bit repeatable step.  Two processes whose waves SHARE compute units are not quite that
    (about one scatter launch in 150-600 writes the feature-0 value of ~16 items of one level wrong -- never when the process owns
    its compute units: profiles/r06_two_processes_one_gpu_scatter_probe.txt), so the two ranks of this stand-in for two GPUs get
    disjoint halves of the chip (HSA_CU_MASK per rank, `_two_ranks`): 0 events in 2 400 launches beside two training processes in that
    configuration.  One attempt."""
    nat = _two_ranks(tmp_path, dp_mode, True)
    per = _two_ranks(tmp_path, dp_mode, False)
    assert nat[0] == per[0] and nat[1] == per[1]


def test_a_frame_on_one_rank_in_the_middle_of_a_window_keeps_both_ranks_on_one_path(tmp_path):
    """rank 0 renders a frame between iterations 5 and 6: its marches are rewound, it cannot run iteration 6 natively -- and rank 1,
    which could, must not (the native loop and the per-iteration path exchange gradients through different communicators).  The span
    is agreed (Trainer._agreed_span): both take the per-iteration path for that iteration, both return to the native loop, and the
    replicas are identical at every checkpoint (the worker's own assertions)."""
    _two_ranks(tmp_path, 'allreduce', True, render_mid=True)


NCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from xrnerf_amd import dist as xd
rank, local, world = xd.init_from_env('nccl')          # world_size 1: RCCL is loaded and initialised, the collectives are self-copies
if not dist.is_initialized():
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
dev = torch.device('cuda', 0)
a = torch.arange(1024, dtype=torch.float32, device=dev)
dist.all_reduce(a)
out = torch.empty(1024, dtype=torch.float32, device=dev)
dist.reduce_scatter_tensor(out, a.clone())
g = torch.empty(1024, dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(g, out)
torch.cuda.synchronize()
assert torch.equal(g, torch.arange(1024, dtype=torch.float32, device=dev))
z = xd.Zero1GradSync(1, 0)
p = torch.nn.Parameter(torch.ones(1001, device=dev))
sh = z.attach(p)
assert sh.numel() == 1004 and p.numel() == 1001 and p.data_ptr() == sh.data_ptr()
print('rccl ok', dist.get_backend())
dist.destroy_process_group()
'''


def test_rccl_backend_initialises_and_runs_the_three_collectives(tmp_path):
    """backend 'nccl' (= RCCL on ROCm) with world_size 1 on the test box's one GPU: library load, communicator init, and the three
    collectives the data-parallel paths use (all-reduce, reduce-scatter, all-gather) -- everything short of a second GPU"""
    import socket
    script = tmp_path / 'n.py'
    script.write_text(NCCL_WORKER % ROOT)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b'rccl ok nccl' in r.stdout, r.stdout.decode()[-2000:]


RCCL_NATIVE = r'''
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from xrnerf_amd import _lib, dist as xd
torch.cuda.set_device(0)
ex = xd.RcclExchange(1, 0)                       # librccl dlopen'ed, a one-rank communicator on cuda:0, its stream and events
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
a = torch.arange(4096, dtype=torch.float32, device='cuda')
ref = a.clone()
assert ex.c.all_reduce(ex.c.ctx, a.data_ptr(), a.numel(), st) == 0
out = torch.zeros(4096, dtype=torch.float32, device='cuda')
assert ex.c.reduce_scatter(ex.c.ctx, a.data_ptr(), out.data_ptr(), 4096, st) == 0
assert ex.c.finish(ex.c.ctx, st) == 0
g = torch.zeros(4096, dtype=torch.float32, device='cuda')
assert ex.c.all_gather(ex.c.ctx, out.data_ptr(), g.data_ptr(), 4096, st) == 0
assert ex.c.finish(ex.c.ctx, st) == 0
torch.cuda.synchronize()
assert torch.equal(a, ref) and torch.equal(out, ref) and torch.equal(g, ref)
del ex
print('native rccl ok')
'''


def test_rccl_driven_from_native_code_at_world_size_one(tmp_path):
    """csrc/xr_dist.hip: librccl dlopen'ed, a communicator from a unique id, the three collectives of the data-parallel loop enqueued
    from native code on the communicator's own stream, `finish` ordering the caller's stream behind them -- everything short of a
    second GPU (what `bench.py --gpus N` uses under backend nccl)"""
    script = tmp_path / 'r.py'
    script.write_text(RCCL_NATIVE % ROOT)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b'native rccl ok' in r.stdout, r.stdout.decode()[-2000:]


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (here: gloo transport, both ranks on the one GPU of the test box)"""
    env = dict(os.environ, XRNERF_DIST_BACKEND='gloo', XRNERF_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '16', '--warmup', '0', '--n-img', '2',
                        '--no-preroll', '--no-cpu-baseline', '--no-mip', '--no-kilo', '--no-unbounded', '--no-f16', '--no-strict', '--no-extra'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['value'] > 0 and 'roofline' in d and d['scaling'] == 'weak'
