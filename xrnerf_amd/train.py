"""Host-side driver pieces around the network for benchmarking and smoke tests: the fused optimiser,
a device-resident synthetic "Lego-shaped" dataset, the per-iteration hook semantics of the reference
(`PassSamplerIterHook`, `ModifyBatchsizeHook`, `PassDatasetHook`;
/root/reference/xrnerf/core/hooks/hash_hook.py:12-42) and frame rendering.

This is NOT a re-implementation of the reference's mmcv runner / dataset stack (out of scope,
SURVEY.md section 2a rows 10-16): only what a bench needs to drive `HashNerfNetwork.train_step`.
"""
import math

import numpy as np
import os

import ctypes as C
import time

import torch

from . import _lib, ops, switches, synthetic
from .builder import build_network
from .datasets import DeviceRayTable


def ngp_lego_model_cfg(n_rays=4096):
    """The `model` dict of /root/reference/configs/instant_ngp/nerf_blender_local01.py:78-138,
    restated (the reference tree does not exist on the GPU box; tests/test_capi_and_host.py::test_reference_config_builds_unchanged checks this
    against the real file when it is present and against tests/golden/ngp_model_cfg.json)."""
    return dict(
        type='HashNerfNetwork',
        cfg=dict(phase='train', chunk=4096, bs_data='rays_o'),
        mlp=dict(
            type='HashNerfMLP', bound=1,
            embedder_pos=dict(n_input_dims=3, encoding_config=dict(
                otype='HashGrid', n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                interpolation='Linear')),
            embedder_dir=dict(n_input_dims=3, encoding_config=dict(otype='SphericalHarmonics', degree=4)),
            density_net=dict(n_input_dims=32, n_output_dims=16, network_config=dict(
                otype='FullyFusedMLP', activation='ReLU', output_activation='None', n_neurons=64, num_layers=1)),
            color_net=dict(n_output_dims=3, network_config=dict(
                otype='FullyFusedMLP', activation='ReLU', output_activation='None', n_neurons=64, num_layers=2))),
        sampler=dict(type='NGPGridSampler', update_grid_freq=16, update_block_size=5000000, n_rays_per_batch=n_rays,
                     cone_angle_constant=0.00390625, near_distance=0.2, target_batch_size=1 << 18, rgb_activation=2,
                     density_activation=3),
        render=dict(type='HashNerfRender', bg_color=[0, 0, 0]),
    )


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient) + the EMA copy that
    mmcv's EMAHook keeps (configs/instant_ngp/nerf_blender_local01.py:14-24), in ONE pass over
    p, g, m, v(, ema) per tensor (xr_adam_step_multi)."""

    def __init__(self, params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6, ema_momentum=None, ema_warm_up=100):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      ema_momentum=ema_momentum, ema_warm_up=ema_warm_up))

    def _state(self, p, group):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['m'] = torch.zeros_like(p)
            st['v'] = torch.zeros_like(p)
            if group['ema_momentum'] is not None:
                st['ema'] = p.detach().clone()
        return st

    @staticmethod
    def _ema_momentum(group, step):
        # mmcv's EMAHook warms its momentum up: min(momentum, (1 + iter) / (warm_up + iter)), iter = 0, 1, ...
        it = step - 1
        return min(group['ema_momentum'], (1.0 + it) / (group['ema_warm_up'] + it))

    @torch.no_grad()
    def fused_update(self, p):
        """Hands THIS step's update of `p` to the native training step, which applies it where the gradient is complete (the
        table: inside the scatter; the MLP tensors: behind the reduction of their gradients -- ops.ngp_train_step(adam=,
        mlp_adam=)): -> ops.adam_fuse with the state tensors and this update's constants, or None when `p` is
        not one of this optimiser's parameters.  The step counter advances here; the following step() finds no .grad on `p` and
        leaves it alone."""
        for group in self.param_groups:
            if any(q is p for q in group['params']):
                st = self._state(p, group)
                st['step'] += 1
                ema = st.get('ema') if group['ema_momentum'] is not None else None
                mom = self._ema_momentum(group, st['step']) if ema is not None else 0.0
                return ops.adam_fuse(p.data, st['m'], st['v'], ema, st['step'], group['lr'], group['betas'][0], group['betas'][1],
                                     group['eps'], group['weight_decay'], mom, 1.0)
        return None

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """`grad_scale`: factor on every gradient as the update reads it (the data-parallel trainer passes the 1/world_size
        its fused step left out: same update as scaling the gradients first, one 98-MB pass over them less)"""
        for group in self.param_groups:
            todo = []
            for p in group['params']:
                if p.grad is None or p.numel() == 0:
                    continue
                st = self._state(p, group)
                st['step'] += 1
                todo.append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st))
            # tensors that share a step count go out in one launch (up to 4 per launch)
            while todo:
                chunk = [t for t in todo[:4] if t[2]['step'] == todo[0][2]['step']]
                todo = [t for t in todo if all(t is not c for c in chunk)]
                emas = [c[2].get('ema') for c in chunk] if group['ema_momentum'] is not None else None
                mom = self._ema_momentum(group, chunk[0][2]['step']) if emas else 0.0
                ops.adam_step_multi([c[0] for c in chunk], [c[1] for c in chunk], [c[2]['m'] for c in chunk],
                                    [c[2]['v'] for c in chunk], chunk[0][2]['step'], group['lr'], group['betas'][0],
                                    group['betas'][1], group['eps'], group['weight_decay'], emas, mom, grad_scale=grad_scale)


def step_lr(base_lr, it, step=10000, gamma=0.2):
    """lr_config = dict(policy='step', step=10000, gamma=0.2) (nerf_blender_local01.py:22)"""
    return base_lr * gamma ** (it // step)


class SyntheticLego(DeviceRayTable):
    """Device-resident stand-in for HashNerfDataset (hashnerf_dataset.py:26-73): cameras on the
    Blender hemisphere in NGP space, all training rays [N*H*W, 11] = (o3, d3, rgba4, img_id)
    precomputed ON THE DEVICE with xr_gen_rays, targets rendered analytically from a union of
    axis-aligned boxes ("Lego-shaped", ~6 % of the level-0 cells), pre-shuffled once like the
    reference's np.random.shuffle.  (Real Blender scenes: datasets.HashNerfDataset, same interface.)"""

    def __init__(self, device, n_img=20, H=800, W=800, seed=1, shuffle=True, shuffle_seed=None):
        """`seed` fixes the CAMERAS (identical on every data-parallel rank: K7's visibility mask, hence the density
        grid and the bitfield, must not depend on the rank); `shuffle_seed` the ray order (per rank)."""
        self.boxes = _lego_boxes()
        boxes = self.boxes.to(device)
        super().__init__(device, synthetic.lego_cameras(n_img, seed=seed), lambda k, o, d: _render_boxes(o, d, boxes),
                         H, W, float(synthetic.LEGO_FOCAL) * W / 800.0, seed=seed if shuffle_seed is None else shuffle_seed,
                         shuffle=shuffle)


class SyntheticFern(DeviceRayTable):
    """BASELINE config #4 stand-in: an UNBOUNDED forward-facing scene (nerf_llff_data/fern-shaped: 20 cameras on a small
    patch all looking the same way, 1008 x 756, focal 815 px), aabb_scale = 16 -> five occupancy cascades are active
    (ngp_grid_sampler.py:83-85; ray_sampler_header.h:37-54 picks the cascade from the step size and the position).  Near
    geometry sits inside the unit cube around (0.5, 0.5, 0.5), far geometry up to 7 units behind it.  The reference has
    no such config (configs/instant_ngp/ only holds the bounded Blender one; HashNerfDataset hard-codes aabb_scale = 1):
    same model, sampler and kernels, only `aabb_scale` differs."""

    def __init__(self, device, n_img=20, H=756, W=1008, seed=6, shuffle=True, shuffle_seed=None):
        rng = np.random.default_rng(seed)
        focal = 815.0 * W / 1008.0
        poses44 = []
        for _ in range(n_img):
            c2w = np.eye(4)                                      # NeRF convention: the camera looks down its -z axis
            c2w[:3, 3] = [rng.uniform(-0.9, 0.9), rng.uniform(-0.6, 0.6), 3.6 + rng.uniform(-0.2, 0.2)]
            a, b = rng.uniform(-0.05, 0.05, 2)                   # small pan / tilt
            ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
            c2w[:3, :3] = ry @ rx
            poses44.append(c2w)
        poses = synthetic.poses_nerf2ngp(np.stack(poses44))
        # geometry along the common viewing direction, in NGP space: the centre ray of camera 0
        o0, d0 = synthetic.camera_rays(poses[0], H, W, focal, np.array([(H // 2) * W + W // 2]))
        fwd = d0[0] / np.linalg.norm(d0[0])
        centre = np.array([0.5, 0.5, 0.5])
        lo, hi = [], []
        for k in range(36):
            depth = [0.0, 0.1, 0.25, 0.6, 1.2, 2.0, 3.2, 4.8, 6.5][k % 9] + rng.uniform(-0.05, 0.05)
            size = 0.06 + 0.09 * depth                           # farther boxes are bigger (similar angular size)
            lateral = rng.uniform(-0.45, 0.45, 3) * (1.0 + 1.1 * depth)
            lateral -= fwd * lateral.dot(fwd)
            c = centre + fwd * depth + lateral
            h = size * rng.uniform(0.5, 1.0, 3)
            lo.append(c - h); hi.append(c + h)
        self.boxes = torch.tensor(np.stack([np.array(lo), np.array(hi)], 1), dtype=torch.float32)
        boxes = self.boxes.to(device)
        super().__init__(device, poses, lambda k, o, d: _render_boxes(o, d, boxes), H, W, focal,
                         seed=seed if shuffle_seed is None else shuffle_seed, shuffle=shuffle, aabb_scale=16)


def _lego_boxes(seed=2, fill=0.07, n_boxes=40):
    """the same boxes synthetic.lego_density_grid rasterises -> [B,2,3] (lo, hi)"""
    rng = np.random.default_rng(seed)
    lo, hi = [np.array([0.30, 0.38, 0.33])], [np.array([0.72, 0.62, 0.47])]
    vol = np.prod(hi[0] - lo[0])
    while vol < fill and len(lo) < n_boxes:
        c = rng.uniform(0.28, 0.72, 3)
        h = rng.uniform(0.02, 0.09, 3)
        lo.append(c - h); hi.append(c + h)
        vol += np.prod(2 * h) * 0.6
    return torch.tensor(np.stack([np.array(lo), np.array(hi)], 1), dtype=torch.float32)


def _render_boxes(o, d, boxes, chunk=1 << 20):
    """analytic RGBA of opaque coloured boxes: nearest slab hit, colour from the hit position."""
    out = []
    for s in range(0, o.shape[0], chunk):
        oo, dd = o[s:s + chunk, None, :], d[s:s + chunk, None, :]
        inv = 1.0 / torch.where(dd.abs() < 1e-9, torch.full_like(dd, 1e-9), dd)
        t0 = (boxes[None, :, 0, :] - oo) * inv
        t1 = (boxes[None, :, 1, :] - oo) * inv
        tn = torch.minimum(t0, t1).amax(-1)
        tf = torch.maximum(t0, t1).amin(-1)
        hit = (tf >= tn) & (tf > 0)
        tn = torch.where(hit, tn.clamp_min(0), torch.full_like(tn, 1e9))
        t, _ = tn.min(1)
        any_hit = t < 1e8
        p = o[s:s + chunk] + t[:, None].clamp(max=10) * d[s:s + chunk]
        rgb = (0.5 + 0.5 * torch.sin(p * 37.0)) * any_hit[:, None]
        out.append(torch.cat([rgb, any_hit[:, None].float()], 1))
    return torch.cat(out, 0)


class Trainer:
    """One process = one GPU.  Per iteration: hooks (set_iter, batch size) -> batch -> train_step ->
    backward -> [gradient all-reduce] -> fused Adam(+EMA)."""

    def __init__(self, device, n_img=20, H=800, W=800, seed=0, world_size=1, rank=0, dataset=None, ema=True, native_loop=True,
                 fuse_adam=True, direct_step=True, march_window='side', prefetch_k6=True):
        """The keyword switches (each overridable from the environment: XRNERF_TRAINER="fuse_adam=0,..."; xrnerf_amd/switches.py):
        native_loop     the iterations between two grid refreshes as native calls (xr_ngp_loop_run); False: one Python-driven step each
        fuse_adam       one GPU: the table scatter applies this optimiser's update itself (False: scatter, then the optimiser's launches)
        direct_step     one GPU: the fused step without an autograd graph (False: loss.backward() + optimizer.step())
        march_window    'side' (default): right behind a grid refresh the batches of the iterations up to the next refresh are drawn and
                        marched as ONE series of launches (the bitfield and the batch size do not change in between, K1 reads no
                        weights) on a side stream, beside the refresh iteration's own step -- the other iterations then run on one
                        in-order stream with nothing beside them; 'main': the same series on the compute stream; 'off': every
                        iteration marches in place
        prefetch_k6     the refresh's sample generation one iteration early, on the side stream"""
        opts = dict(native_loop=native_loop, fuse_adam=fuse_adam, direct_step=direct_step, march_window=march_window, prefetch_k6=prefetch_k6)
        opts.update(switches.trainer_overrides())
        if opts['march_window'] not in ('side', 'main', 'off'):
            raise ValueError("march_window is 'side', 'main' or 'off'")
        torch.manual_seed(seed)                       # identical initial weights on every rank
        self.device = device
        self.net = build_network(ngp_lego_model_cfg()).to(device)
        self.data = dataset or SyntheticLego(device, n_img, H, W, seed=1, shuffle_seed=1 + rank)
        self.net.sampler.set_data(self.data.get_alldata(), self.data.get_info())     # PassDatasetHook
        self.net.sampler.on_sampled = self._on_sampled
        self.net.sampler.on_rewind = self._on_rewind
        self.net.sampler.on_refreshed = self._march_ahead
        self.base_lr = 1e-2
        self.iter = 0
        self.world_size, self.rank = world_size, rank
        self.dp_mode = os.environ.get('XRNERF_DP', 'allreduce')
        if self.dp_mode not in ('allreduce', 'allreduce_bf16', 'zero1'):
            raise ValueError("XRNERF_DP must be 'allreduce', 'allreduce_bf16' or 'zero1' (got %r)" % self.dp_mode)
        opt_params = [p for p in self.net.parameters() if p.numel() > 0]
        if world_size > 1 and self.dp_mode == 'zero1':
            # reduce-scatter -> Adam on this rank's shard of the table -> all-gather
            from . import dist as xdist
            self.net.grad_sync = xdist.Zero1GradSync(world_size, rank)
            table = self.net.mlp.embedder_pos.params
            shard = self.net.grad_sync.attach(table)
            opt_params = [shard] + [p for p in opt_params if p is not table]
        self.opt = FusedAdam(opt_params, lr=self.base_lr, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6,
                             ema_momentum=0.05 if ema else None)
        if world_size > 1 and self.dp_mode in ('allreduce', 'allreduce_bf16'):
            from . import dist as xdist                 # fused step: bucketed reduction under the table scatter
            self.net.grad_sync = xdist.BucketedGradSync(world_size, torch.bfloat16 if self.dp_mode == 'allreduce_bf16' else None)
        if world_size > 1:
            # this trainer's optimiser applies the 1/world_size itself (FusedAdam.step(grad_scale=...)): the fused step leaves
            # the SUMMED gradients in .grad and reports the factor in net._pending_grad_scale
            self.net._defer_grad_scale = True
        # one GPU: the table scatter applies this optimiser's update to the table itself.  Valid because this trainer back-propagates
        # a unit root gradient and clears the gradients every step.  (The switch is raised around this trainer's own train_step
        # calls only: anyone else calling net.train_step gets gradients.)
        self.fuse_adam = world_size == 1 and opts['fuse_adam']
        self.direct_step = world_size == 1 and opts['direct_step']
        self._opt_params = opt_params
        self.rays_done = 0
        self.lazy_log = True
        self.march_window = opts['march_window']
        self.prefetch_k6 = opts['prefetch_k6']      # the refresh's K6 one iteration early, on the side stream
        # the iterations between two grid refreshes as native calls (xr_ngp_loop_run: the steps of the marched window with the updates
        # inside, enqueued from C++ -- the interpreter had been pacing the loop at 0.36 ms of host work per 0.42-ms iteration).
        # Needs what the fast Python path needs (one GPU, fused updates, the direct step, a marched window, a device-resident
        # ray table); anything else keeps the per-iteration path.
        self.native_loop = opts['native_loop']
        self._loop = None
        # data parallel: which path an iteration takes is a COLLECTIVE decision (_agreed_span): the native loop exchanges gradients through
        # its own communicator, the per-iteration path through torch.distributed's -- a rank that took the other path than its peers (a
        # frame rendered on rank 0 only rewinds that rank's marches; an error fallback) would pair its collectives with the wrong
        # ones.  The agreement runs over a host-side (gloo) group so that it never drains the GPU.
        self._ctrl_group = None
        if world_size > 1:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                self._ctrl_group = tdist.group.WORLD if tdist.get_backend() == 'gloo' else tdist.new_group(backend='gloo')
        self._queue = []               # [(iteration, the sampler's queued entry)] marched ahead, in order (entry.batch(): its batch dict)
        self._one = None

    def step(self):
        if self._agreed_span(self._native_span(1)):
            return self._run_native(1)
        return self._step_py()

    def _agreed_span(self, n):
        """data parallel: the smallest span any rank can run natively (0: every rank takes the per-iteration path for this iteration).
        Refresh iterations are per-iteration on every rank by construction (no message); everything else is agreed with one 8-byte
        host-side all-reduce -- once per refresh window in the steady state."""
        if self._ctrl_group is None or self.iter % self.net.sampler.update_grid_freq == 0:
            return n
        import torch.distributed as tdist
        t = torch.tensor([int(n)], dtype=torch.int64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MIN, group=self._ctrl_group)
        return int(t[0])

    def run(self, k, iter_events=None):
        """k iterations.  `iter_events` (k + 1 ops._CEvent timing events, optional): recorded on the compute stream in front of every
        iteration and behind the last (bench.py's per-iteration device times)."""
        out, done = None, 0
        while done < k:
            n = self._agreed_span(self._native_span(k - done))
            if n:
                out = self._run_native(n, iter_events[done:done + n + 1] if iter_events is not None else None)
            else:
                n = 1
                if iter_events is not None:
                    ops.record_event(iter_events[done])
                out = self._step_py()
                if iter_events is not None:
                    ops.record_event(iter_events[done + 1])
            done += n
        return out

    def _native_ok(self):
        st = getattr(self, '_native_static', None)
        if st is None:
            from .mlps import HashNerfMLP
            from .renders import HashNerfRender
            from .samplers import NGPGridSampler
            net = self.net
            one_gpu = self.world_size == 1 and self.fuse_adam and self.direct_step
            # data parallel: the loop hands the gradient buckets to an exchange (dist.native_exchange) and runs ONE optimiser launch on the
            # summed gradients per iteration -- fp32 on the wire (allreduce, zero1); the bf16 wire format keeps the per-iteration path
            dp = self.world_size > 1 and self.dp_mode in ('allreduce', 'zero1') and getattr(net, 'grad_sync', None) is not None
            st = self._native_static = bool(
                self.native_loop and (one_gpu or dp) and self.march_window != 'off' and
                self.device.type == 'cuda' and self._window_ok() and type(net.sampler) is NGPGridSampler and type(net.mlp) is HashNerfMLP and
                type(net.render) is HashNerfRender and (net.mlp.density_net.n_hidden, net.mlp.color_net.n_hidden) in ops._FUSED_BWD and
                isinstance(self.opt, FusedAdam))
        if not st or (self.world_size == 1 and getattr(self.net, 'grad_sync', None) is not None):
            return False
        if switches.step_mode() != 'fused':
            return False
        if ops.TIMER is not None and not ops.TIMER.native_stage()[0]:
            return False
        return True

    def _native_span(self, want):
        """-> how many of the next `want` iterations the native loop can run in one call: marched ahead (in this trainer's queue and the
        sampler's, in the sampler's current window allocation), none of them a grid refresh"""
        if not self._queue or self._queue[0][0] != self.iter or not self._native_ok():
            return 0
        sampler = self.net.sampler
        q = sampler.__dict__.get('_prefetched_q') or []
        win = sampler.__dict__.get('_window')
        n = 0
        while (n < want and n < len(self._queue) and n < len(q) and self._queue[n][0] == self.iter + n and q[n].get('iter') == self.iter + n
               and q[n].get('window') is win and (self.iter + n) % sampler.update_grid_freq != 0):
            n += 1
        return n

    def _window_ok(self):
        """the marches of a window are drawn from a device-resident ray table by the sampler that owns the window's buffers"""
        ok = getattr(self, '_window_static', None)
        if ok is None:
            from .samplers import NGPGridSampler
            ok = self._window_static = hasattr(self.data, 'rays_rgb') and isinstance(self.net.sampler, NGPGridSampler) and self._data_takes_batches()
        return ok

    def _data_takes_batches(self):
        if getattr(self, '_data_takes_out', None) is None:
            import inspect
            self._data_takes_out = 'out' in inspect.signature(self.data.next_batch).parameters
        return self._data_takes_out

    def _run_native(self, k, iter_events=None):
        if self._loop is None:
            self._loop = _NativeLoop(self)
        return self._loop.run(k, iter_events)

    def _draw(self):
        """this iteration's batch, drawn now: into its chunk of the sampler's window where there is one (persistent buffers)"""
        data, sampler = self.data, self.net.sampler
        if self._window_ok() and sampler.device is not None:
            n = min(data.N_rand, data.rays_rgb.shape[0])
            win = sampler.window_for(n, sampler.train_max_samples(n))
            return data.next_batch(out=win.batch_out(self.iter % sampler.WINDOW))
        return data.next_batch()

    def _step_py(self):
        net, data = self.net, self.data
        net.sampler.set_iter(self.iter)                                   # PassSamplerIterHook
        for g in self.opt.param_groups:
            g['lr'] = step_lr(self.base_lr, self.iter)
        if self._queue and self._queue[0][0] < self.iter:
            net.sampler.rewind_marches()                                  # marched for iterations that are over
        batch = self._queue.pop(0)[1].batch() if (self._queue and self._queue[0][0] == self.iter) else None
        if batch is None:
            batch = self._draw()
        n_rays = batch['rays_o'].shape[0]
        # the reference's DataLoader(batch_size=1) collates a leading batch axis that train_step unfolds
        batch = {k: v[None] for k, v in batch.items()}
        # root gradient = a persistent ones tensor (loss.backward() would fill a fresh one every step)
        if self._one is None:
            self._one = torch.ones((), dtype=torch.float32, device=self.device)
            net._unit_root_grad = self._one        # lets the fused step skip its gradient-scaling launch
        # one GPU: the fused step without autograd (`direct`): this loop back-propagates a unit root gradient into cleared .grads and
        # nothing else, which the step can do itself -- it applies the updates (fuse_adam) or leaves its gradients in .grad.  The
        # engine pass, Function.apply and the optimiser wrappers were ~0.1 ms of the ~0.43 ms of interpreter time per iteration,
        # and the interpreter, not the GPU, set the pace (tools/hosttime2.py).  direct_step=False: through autograd.
        direct = self.direct_step
        if direct:
            for p_ in self._opt_params:
                p_.grad = None
        net._fuse_table_update = self.fuse_adam
        try:
            out = net.train_step(batch, self.opt, lazy_log=self.lazy_log, direct=direct)
        finally:
            net._fuse_table_update = False
        if out.get('grads_ready'):
            if not out['updates_applied']:
                self.opt.step()
        else:
            self.opt.zero_grad(set_to_none=True)
            torch.autograd.backward(out['loss'], grad_tensors=self._one)
            if self.world_size > 1 and not net._fused_ok():
                from . import dist as xdist                 # modular step: reduce after backward (DDP semantics)
                xdist.allreduce_grads([p for p in net.parameters() if p.grad is not None], self.world_size)
            self.opt.step(grad_scale=getattr(net, '_pending_grad_scale', 1.0))
            net._pending_grad_scale = 1.0
        if self.world_size > 1 and self.dp_mode == 'zero1':
            net.grad_sync.gather_params()               # every rank's updated shard -> the full table, in place
        data.set_batchsize(net.sampler.n_rays_per_batch)                  # ModifyBatchsizeHook
        self.iter += 1
        self.rays_done += n_rays
        return out

    def _march_ahead(self):
        """Nothing marched ahead: the iterations up to the next grid refresh are drawn and marched now, as one series of launches
        (NGPGridSampler.march_window).  Called by the sampler right behind a grid refresh (the series then starts beside this
        iteration's own march in place and runs on beside its step) and once more when this iteration's samples exist -- which is
        where it happens after a rewind (a frame rendered in the middle of a window took the marches back).  A march reads the rays
        and the occupancy bitfield only: it is never issued across a grid refresh (iterations = 0 mod update_grid_freq), and its
        batches have the size their iterations will have (the size changes after iterations = update_grid_freq - 1 mod
        update_grid_freq, i.e. together with the refresh)."""
        sampler = self.net.sampler
        it, f = self.iter, sampler.update_grid_freq
        if (self.march_window == 'off' or self._queue or sampler.__dict__.get('_prefetched_q') or not self._window_ok()
                or not sampler.can_march_ahead(it + 1)):
            return
        W = sampler.WINDOW
        n_iters = min(f - (it + 1) % f, W - (it + 1) % W)             # up to the next refresh, inside the window's chunks
        data = self.data
        n = min(data.N_rand, data.rays_rgb.shape[0])
        end, made = sampler.march_window(data.rays_rgb, data.cur_i, data.batches_drawn, it + 1, n_iters, n,
                                         on_side=self.march_window == 'side')
        data.cur_i, data.batches_drawn = end, data.batches_drawn + n_iters
        self._queue.extend((it + 1 + j, pf) for j, pf in enumerate(made))

    def _on_sampled(self):
        """Called by the sampler as soon as THIS iteration's samples exist (and the step is enqueued)"""
        sampler = self.net.sampler
        it, f = self.iter, sampler.update_grid_freq
        self._march_ahead()
        if (it + 1) % f == 0 and self.prefetch_k6 and hasattr(sampler, 'prefetch_grid_samples') and sampler._streams():
            # the next iteration starts with a grid refresh: its sample generation (K6 twice + the clear of the temporary grid) depends
            # on nothing this iteration changes
            with torch.cuda.stream(sampler.side_stream()):
                sampler.prefetch_grid_samples(it + 1)

    def _on_rewind(self, first):
        """the sampler took back the marches issued ahead (a test-mode launch in between): the batch cursor and the batch generator's
        call index go back to the first of them"""
        self.data.cur_i, self.data.batches_drawn = int(first['cur_ray']), int(first['batch_index'])
        del self._queue[:]

    @property
    def samples_done(self):
        return self.net.sampler.total_valid_samples()


class _NativeLoop:
    """Host side of xr_ngp_loop_run (include/xrnerf_mi355.h): builds the descriptor from the trainer's, the sampler's and the
    optimiser's persistent buffers, mirrors the counters both paths share, and consumes the marched iterations from the queues the
    per-iteration path would consume them from."""

    def __init__(self, tr):
        self.tr = tr
        self.state = _lib.LoopState()
        self.enqueue_s, self.enqueued = 0.0, 0
        self._keep = None
        self.exchange = None               # data parallel: dist.native_exchange, created with the first descriptor

    # ------------------------------------------------------------------ counters shared with the per-iteration path
    def _adam_states(self):
        tr = self.tr
        mlp = tr.net.mlp
        out = []
        sync = getattr(tr.net, 'grad_sync', None)
        first = sync.shard_param if hasattr(sync, 'shard_param') else mlp.embedder_pos.params     # zero1: the optimiser owns this rank's shard
        for p in (first, mlp.density_net.params, mlp.color_net.params):
            grp = [g for g in tr.opt.param_groups if any(q is p for q in g['params'])]
            if not grp:
                raise _lib.XrError('the native loop needs the three NGP tensors in the trainer\'s FusedAdam')
            out.append(tr.opt._state(p, grp[0]))
        self._group = grp[0]
        return out

    def _descriptor(self, win, n_rows, sets, states, g):
        tr, L = self.tr, _lib.load()
        net = tr.net
        sampler, mlp = net.sampler, net.mlp
        dev = tr.device
        table, wd, wc = mlp.embedder_pos.params, mlp.density_net.params, mlp.color_net.params
        meta = mlp.embedder_pos.meta
        if not ops.hashgrid_bwd_adam_supported(n_rows, meta):
            raise _lib.XrError('the fused table update has no non-atomic scatter path at this row capacity')
        ops._ensure_helper(dev)
        D = _lib.LoopDesc()
        vp = lambda t: t.data_ptr() if t is not None else None
        D.table, D.w_density, D.w_color = vp(table), vp(wd), vp(wc)
        nhd, nhc = mlp.density_net.n_hidden, mlp.color_net.n_hidden
        D.n_hidden_density, D.n_hidden_color, D.pad_value, D.mlp_mode = nhd, nhc, float(mlp.pad_value), ops._mlp_mode(nhd, nhc)
        s_, r_, o_ = meta._args()
        D.n_levels, D.scale_host, D.resolution_host, D.offset_host = meta.n_levels, s_, r_, o_
        sync = getattr(net, 'grad_sync', None)
        zero1 = hasattr(sync, 'shard_param')
        for name, p, st in zip(('adam_table', 'adam_w_density', 'adam_w_color'), (sync.shard_param if zero1 else table, wd, wc), states):
            ema = st.get('ema') if g['ema_momentum'] is not None else None
            a = ops.adam_fuse(p.data, st['m'], st['v'], ema, 0, 0.0, g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], 0.0, 1.0)
            setattr(D, name, a)
        if sync is not None:
            from . import dist as xdist
            if self.exchange is None:
                self.exchange = xdist.native_exchange(tr.world_size, tr.rank)
            ex = self.exchange
            ex.register(*[b.zero_block for b in sets])
            D.exchange = C.pointer(ex.c)
            D.dp_mode = 1 if zero1 else 0
            D.split_level = meta.n_levels - 8 if (not zero1 and meta.n_levels > 8 and getattr(sync, 'split_levels', True)) else 0
            if zero1:
                D.shard_grad, D.table_padded, D.shard_floats = vp(sync.shard_grad), vp(sync.param_padded), sync.shard
                ex.register(sync.shard_grad, sync.param_padded, *[sync._padded_of(b.g_table) for b in sets])
            else:
                ex.register(*[b.g_table for b in sets])
        D.density_grid_mean = vp(sampler.density_grid_mean)
        D.rgb_activation, D.density_activation, D.huber_delta, D.loss_scale = int(sampler.rgb_activation), int(sampler.density_activation), 0.1, 5.0
        D.n_rows, D.ld = n_rows, sets[0].ld
        D.window = win.c
        for i, b in enumerate(sets):
            B = D.step[i]
            B.enc_t, B.raw, B.draw, B.denc_t, B.rgb_out, B.zero_block = vp(b.enc_t), vp(b.raw), vp(b.draw), vp(b.denc_t), vp(b.rgb), vp(b.zero_block)
            B.zero_floats = b.zero_block.numel()
            B.grad_w_density, B.grad_w_color, B.loss_mse, B.live_seg_count = vp(b.g_wd), vp(b.g_wc), vp(b.loss_mse), vp(b.live_seg)
            B.grad_table = vp(b.g_table) if sync is not None else None
        ws_mlp, live_list, _, live_stats = ops._list_slots(dev, n_rows, nhd, nhc)
        ws_sc = ops._ws(dev, L.xr_hashgrid_bwd_workspace_bytes(n_rows, meta.n_levels, r_, o_), 'hgb')
        D.ws_mlp_bwd, D.ws_mlp_bwd_bytes = vp(ws_mlp), ws_mlp.numel()
        D.ws_scatter, D.ws_scatter_bytes = vp(ws_sc), ws_sc.numel()
        D.stream = ops._stream()
        self._hold = (win, sets, ws_mlp, ws_sc, states)          # (what the pointers name stays alive)
        return D, live_list, live_stats

    # ------------------------------------------------------------------ k marched iterations
    def run(self, k, iter_events=None):
        tr, S, L = self.tr, self.state, _lib.load()
        net, data = tr.net, tr.data
        sampler, mlp = net.sampler, net.mlp
        f = sampler.update_grid_freq
        q = sampler._prefetched_q
        if k < 1 or len(q) < k or len(tr._queue) < k or q[0].get('iter') != tr.iter:
            raise _lib.XrError('the native loop runs marched iterations only')
        if tr.iter % f == 0 or tr.iter % f + k > f:
            raise _lib.XrError('a native window never crosses a grid refresh')
        dev = tr.device
        pfs = q[:k]
        win, n_rays, max_samples = pfs[0]['window'], pfs[0]['n'], pfs[0]['max_samples']
        n_rows = min(sampler.target_batch_size, max_samples)
        table, wd, wc = mlp.embedder_pos.params, mlp.density_net.params, mlp.color_net.params
        meta = mlp.embedder_pos.meta
        sets = getattr(net, '_step_bufs', None)
        if (sets is None or sets[0].n_rows != n_rows or sets[0].ray_cap < n_rays or sets[0].g_table.shape != table.shape
                or sets[0].g_table.device != table.device):
            sets = net._step_bufs = [ops.TrainStepBuffers(dev, n_rows, max(n_rays, 1 << 15), table.numel(), wd.numel(), wc.numel(), meta,
                                                          getattr(getattr(net, 'grad_sync', None), 'pad_grad', None)) for _ in range(2)]
            net._step_turn = 0
        states = self._adam_states()
        g = self._group
        steps = [st['step'] for st in states]
        if len(set(steps)) != 1:
            raise _lib.XrError('the three NGP tensors have different optimiser step counts: %r' % (steps,))
        S.iter, S.step_turn, S.adam_step = tr.iter, getattr(net, '_step_turn', 0) & 1, steps[0]
        # the descriptor is rebuilt only when something it names has changed (a buffer that grew, another precision mode): ~40 pointer
        # conversions and three workspace queries otherwise sit in front of every window's first kernel.  Keyed on what the pointers
        # ARE (data pointers and sizes), not on object identities
        dp = lambda t: (t.data_ptr(), t.numel()) if t is not None else (0, 0)
        key = (n_rows, dp(win.coords), dp(win.rays_o), dp(win.xyz), win.ray_stride, win.coords_stride,
               tuple(dp(b.enc_t) + dp(b.zero_block) + dp(b.rgb) for b in sets), dp(table), dp(wd), dp(wc),
               dp(sampler.density_grid_mean), ops._mlp_mode(mlp.density_net.n_hidden, mlp.color_net.n_hidden), ops._stream().value,
               tuple(dp(b.g_table) for b in sets), id(getattr(net, 'grad_sync', None)),
               tuple(dp(st['m']) + dp(st['v']) + dp(st.get('ema')) for st in states),
               dp(ops._workspaces.get((str(dev), 'mlpbwd'))), dp(ops._workspaces.get((str(dev), 'hgb'))))
        if self._keep is not None and self._keep[0] == key:
            _, D, live_list, live_stats = self._keep
        else:
            D, live_list, live_stats = self._descriptor(win, n_rows, sets, states, g)
            key = key[:-2] + (dp(ops._workspaces.get((str(dev), 'mlpbwd'))), dp(ops._workspaces.get((str(dev), 'hgb'))))
            self._keep = (key, D, live_list, live_stats)
        # the schedules are this trainer's: lr per iteration, the EMA momentum of mmcv's EMAHook per update
        lr = (C.c_float * k)(*[step_lr(tr.base_lr, tr.iter + j) for j in range(k)])
        mom = (C.c_float * k)(*[(FusedAdam._ema_momentum(g, int(S.adam_step) + 1 + j) if g['ema_momentum'] is not None else 0.0) for j in range(k)])
        stage, tev, tarr = None, None, None
        if ops.TIMER is not None:
            ok, stage = ops.TIMER.native_stage()
            if not ok:
                raise _lib.XrError('this KernelTimer needs the per-entry-point launch sequence')
            if stage is not None:
                tev = [ops._CEvent() for _ in range(2 * k)]
                tarr = (C.c_void_p * (2 * k))(*[e.h for e in tev])
        iarr = (C.c_void_p * (k + 1))(*[e.h for e in iter_events]) if iter_events is not None else None
        ops.LIVE_STATS = live_stats
        sets[0].live = sets[1].live = (live_list, live_stats) if os.environ.get('XR_MLP_LIVE') != '0' else None
        sampler._wait_march(pfs[0])                       # the compute stream behind the window's marches: once per window
        ops.mlp_range_tracking(dev, False)                # the iterations between two refreshes run the forward without its range count
        t_enq = time.perf_counter()
        try:
            rc = L.xr_ngp_loop_run(C.byref(D), C.byref(S), k, n_rays, lr, mom, stage.encode() if stage else None, tarr, iarr)
        finally:
            self.enqueue_s += time.perf_counter() - t_enq                # host time inside the native call (tools/hosttime2.py)
            ops.mlp_range_tracking(dev, True)
        self.enqueued += k
        if rc != 0:
            if self.exchange is not None and self.exchange.error is not None:
                err, self.exchange.error = self.exchange.error, None
                raise err
            _lib.check(rc, 'xr_ngp_loop_run')
        if stage is not None:
            ops.TIMER.events.setdefault(stage, []).extend((tev[2 * j], tev[2 * j + 1], 0) for j in range(k))
        sampler._pending_counts.extend(pf['host'] for pf in pfs)
        del q[:k]
        del tr._queue[:k]
        tr.iter = int(S.iter)
        net._step_turn = int(S.step_turn)
        for st in states:
            st['step'] = int(S.adam_step)
        tr.rays_done += k * n_rays
        # the sampler's public state = the last iteration's (what a reader between two steps sees on the per-iteration path too)
        last, pf = tr.iter - 1, pfs[-1]
        sampler.iter_n = last
        sampler.coords, sampler.xyz = pf['out'][0][:n_rows], pf['xyz']
        sampler.rays_index, sampler.rays_numsteps, sampler.rays_numsteps_compacted = pf['out'][1], pf['out'][2], pf['clipped'][0]
        sampler.n_valid_dev = pf['clipped'][1][0:1]
        b = sets[int(S.last_step_set)]
        net._last = {'rgb': b.rgb[:n_rays], 'loss_mse': b.loss_mse, 'raw': b.raw}
        if last % f == f - 1:
            sampler.update_batch_rays(True, max_samples)                  # drains the window's counters (waits for the side stream's copy)
            if tr.prefetch_k6 and hasattr(sampler, 'prefetch_grid_samples'):
                with torch.cuda.stream(sampler.side_stream()):
                    sampler.prefetch_grid_samples(last + 1)
        data.set_batchsize(sampler.n_rays_per_batch)                      # ModifyBatchsizeHook
        from .networks import _LazyPsnr
        loss = b.loss_mse[0:1].reshape(())
        for p_ in tr._opt_params:
            p_.grad = None                  # (the summed gradients were consumed by the loop's own optimiser launches)
        return {'loss': loss, 'log_vars': {'loss': loss, 'psnr': _LazyPsnr(b.loss_mse, n_rays)}, 'num_samples': n_rays,
                'grads_ready': True, 'updates_applied': True}


@torch.no_grad()
def render_frame(net, pose43, H, W, focal, chunk=None, row0=0, nrows=None, idx=0):
    """val/test forward of one camera (HashGetRays -> FlattenRays -> HashSetImgids -> batchify_forward;
    pipelines/create.py:356-425, networks/nerf.py:50-69) with the rays generated on the device.
    rows [row0,row0+nrows) only -> the image-space shard of one rank."""
    dev = next(net.parameters()).device
    nrows = H - row0 if nrows is None else nrows
    o, d = ops.gen_rays(pose43, H, W, focal, focal, 0.5 * W, 0.5 * H, row0, nrows, device=dev)
    data = {'rays_o': o, 'rays_d': d, 'img_ids': torch.full((o.shape[0], 1), idx, dtype=torch.int32, device=dev)}
    old = net.chunk
    net.chunk = chunk or o.shape[0]
    try:
        ret = net.batchify_forward(data, is_test=True)
    finally:
        net.chunk = old
    return ret['rgb'].reshape(nrows, W, 3), ret['alpha'].reshape(nrows, W, 1)


ERT_SLICES = (0, 4, 8, 16, 32, 64, 128, 256, 512, 1024)


@torch.no_grad()
def render_rays_ert(net, o, d, eps=1e-4, bg=None):
    """Early-terminated rendering of a set of rays (the optional fast path behind XRNERF_FRAME=ert and `render_frame_ert`; the reference
    has no early termination -- its K5 leaves EPSILON unused, calc_rgb.cu:144-206 -- so the default frame path evaluates every marched
    sample).  The rays are marched once (K1), then evaluated in depth slices: before each slice the rays whose transmittance is
    still above `eps` are compacted (wave ballots) and ONLY their samples go through encode + MLP.  Pixels differ from the full
    evaluation by less than `eps` (everything skipped is weighted by T < eps).  -> (rgb [n,3], alpha [n,1])"""
    sampler, mlp, render = net.sampler, net.mlp, net.render
    dev = o.device
    data = sampler.sample({'rays_o': o, 'rays_d': d}, mlp, True)          # K1 only; one read-back
    coords, numsteps = sampler.coords, sampler.rays_numsteps
    n_rays, total = o.shape[0], coords.shape[0]
    T = torch.ones((n_rays,), dtype=torch.float32, device=dev)
    acc = torch.zeros((n_rays, 3), dtype=torch.float32, device=dev)
    ray_off = torch.empty((n_rays,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    rows = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
    ra, da = int(sampler.rgb_activation), int(sampler.density_activation)
    meta, wd, wc = mlp.embedder_pos.meta, mlp.density_net.params, mlp.color_net.params
    evaluated = 0
    for s0, s1 in zip(ERT_SLICES[:-1], ERT_SLICES[1:]):
        ops.render_slice_select(numsteps, T, s0, s1, eps, rows, ray_off, count)
        m = int(count.item())                                            # slice size decides the launches
        if m == 0:
            break
        evaluated += m
        r = rows[:m]
        enc_t = ops.hashgrid_fwd(mlp.embedder_pos.params, coords[:, :3], meta, rows=r)
        raw = ops.nerf_mlp_fwd(enc_t, coords[:, 4:], m, wd, wc, mlp.density_net.n_hidden, mlp.color_net.n_hidden,
                               mlp.pad_value, rows=r)
        ops.render_slice_composite(raw, coords, numsteps, ray_off, s0, s1, ra, da, T, acc)
    bgc = (render.bg_color if bg is None else torch.as_tensor(bg, dtype=torch.float32)).to(dev)
    render_rays_ert.last_evaluated = (evaluated, total)
    return acc + T[:, None] * bgc[None, :], (1.0 - T)[:, None]


@torch.no_grad()
def render_frame_ert(net, pose43, H, W, focal, eps=1e-4, row0=0, nrows=None, bg=None):
    """`render_rays_ert` for one camera (rows [row0, row0 + nrows) of its image)"""
    dev = next(net.parameters()).device
    nrows = H - row0 if nrows is None else nrows
    o, d = ops.gen_rays(pose43, H, W, focal, focal, 0.5 * W, 0.5 * H, row0, nrows, device=dev)
    rgb, alpha = render_rays_ert(net, o, d, eps, bg)
    render_frame_ert.last_evaluated = render_rays_ert.last_evaluated
    return rgb.reshape(nrows, W, 3), alpha.reshape(nrows, W, 1)
