"""`HashNerfNetwork`: sampler -> mlp -> render, the registered type of
/root/reference/xrnerf/models/networks/hashnerf.py:16-112 (base class behaviour from
networks/nerf.py:23-69,171-173 and networks/base.py:9-37)."""
import os
import time

import torch
from torch import nn

from . import builder, switches
from .builder import NETWORKS


def get_dist_info():
    """mmcv.runner.get_dist_info"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# networks/utils/metrics.py:3-16
def img2mse(x, y):
    return torch.mean((x - y) ** 2)


_LOG10 = float(torch.log(torch.tensor(10.0)))


def mse2psnr(x):
    # the reference divides by torch.log(torch.Tensor([10.]).to(x.device)): a pageable host-to-device copy,
    # i.e. a full stream synchronisation per call.  Same value, no copy.
    return -10. * torch.log(x) / _LOG10


class _HuberSumFn(torch.autograd.Function):
    """sum-reduced HuberLoss and its gradient in one launch (xr_huber_loss_grad) instead of ~10 elementwise ops"""

    @staticmethod
    def forward(ctx, x, y, delta):
        from . import ops
        loss, grad = ops.huber_loss_grad(x.contiguous(), y.contiguous(), delta, 1.0)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def HuberLoss(x, y, delta=0.1, reduction='sum'):
    if reduction == 'sum' and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape:
        return _HuberSumFn.apply(x, y, delta)
    rel = (x - y).abs()
    sqr = 0.5 / delta * rel * rel
    loss = torch.where(rel > delta, rel - 0.5 * delta, sqr)
    if reduction == 'mean':
        loss = loss.mean()
    elif reduction == 'sum':
        loss = loss.sum()
    return loss


def unfold_batching(data):   # networks/utils/batching.py:5-12
    if len(data.shape) > 1:
        bs = data.shape[0]
        if bs == 1:
            return data[0]          # same values as the reference's torch.cat of one piece, without the copy
        data = torch.cat([data[b] for b in range(bs)], 0)
    return data


def recover_shape(data, to_shape):   # networks/utils/transforms.py:5-9
    to_shape = list(to_shape[:-1]) + list(data.shape[1:])
    return torch.reshape(data, to_shape)


from ._fastattr import _FastAttr


class _DirectCtx:
    """stands in for the autograd context when the trainer runs the fused step without autograd (HashNerfNetwork.train_step(...,
    direct=True)): the step either applies the updates itself or hands its gradients straight to `.grad`"""

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


class _FusedTrainStepFn(torch.autograd.Function):
    """The whole training forward AND backward of HashNerfNetwork as ONE autograd node.

    forward  = sampler.sample -> encode -> fused MLP -> K3 composite -> 5*Huber (+ masked mse)
               -> K4 -> MLP backward -> hash-grid scatter, all enqueued back to back (10 launches);
    backward = hand the gradients computed above to autograd (scaled by the incoming gradient).
    Same kernels and same results as the modular path (mlp(data) -> render -> HuberLoss -> .backward()),
    without the per-node autograd/Python latency between them (tests/test_gpu_network.py)."""

    @staticmethod
    def forward(ctx, table, wd, wc, net, data):
        from . import ops
        mlp, sampler = net.mlp, net.sampler
        with torch.no_grad():
            # the sampler's `on_sampled` hook (the trainer issues the NEXT batch's march on a side stream from it: ~10
            # Python-side launches, events, pinned copies) is deferred until this step's encode and MLP forward are
            # enqueued: in the kernel trace the main stream sat idle ~90 us between K2 and the encode while the host was
            # busy issuing that prefetch (profiles/r02_trace_normal_iteration.txt)
            cb, sampler.on_sampled = getattr(sampler, 'on_sampled', None), None
            try:
                data = sampler.sample(data, mlp, False)
            finally:
                sampler.on_sampled = cb
            sync = getattr(net, 'grad_sync', None)
            if table.is_cuda and (mlp.density_net.n_hidden, mlp.color_net.n_hidden) in ops._FUSED_BWD and \
                    switches.step_mode() == 'fused' and (ops.TIMER is None or ops.TIMER.native_stage()[0]):
                # the whole device side of the step as ONE native call (csrc/xr_step.hip) -- the same entry points in the same
                # order as the Python sequence below, which stays for the kernels' host build and whenever a KernelTimer wants
                # events around the individual entry points (bench.py's roofline windows).  Data parallel: the native call
                # stops after the scatter of the 8 finest hash levels; their 32-MB gradient slice goes to the collective, which
                # then runs under the scatter of the coarser levels (one more native call) -- the same bucket protocol as the
                # Python sequence, without ~0.5 ms of interpreter time per iteration on every rank
                n_rows = sampler.coords.shape[0]
                sets = getattr(net, '_step_bufs', None)
                n_rays = sampler.rays_numsteps.shape[0]
                if (sets is None or sets[0].n_rows != n_rows or sets[0].ray_cap < n_rays or sets[0].g_table.shape != table.shape
                        or sets[0].g_table.device != table.device):
                    sets = net._step_bufs = [ops.TrainStepBuffers(table.device, n_rows, max(n_rays, 1 << 15), table.numel(), wd.numel(),
                                                                  wc.numel(), mlp.embedder_pos.meta, getattr(sync, 'pad_grad', None))
                                             for _ in range(2)]
                    net._step_turn = 0
                # two sets alternate: the one the optimiser still holds as .grad is not reused.  A caller that keeps gradients
                # alive across steps (zero_grad(set_to_none=False), accumulation over several backward passes) may still hold
                # EITHER set as .grad: the step WRITES its gradient buffers, so such a set is skipped, and when both are held a
                # fresh one takes the place of the older
                # (zero1: the full table is not the optimiser's -- its shard is -- so no gradient is delivered to it, see below)
                deliver_table = not hasattr(sync, 'shard_param')
                held = {p_.grad.data_ptr() for p_ in ((table, wd, wc) if deliver_table else (wd, wc)) if p_.grad is not None}
                for _ in range(2):
                    net._step_turn ^= 1
                    b = sets[net._step_turn]
                    if not held & {b.g_table.data_ptr(), b.g_wd.data_ptr(), b.g_wc.data_ptr()}:
                        break
                else:
                    b = sets[net._step_turn] = ops.TrainStepBuffers(table.device, n_rows, sets[0].ray_cap, table.numel(), wd.numel(),
                                                                     wc.numel(), mlp.embedder_pos.meta, getattr(sync, 'pad_grad', None))
                meta = mlp.embedder_pos.meta
                split = meta.n_levels - 8 if (sync is not None and meta.n_levels > 8 and getattr(sync, 'split_levels', True)) else 0
                # one GPU, a trainer that opted in (net._fuse_table_update: unit root gradient, gradients cleared every step) and
                # an optimiser that can hand its updates over (FusedAdam.fused_update): the step applies them itself -- the table's
                # inside the scatter, the MLP tensors' behind the reduction of their gradients
                adam = mlp_adam = None
                opt = getattr(net, '_step_optimizer', None)
                if (sync is None and getattr(net, '_fuse_table_update', False) and hasattr(opt, 'fused_update')
                        and getattr(net, '_unit_root_grad', None) is not None):
                    key = (n_rows, id(meta))
                    if getattr(net, '_fuse_ok_key', None) != key:
                        net._fuse_ok_key, net._fuse_ok = key, ops.hashgrid_bwd_adam_supported(n_rows, meta)
                    if net._fuse_ok:
                        adam = opt.fused_update(table)
                        mlp_adam = (opt.fused_update(wd), opt.fused_update(wc)) if adam is not None else None
                        if mlp_adam is not None and (mlp_adam[0] is None or mlp_adam[1] is None or mlp_adam[0].step != mlp_adam[1].step):
                            raise RuntimeError('the optimiser handed over the table update but not a joint update of the two MLP tensors')
                rgb = ops.ngp_train_step(table, wd, wc, mlp.density_net.n_hidden, mlp.color_net.n_hidden, mlp.pad_value, meta, sampler.coords, data.get('n_valid_dev'),
                                         sampler.rays_numsteps, sampler.rays_numsteps_compacted, data['bg_color'],
                                         data['target_s'].contiguous(), data['alpha'].contiguous(), sampler.density_grid_mean,
                                         int(sampler.rgb_activation), int(sampler.density_activation), b, scatter_level0=split,
                                         xyz=getattr(sampler, 'xyz', None), adam=adam, mlp_adam=mlp_adam)
                if sync is not None:
                    sync.ready(b.g_mlp)
                    if split:
                        cut = 2 * int(meta.offset[split])
                        sync.ready(b.g_table[cut:])
                        ops.hashgrid_bwd(sampler.coords[:n_rows], b.denc_t, meta, b.g_table, live=b.live, n_dev=data.get('n_valid_dev'),
                                         levels=(0, split), overwrite=True)
                        sync.ready(b.g_table[:cut])
                    else:
                        sync.ready(b.g_table)
                if cb is not None:
                    cb()
                # XRNERF_DP=zero1: the table gradient's consumer is the reduce-scatter (sync.ready above) and the optimiser owns this rank's
                # SHARD, whose .grad sync.finish() sets; a .grad on the full table would never be cleared by that optimiser's zero_grad
                # and would pin a 48.8-MB buffer set per step
                ctx.grads = () if adam is not None else (b.g_table, b.g_wd, b.g_wc) if deliver_table else (b.g_wd, b.g_wc)
                ctx.params = () if adam is not None else (table, wd, wc) if deliver_table else (wd, wc)
                ctx.table_updated = adam is not None
                ctx.sync = sync
                ctx.unit_root_grad = getattr(net, '_unit_root_grad', None)
                ctx.net = net
                ctx.mark_non_differentiable(rgb)
                ctx.set_materialize_grads(False)
                # LIFETIME: loss, rgb and raw are views of the recycled buffer set -- valid until the step after next writes it.
                # train_step reads its log scalars at once unless the caller opts into lazy_log, whose contract says so.
                net._last = {'rgb': rgb, 'loss_mse': b.loss_mse, 'raw': b.raw}
                return b.loss_mse[0:1].reshape(()), rgb
            pts, dirs = mlp._rows(data['pts']), mlp._rows(data['viewdirs'])
            n, n_dev = pts.shape[0], data.get('n_valid_dev')
            meta, nhd, nhc = mlp.embedder_pos.meta, mlp.density_net.n_hidden, mlp.color_net.n_hidden
            # One stream, one launch per kernel.  (Measured and removed: a two-stream pipeline over 4 row chunks --
            # encode of chunk c+1 beside the MLP of chunk c -- 1.55 -> 2.48 ms/step: the MFMA workgroups need a
            # whole CU each and do not co-schedule with the gather / scatter waves.)
            ld = (n + 63) // 64 * 64
            enc_t = torch.empty((meta.n_output_dims, ld), dtype=torch.float32, device=pts.device)
            raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
            ops.hashgrid_fwd(table, pts, meta, enc_t=enc_t, ld=ld, n_dev=n_dev)
            ops.nerf_mlp_fwd(enc_t, dirs, n, wd, wc, nhd, nhc, mlp.pad_value, raw=raw, n_dev=n_dev)
            if cb is not None:
                cb()
            ra, da = int(sampler.rgb_activation), int(sampler.density_activation)
            # one small zero-fill for what must start at zero: the two MLP gradient buffers and the (loss, mse)
            # accumulators.  dL/draw needs none when the valid row count is on the device: rows [0, n_valid) are exactly
            # the rays' (base, count) ranges, all written by K4, and nothing downstream reads a row behind n_valid
            n_seg = ops.live_segments(n)
            nz = wd.numel() + wc.numel() + 4 + n_seg
            zbuf = torch.zeros((nz,), dtype=torch.float32, device=raw.device)
            draw = torch.empty_like(raw) if n_dev is not None else torch.zeros_like(raw)
            g_mlp = zbuf[:wd.numel() + wc.numel()]                                    # both MLP gradients, contiguous
            g_wd, g_wc = g_mlp[:wd.numel()], g_mlp[wd.numel():]
            loss_mse = zbuf[nz - n_seg - 4:nz - n_seg - 2]
            live_seg = zbuf[nz - n_seg:].view(torch.int32)       # per-segment live-row counts, filled by the compositor
            # K3 -> 5 * Huber (+ masked MSE for the logged PSNR) -> K4 as ONE launch (xr_composite_train)
            rgb = ops.composite_train(raw, sampler.coords, sampler.rays_numsteps, sampler.rays_numsteps_compacted,
                                      data['bg_color'], data['target_s'].contiguous(), data['alpha'].contiguous(),
                                      sampler.density_grid_mean, ra, da, None, draw, delta=0.1, scale=5.0, live_seg=live_seg)
            ops.train_loss_scalars(rgb, data['target_s'].contiguous(), data['alpha'].contiguous(), 0.1, 5.0, out=loss_mse)
            g_table = torch.zeros_like(table)
            denc_t = torch.empty_like(enc_t)
            # samples behind an opaque surface have an exactly-zero dL/d(raw) row (T == 0): the MLP backward and the scatter
            # run on the list of the others (more than half of the rows are dead in steady state)
            live = ops.live_rows(draw, n, n_dev=n_dev, seg_counts=live_seg)       # (the compositor counted them per segment)
            ops.nerf_mlp_bwd(enc_t, dirs, n, wd, wc, nhd, nhc, draw, g_wd, g_wc, mlp.pad_value, denc_t=denc_t, n_dev=n_dev, live=live)
            sync = getattr(net, 'grad_sync', None)
            if sync is None:
                ops.hashgrid_bwd(pts, denc_t, meta, g_table, live=live)
            elif meta.n_levels > 8 and getattr(sync, 'split_levels', True):
                # data parallel: reduce each gradient bucket across the ranks while the next one is produced
                split = meta.n_levels - 8
                cut = 2 * int(meta.offset[split])
                sync.ready(g_mlp)
                ops.hashgrid_bwd(pts, denc_t, meta, g_table, live=live, levels=(split, meta.n_levels))
                sync.ready(g_table[cut:])
                ops.hashgrid_bwd(pts, denc_t, meta, g_table, live=live, levels=(0, split))
                sync.ready(g_table[:cut])
            else:
                if hasattr(sync, 'pad_grad'):          # padded storage for the reduce-scatter
                    g_table = sync.pad_grad(table.device)[1]
                ops.hashgrid_bwd(pts, denc_t, meta, g_table, live=live)
                sync.ready(g_mlp)
                sync.ready(g_table)
        deliver_table = not hasattr(sync, 'shard_param')
        ctx.grads = (g_table, g_wd, g_wc) if deliver_table else (g_wd, g_wc)
        ctx.params = (table, wd, wc) if deliver_table else (wd, wc)
        ctx.sync = getattr(net, 'grad_sync', None)
        ctx.unit_root_grad = getattr(net, '_unit_root_grad', None)
        ctx.net = net
        ctx.mark_non_differentiable(rgb)
        ctx.set_materialize_grads(False)         # no zero-filled dL/drgb tensor for the non-differentiable output
        net._last = {'rgb': rgb, 'loss_mse': loss_mse, 'raw': raw}
        return loss_mse[0:1].reshape(()), rgb

    @staticmethod
    def backward(ctx, g, _g_rgb):
        _FusedTrainStepFn.deliver(ctx, g)
        return None, None, None, None, None

    @staticmethod
    def deliver(ctx, g):
        """hands the step's gradients to the parameters' .grad (the body of backward; also called directly by the autograd-free
        step with the registered unit root gradient)"""
        from . import ops
        grads = list(ctx.grads)
        params = ctx.params
        ctx.grads = ctx.params = None
        factor = ctx.sync.finish() if ctx.sync is not None else 1.0      # all buckets reduced; average over the ranks
        if g is None:                            # only the non-differentiable output was used downstream
            return
        unit = ctx.unit_root_grad
        is_unit = unit is not None and g.data_ptr() == unit.data_ptr()
        if getattr(ctx, 'table_updated', False) and not (is_unit and factor == 1.0):
            raise RuntimeError('this step already applied the table update for a unit root gradient (net._fuse_table_update): '
                               'back-propagate the registered unit gradient, or switch the fused update off')
        net = ctx.net
        ctx.net = None
        if factor != 1.0 and is_unit and getattr(net, '_defer_grad_scale', False):
            # data parallel, trainer-owned optimiser: .grad keeps the all-reduced SUM and the optimiser multiplies by
            # 1/world_size while it reads the gradient (xr_adam_step_multi's grad_scale) -- no scaling pass over 48.8 MB
            # (set, not compounded: the factor belongs to the SUM the collective left in .grad; the trainer's optimiser consumes
            # and clears it -- a second backward before that adds another all-reduced sum with the same 1 / world_size)
            net._pending_grad_scale = factor
        elif not (factor == 1.0 and is_unit):
            # (the trainer back-propagates from a persistent all-ones root gradient it registers as
            # `net._unit_root_grad`: the scaling launch -- which would read that 1.0 and do nothing -- is skipped then)
            g = g.reshape(1) if g.dtype == torch.float32 and ops._on_device(g) else g.to(grads[0].device, torch.float32).reshape(1)
            ops.scale_multi(grads, g, factor)        # one launch; no memory traffic when the factor is exactly 1
        # Hand the gradients to the parameters the way AccumulateGrad would, but without its defensive clone
        # (a Python-created gradient is never "stolen": 48.8 MB copied per step): first gradient -> becomes
        # .grad, otherwise accumulate in place.
        for p_, g_ in zip(params, grads):
            if p_.grad is None or p_.grad.data_ptr() == g_.data_ptr():
                p_.grad = g_                    # (same storage: the step already wrote this gradient into it)
            else:
                p_.grad.add_(g_)


class _LazyPsnr:
    """PSNR of the alpha-masked prediction (networks/hashnerf.py:40-42), evaluated only when somebody looks
    (float() / .item()): the three tiny elementwise launches and the read-back stay off the training loop"""

    def __init__(self, loss_mse, bs):
        self._t, self._bs = loss_mse, bs

    def tensor(self):
        with torch.no_grad():
            return mse2psnr(self._t[1] / (3.0 * self._bs))

    def item(self):
        return float(self.tensor())

    __float__ = item


class BaseNerfNetwork(nn.Module):
    def __init__(self, **kwarg):
        super().__init__()

    def train_step(self, data, optimizer, **kwargs):
        raise NotImplementedError

    def val_step(self, data, **kwargs):
        raise NotImplementedError


@NETWORKS.register_module()
class HashNerfNetwork(_FastAttr, BaseNerfNetwork):
    _FAST_ATTRS = frozenset(('_fuse_table_update', '_step_optimizer', '_step_turn', '_last', '_pending_grad_scale', '_direct_step',
                             '_fuse_ok_key', '_fuse_ok'))

    def __init__(self, cfg, sampler=None, mlp=None, render=None):
        super().__init__()
        cfg = builder.ConfigDict.wrap(dict(cfg))
        self.phase = cfg.get('phase', 'train')
        if 'chunk' in cfg: self.chunk = cfg.chunk
        if 'bs_data' in cfg: self.bs_data = cfg.bs_data
        self.sampler = builder.build_sampler(sampler)
        self.mlp = builder.build_mlp(mlp)
        self.render = builder.build_render(render)

    def forward(self, data, is_test=False):
        data = self.sampler.sample(data, self.mlp, is_test)
        if is_test:
            data['reuse_buffers'] = True          # (the renderer below consumes `raw` at once: HashNerfMLP may hand out its persistent buffer)
        data = self.mlp(data)
        data, ret = self.render(data, self.sampler, is_test)
        return ret

    def batchify_forward(self, data, is_test=False):
        """forward in smaller minibatches (networks/nerf.py:50-69).  A test-mode frame cut into several chunks (the config's
        chunk = 4096: 157 chunks per 800x800 frame) is marched WITHOUT a host read-back per chunk (samplers.begin_async_test):
        one check at the end of the frame, chunks whose sample buffer overflowed are done again -- same pixels as the
        synchronous form (XRNERF_FRAME=sync), which costs one device-to-host round trip per chunk."""
        N = data[self.bs_data].shape[0]
        from .samplers import NGPGridSampler
        if is_test and switches.frame_mode() == 'ert' and type(self.sampler) is NGPGridSampler and self._fused_types():
            # optional (the reference evaluates every marched sample): early ray termination behind the registry's val / test steps
            from .train import render_rays_ert
            rgb, alpha = render_rays_ert(self, data['rays_o'].contiguous().float(), data['rays_d'].contiguous().float())
            return {'rgb': rgb, 'alpha': alpha}
        if (is_test and (N > self.chunk or getattr(self.sampler, 'frame_ray0', 0)) and type(self.sampler) is NGPGridSampler
                and switches.frame_mode() == 'one_launch'):
            # The whole frame as ONE launch per kernel with the SAME pixels as the chunk loop: encode, MLP and compositor are
            # per-sample / per-ray maps, and K1 -- whose hidden generator the loop would advance once per chunk -- draws every
            # ray's jitter as the ray's chunk-th launch would (`frame_chunk`).  61 ms -> 6 ms per 800x800 frame at chunk = 4096
            # (157 chunks x ~0.4 ms of launches); XRNERF_FRAME=async | sync runs the loop.
            self.sampler.frame_chunk = int(self.chunk)
            try:
                return self.forward(dict(data), is_test)
            finally:
                self.sampler.frame_chunk = 0
        pieces = []
        for i in range(0, N, self.chunk):
            data_chunk = {}
            for k in data:
                if torch.is_tensor(data[k]) and data[k].dim() > 0 and data[k].shape[0] == N:
                    data_chunk[k] = data[k][i:i + self.chunk]
                else:
                    data_chunk[k] = data[k]
            pieces.append(data_chunk)
        use_async = (is_test and len(pieces) > 1 and hasattr(self.sampler, 'begin_async_test') and self.sampler._streams()
                     and switches.frame_mode() != 'sync')
        if use_async:
            self.sampler.begin_async_test()
        rets = [self.forward(dict(c), is_test) for c in pieces]
        if use_async:
            for ci, k1_index in self.sampler.end_async_test():
                after = self.sampler.k1_calls
                self.sampler.k1_calls = k1_index                 # same jitter stream as the first attempt
                rets[ci] = self.forward(dict(pieces[ci]), is_test)
                self.sampler.k1_calls = after
        all_ret = {}
        for ret in rets:
            for k in ret:
                all_ret.setdefault(k, []).append(ret[k])
        return {k: torch.cat(all_ret[k], 0) for k in all_ret}

    def _fused_types(self):
        from .mlps import HashNerfMLP
        from .renders import HashNerfRender
        return type(self.mlp) is HashNerfMLP and type(self.render) is HashNerfRender

    def _fused_ok(self):
        from .mlps import HashNerfMLP
        from .renders import HashNerfRender
        from .samplers import NGPGridSampler
        from . import ops
        import os
        return (switches.step_mode() != 'modular' and type(self.sampler) is NGPGridSampler and
                type(self.mlp) is HashNerfMLP and type(self.render) is HashNerfRender and
                ops._on_device(self.mlp.embedder_pos.params) and torch.is_grad_enabled())

    def _train_step_fused(self, data, **kwargs):
        direct = kwargs.get('direct', False) and getattr(self, 'grad_sync', None) is None and getattr(self, '_unit_root_grad', None) is not None
        if direct:
            # no autograd graph: the caller promises what the graph would have been used for -- one backward pass from a unit root
            # gradient with cleared .grads (xrnerf_amd.train.Trainer).  ~80 us of engine, Function.apply and optimiser-wrapper time per
            # iteration that the host does not have: at 0.45 ms per step the loop was bound by the interpreter, not by the GPU
            ctx = _DirectCtx()
            with torch.no_grad():
                loss, rgb = _FusedTrainStepFn.forward(ctx, self.mlp.embedder_pos.params, self.mlp.density_net.params,
                                                      self.mlp.color_net.params, self, data)
            applied = getattr(ctx, 'table_updated', False)
            if not applied:
                _FusedTrainStepFn.deliver(ctx, self._unit_root_grad)
        else:
            applied = False
            loss, rgb = _FusedTrainStepFn.apply(self.mlp.embedder_pos.params, self.mlp.density_net.params,
                                                self.mlp.color_net.params, self, data)
        bs = rgb.shape[0]
        if kwargs.get('lazy_log', False):
            # lazy_log contract: the two values are device-side views of the step's recycled buffers and must be read (float())
            # before the step after next runs -- xrnerf_amd.train.Trainer reads them when it logs that very iteration.  A
            # logger that keeps them for longer (an averaging LogBuffer) must not ask for lazy_log.
            log_vars = {'loss': loss.detach(), 'psnr': _LazyPsnr(self._last['loss_mse'], bs)}
        else:
            with torch.no_grad():
                mse_loss = self._last['loss_mse'][1] / (3.0 * bs)        # img2mse of the alpha-masked images
                psnr = mse2psnr(mse_loss)
            log_vars = {'loss': loss.item(), 'psnr': psnr.item()}
        out = {'loss': loss, 'log_vars': log_vars, 'num_samples': bs}
        if direct:
            out['grads_ready'] = True                   # .grad holds the gradients (or nothing: updates_applied) -- no backward() needed
            out['updates_applied'] = applied
        return out

    def train_step(self, data, optimizer, **kwargs):
        for k in data:
            data[k] = unfold_batching(data[k])
        if self._fused_ok():
            self._step_optimizer = optimizer           # (a FusedAdam can hand the table's update to the step, see _FusedTrainStepFn)
            try:
                return self._train_step_fused(data, **kwargs)
            finally:
                self._step_optimizer = None
        ret = self.forward(data, is_test=False)
        bs = ret['rgb'].shape[0]
        alpha = data['alpha'].detach()
        huber_loss = HuberLoss(ret['rgb'], data['target_s'], 0.1, 'sum')
        with torch.no_grad():      # the reference builds (and never uses) an autograd graph for the logged PSNR
            mse_loss = img2mse(ret['rgb'] * alpha, data['target_s'] * alpha)
            psnr = mse2psnr(mse_loss)
        loss = huber_loss * 5
        if kwargs.get('lazy_log', False):
            # same values, read back by the caller when it actually logs (the reference's TextLoggerHook
            # prints every 500 iterations): no per-iteration host synchronisation
            log_vars = {'loss': loss.detach(), 'psnr': psnr.detach()}
        else:
            log_vars = {'loss': loss.item(), 'psnr': psnr.item()}
        return {'loss': loss, 'log_vars': log_vars, 'num_samples': bs}

    def _render_rows(self, frame, is_test=True):
        """batchify_forward of one frame's flattened rays.  One rank: the whole frame, like the reference (which renders on
        rank 0 only, networks/hashnerf.py:58-59,97-98, while the other ranks idle).  Several ranks: every rank marches and
        evaluates ITS contiguous band of image rows and ONE all-gather puts the RGBA image together on every rank (north_star:
        image-space ray sharding + all-gather of the rendered tiles); XRNERF_VAL_RANK0_ONLY=1 restores the reference's form.
        -> {'rgb': [H*W,3], 'alpha': [H*W,1]} of the full frame (None on ranks > 0 in the rank-0-only form)."""
        from . import dist as xdist
        rank, world = get_dist_info()
        if world == 1 or os.environ.get('XRNERF_VAL_RANK0_ONLY') == '1':
            return self.batchify_forward(frame, is_test=is_test) if rank == 0 else None
        shape = tuple(int(v) for v in frame['src_shape'])
        H, W = shape[0], shape[1]
        N = frame[self.bs_data].shape[0]
        assert N == H * W, 'row-band sharding needs the frame as H*W flattened rays'
        row0, nrows = xdist.row_band(H, rank, world)
        band = {k: (v[row0 * W:(row0 + nrows) * W] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == N else v) for k, v in frame.items()}
        # the band's rays draw the march jitter they have in the WHOLE frame's chunk series (frame_ray0: xr_rays_sampler's rng_ray0), and every rank's
        # hidden-generator counter moves on by the whole frame's launches: the pixels are those of the one-GPU / reference frame whatever
        # the world size, and the ranks' training RNG streams stay in step
        from .samplers import NGPGridSampler
        exact = type(self.sampler) is NGPGridSampler and switches.frame_mode() == 'one_launch' and N > self.chunk
        k1_before = getattr(self.sampler, 'k1_calls', 0)
        if exact:
            self.sampler.frame_ray0 = row0 * W
        try:
            ret = self.batchify_forward(band, is_test=is_test)
        finally:
            if exact:
                self.sampler.frame_ray0 = 0
                self.sampler.k1_calls = k1_before + (N + self.chunk - 1) // self.chunk
        tile = torch.cat([ret['rgb'].reshape(nrows, W, 3), ret['alpha'].reshape(nrows, W, 1)], -1)
        img = xdist.gather_image(tile, H, rank, world)
        return {'rgb': img[..., :3].reshape(N, 3), 'alpha': img[..., 3:].reshape(N, 1)}

    def val_step(self, data, optimizer=None, **kwargs):
        if self.phase == 'test':
            return self.test_step(data, **kwargs)
        rank, world_size = get_dist_info()
        sharded = world_size > 1 and os.environ.get('XRNERF_VAL_RANK0_ONLY') != '1'
        if rank != 0 and not sharded:
            return {}
        for k in data:
            data[k] = unfold_batching(data[k])
        poses, images = data['poses'], data['images']
        rgbs, disps, gt_imgs, elapsed_time_list = [], [], [], []
        for i in range(poses.shape[0]):
            start = time.time()
            frame = self.val_pipeline({'pose': poses[i], 'idx': i})
            ret = self._render_rows(frame)
            if rank != 0:
                continue
            rgb = recover_shape(ret['rgb'], frame['src_shape'])
            rgb = rgb.cpu().numpy()           # D2H inside the timer ends the frame, as the reference's does
            elapsed_time_list.append(time.time() - start)
            rgbs.append(rgb)
        if rank != 0:
            return {}
        # the alpha-masked images the reference builds between two frames (networks/hashnerf.py:78-83), built behind the LAST frame
        # instead: ~5 ms of host-side numpy per frame during which the GPU idles and clocks down -- the next frame's timer then read
        # 14 ms for 6 ms of device work (profiles/r04_registry_frame_host_time.txt).  Same arrays.
        for i in range(len(rgbs)):
            img = images[i].cpu().numpy()
            alpha = img[:, :, 3:]
            gt_imgs.append(img[:, :, :3] * alpha)
            rgbs[i] = rgbs[i] * alpha
        return {'rgbs': rgbs, 'disps': disps, 'gt_imgs': gt_imgs, 'elapsed_time': elapsed_time_list}

    def test_step(self, data, **kwargs):
        rank, world_size = get_dist_info()
        sharded = world_size > 1 and os.environ.get('XRNERF_VAL_RANK0_ONLY') != '1'
        if rank != 0 and not sharded:
            return {}
        for k in data:
            data[k] = unfold_batching(data[k])
        idx = data['idx'].item()
        ret = self._render_rows(data)
        if rank != 0:
            return {}
        rgb = recover_shape(ret['rgb'], data['src_shape']).cpu().numpy()
        alpha = recover_shape(ret['alpha'], data['src_shape']).cpu().numpy()
        return {'spiral_rgb': rgb, 'spiral_alpha': alpha, 'idx': idx}

    def set_val_pipeline(self, func):
        self.val_pipeline = func
