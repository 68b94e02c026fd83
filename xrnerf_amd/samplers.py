"""`NGPGridSampler`: occupancy-grid ray marching (Instant-NGP Appendix E), the registered type of
/root/reference/xrnerf/models/samplers/ngp_grid_sampler.py:11-284, on the MI355X kernels.

Same constructor signature, same public attributes (`coords`, `rays_numsteps`,
`rays_numsteps_compacted`, `density_grid_mean`, `aabb_range`, `rgb_activation`,
`density_activation`, `n_rays_per_batch`, registered buffer `density_grid_bitfield`), same schedule
(grid refresh every `update_grid_freq` iterations, rays-per-batch adaptation every 16th).

Deliberate differences, none of which changes a result:
  * no per-kernel device synchronisation (the reference syncs after each of its 4-9 launches); the
    only host read-back per training iteration is the emitted-sample count, as in
    samplers/utils/rays_sampler.py:72;
  * the reference runs the MLP without grad over all S samples only to feed K2, whose transmittance
    loop is dead code (extensions/ngp_raymarch/src/compacted_coord.cu:41-44): K2's outputs do not
    depend on the network output, so that pass is skipped;
  * sample bases are the exclusive prefix sum in ray order (deterministic) instead of atomicAdd
    order, so when nothing is clipped K1's output already IS the compacted layout and K2 is a no-op
    alias (it runs only when S > target_batch_size or K1 overflowed);
  * the occupancy bitfield and the batch size only change at a grid refresh and K1 reads no weights, so the
    marches of the iterations up to the next refresh can be issued right behind a refresh as ONE series of
    launches (`march_window`): same batches, same samples, same RNG call indices as one launch per iteration;
  * the hidden global RNG (`static pcg32 rng{9121}` per translation unit) becomes two explicit call
    counters on this object.
"""
import os

import torch
from torch import nn

from . import _lib, ops
from ._fastattr import _FastAttr
from .builder import SAMPLERS


class _Marched:
    """One iteration of a marched window (NGPGridSampler.march_window), as the sampler's / trainer's queues hold it.  Read like the dict
    it replaces (`pf['out']`, `pf.get('xyz')`), but the tensor views are made when somebody asks: building ~20 views for each of 15
    iterations cost the refresh iteration ~150 us of host time in front of its own march (profiles/r05_trace_refresh_iteration*.txt),
    and the native loop needs none of them."""
    __slots__ = ('window', 'slot', 'n', 'max_samples', 'iter', 'k1_index', 'cur_ray', 'batch_index', 'event', 'gate', 'host_ev')
    _SCALARS = frozenset(__slots__)

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def matches(self, rays_o, max_samples):
        """is `rays_o` this iteration's batch (the chunk's rows of the window, by address)?"""
        w = self.window
        return (rays_o.data_ptr() == w.rays_o.data_ptr() + 12 * self.slot * w.ray_stride and rays_o.shape[0] == self.n
                and self.max_samples == max_samples)

    def batch(self):
        return self.window.batch(self.slot, self.n)

    def __getitem__(self, k):
        if k in self._SCALARS:
            return getattr(self, k)
        w, c, n = self.window, self.slot, self.n
        if k == 'rays_o':
            return w.rays_o[c, :n]
        if k == 'out':
            return (w.coords[c][:self.max_samples], w.rays_index[c][:n], w.numsteps[c][:n], w.counter2[c])
        if k == 'clipped':
            return (w.clipped[c][:n], w.n_valid[c])
        if k == 'xyz':
            return w.xyz[c] if w.xyz is not None else None
        if k == 'host':
            return (self.host_ev, w.pinned[c]) if w.pinned is not None else (None, w.counter2[c])
        raise KeyError(k)

    def get(self, k, default=None):
        try:
            return self[k]
        except KeyError:
            return default

    def __contains__(self, k):
        return k in self._SCALARS or k in ('rays_o', 'out', 'clipped', 'xyz', 'host')


@SAMPLERS.register_module()
class NGPGridSampler(_FastAttr, nn.Module):
    # per-step state (never a Parameter, a registered buffer or a sub-module): assigned ~15 times per training iteration
    _FAST_ATTRS = frozenset(('iter_n', 'on_sampled', 'on_rewind', 'on_refreshed', '_prefetched_q', 'rays_index', 'coords', 'xyz', 'rays_numsteps', 'rays_numsteps_compacted',
                             'n_valid_dev', '_pinned_next', 'k1_calls', '_test_rows_seen',
                             'frame_chunk', 'frame_ray0', '_pending_counts', 'n_rays_per_batch', '_window'))
    def __init__(self, update_grid_freq=16, update_block_size=5000000, n_rays_per_batch=4096,
                 cone_angle_constant=0.00390625, near_distance=0.2, target_batch_size=1 << 18, rgb_activation=2,
                 density_activation=3):
        super().__init__()
        self.update_grid_freq = update_grid_freq
        self.update_block_size = update_block_size
        self.n_rays_per_batch = n_rays_per_batch
        self.target_batch_size = target_batch_size
        self.rgb_activation = rgb_activation
        self.density_activation = density_activation
        self.density_mlp_padded_density_output_width = 1

        n_threads_linear = 128
        self.density_grid_ema_step = 0
        self.NERF_CASCADES = 8
        self.NERF_GRIDSIZE = 128
        # the reference overrides both ctor arguments with the raymarch_shared.h constants
        # (ngp_grid_sampler.py:41,44); kept for result parity
        self.near_distance = 0.05
        self.ema_grid_decay = 0.95
        self.MAX_STEP = 1024
        self.cone_angle_constant = 0.00390625
        self.NERF_MIN_OPTICAL_THICKNESS = 0.01
        self.num_coords_elements = self.n_rays_per_batch * self.MAX_STEP

        g3 = self.NERF_GRIDSIZE ** 3
        self.density_n_elements = self.NERF_CASCADES * g3
        self.density_grid_tmp = torch.zeros([self.density_n_elements], dtype=torch.float32)
        self.density_grid_mean = torch.zeros([(g3 + n_threads_linear - 1) // n_threads_linear], dtype=torch.float32)
        self.register_buffer('density_grid_bitfield', torch.zeros([g3 * self.NERF_CASCADES // 8], dtype=torch.uint8))
        self.measured_batch_size = torch.zeros((1,), dtype=torch.int32)
        self.iter_n = 0
        # explicit replacements of the reference's two hidden per-TU generators
        self.k1_calls = 0
        self.k6_calls = 0
        self.device = None
        self._prefetched_q = []
        self._pending_counts = []
        self.samples_marched = 0

    # ------------------------------------------------------------------ hooks' entry points
    def set_data(self, alldata, datainfo):
        self.resolutions = [datainfo['H'], datainfo['W']]
        self.transforms = torch.as_tensor(alldata['poses'], dtype=torch.float32).contiguous()
        self.focal = torch.as_tensor(alldata['focal'], dtype=torch.float32).contiguous()
        self.aabb_scale = alldata['aabb_scale']
        self.aabb_range = alldata['aabb_range']
        self.metadata = torch.as_tensor(alldata['metadata'], dtype=torch.float32).contiguous()
        self.n_img = alldata['poses'].shape[0]
        self.max_cascade = 0
        while (1 << self.max_cascade) < self.aabb_scale:
            self.max_cascade += 1

    def set_iter(self, iter_n):
        self.iter_n = iter_n

    # ------------------------------------------------------------------ density grid upkeep (K6..K11)
    def update_density_grid_func(self, n_uniform, n_nonuniform, mlp):
        n_elements = self.density_n_elements
        aabb = (float(self.aabb_range[0]), float(self.aabb_range[1]))
        if not hasattr(self, 'density_grid'):
            self.density_grid = ops.mark_untrained_density_grid(self.focal, self.transforms, n_elements,
                                                                self.resolutions)
        planes = hasattr(mlp, 'run_density_planes') and self._streams()
        pre = self.__dict__.pop('_k6_prefetched', None)
        if pre is not None and planes and pre['key'] == (n_uniform, n_nonuniform, self.density_grid_ema_step):
            # K6 and the clear of the temporary grid ran on the side stream during the previous iteration (prefetch_grid_samples)
            pre['event'].wait()
            po, io = self._k6_bufs
            n_tot, n_used = n_uniform + n_nonuniform, pre['n_used']
        else:
            if pre is not None:
                self.k6_calls = pre['k6_calls']                 # a guess that does not apply: as if it had not been made
                pre['event'].wait()                             # ... but its side-stream launches write the buffers this stream is about to write
            po, io, pos_u, idx_u, pos_n, idx_n, n_used = self._grid_samples(n_uniform, n_nonuniform, planes)
            n_tot = n_uniform + n_nonuniform
        if planes:
            positions, indices = po[:, :n_tot], io[:n_tot]
            with torch.no_grad():
                for i in range(0, n_tot, self.update_block_size):
                    pos_i, idx_i = positions[:, i:i + self.update_block_size], indices[i:i + self.update_block_size]
                    # K8 inside the density network's forward where the topology has a fused kernel (same maxima, one launch less)
                    if hasattr(mlp, 'density_splat_planes') and mlp.density_splat_planes(pos_i, idx_i, self.density_grid_tmp):
                        continue
                    density = mlp.run_density_planes(pos_i)
                    ops.splat_grid_samples(density, idx_i, density.stride(0), density.shape[0], self.density_grid_tmp)
        else:
            positions = torch.cat([pos_u, pos_n]) if n_nonuniform > 0 else pos_u
            indices = torch.cat([idx_u, idx_n]) if n_nonuniform > 0 else idx_u
            with torch.no_grad():
                for i in range(0, positions.shape[0], self.update_block_size):
                    density = mlp.run_density(positions[i:i + self.update_block_size])   # [m,1] view, row stride 4
                    ops.splat_grid_samples(density, indices[i:i + self.update_block_size], density.stride(0),
                                           density.shape[0], self.density_grid_tmp)
        # K9, K10, K11 (ngp_grid_sampler.py:150-174: ema_grid_samples_nerf, then update_bitfield) as three launches instead of eleven
        ops.ema_update_bitfield(self.density_grid_tmp, n_used, self.ema_grid_decay, self.density_grid, self.density_grid_mean,
                                self.density_grid_bitfield)
        self.density_grid_ema_step += 1
        if self._streams():
            if getattr(self, '_bitfield_event', None) is None:
                self._bitfield_event = torch.cuda.Event()        # ONE event, recorded again at every refresh (waits capture the record
            self._bitfield_event.record(torch.cuda.current_stream())   # they follow; the native loop's descriptor keeps naming it)

    def _grid_samples(self, n_uniform, n_nonuniform, planes):
        """K6 twice (uniform, then among the occupied cells) + the clear of the temporary grid's part in use"""
        aabb = (float(self.aabb_range[0]), float(self.aabb_range[1]))
        if planes:
            # both K6 calls write ONE [3, n] plane buffer (+ one index vector): no concatenation, coalesced reads in the query
            n_tot = n_uniform + n_nonuniform
            buf = getattr(self, '_k6_bufs', None)
            if buf is None or buf[0].shape[1] < n_tot or buf[0].device != self.density_grid.device:
                buf = self._k6_bufs = (torch.empty((3, n_tot), dtype=torch.float32, device=self.density_grid.device),
                                       torch.empty((n_tot,), dtype=torch.int32, device=self.density_grid.device))
            po, io = buf
        else:
            po = io = None
        pos_u, idx_u = ops.generate_grid_samples(self.density_grid, self.density_grid_ema_step, n_uniform,
                                                 self.max_cascade + 1, -0.01, aabb, self.k6_calls, po, io, 0)
        self.k6_calls += 1
        pos_n, idx_n = ops.generate_grid_samples(self.density_grid, self.density_grid_ema_step, n_nonuniform,
                                                 self.max_cascade + 1, self.NERF_MIN_OPTICAL_THICKNESS, aabb,
                                                 self.k6_calls, po, io, n_uniform)
        self.k6_calls += 1     # the reference's rng advances on every call, also for n == 0
        # K6 draws cells of cascades 0..max_cascade only, so K8 touches, and K9 can change, only that leading part of the two grids:
        # the other cascades' cells hold K7's 0 / -1, which max(p * decay, 0) and the p < 0 rule both leave as they are.  The
        # reference clears and walks all 8 cascades (134 + 200 MB per refresh at aabb_scale 1); same values here from 1/8 of that.
        n_used = (self.max_cascade + 1) * (self.density_n_elements // self.NERF_CASCADES)
        self.density_grid_tmp[:n_used].zero_()
        return po, io, pos_u, idx_u, pos_n, idx_n, n_used

    def _refresh_counts(self, iter_n):
        M = self.NERF_GRIDSIZE ** 3 * (self.max_cascade + 1)
        return (M, 0) if iter_n < 256 else (M // 4, M // 4)

    def prefetch_grid_samples(self, next_iter_n):
        """call inside `with torch.cuda.stream(self.side_stream())` during the iteration BEFORE one that starts with a grid refresh:
        K6 (both calls) and the clear of the temporary grid read the density grid and the RNG call counter only -- neither changes
        until that refresh -- so they leave its critical path (~45 us of the refresh's 0.8 ms).  Same values, same call counters."""
        if not (self._streams() and hasattr(self, 'density_grid')):
            return
        n_uniform, n_nonuniform = self._refresh_counts(next_iter_n)
        side = torch.cuda.current_stream()
        if getattr(self, '_bitfield_event', None) is not None:
            side.wait_event(self._bitfield_event)               # the last refresh (writer of the density grid, reader of the buffers)
        k6_calls = self.k6_calls
        out = self._grid_samples(n_uniform, n_nonuniform, True)
        ev = torch.cuda.Event()
        ev.record(side)
        self._k6_prefetched = dict(key=(n_uniform, n_nonuniform, self.density_grid_ema_step), n_used=out[6], event=ev, k6_calls=k6_calls)

    def update_density_grid(self, mlp):
        self.update_density_grid_func(*self._refresh_counts(self.iter_n), mlp)

    def check_device(self, data):
        device = data['rays_o'].device
        if self.device != device:
            self.device = device
            for attr in ['transforms', 'focal', 'metadata', 'density_grid_mean', 'density_grid_bitfield',
                         'density_grid_tmp']:          # measured_batch_size stays on the host
                if hasattr(self, attr):
                    setattr(self, attr, getattr(self, attr).to(device).contiguous())
            if hasattr(self, 'density_grid'):
                self.density_grid = self.density_grid.to(device)

    # ------------------------------------------------------------------ sampling (K1, K2)
    def sample(self, data, mlp, is_test=False):
        is_training = not is_test
        self.check_device(data)
        k1_reserved = None
        if is_training and (self.iter_n % self.update_grid_freq == 0 or not hasattr(self, 'density_grid')):
            self.update_density_grid(mlp)
            cb = getattr(self, 'on_refreshed', None)
            if cb is not None:
                # the trainer marches the rest of the window from here, beside this iteration's own march in place -- which keeps the
                # hidden generator's call index it has in the reference's order (this iteration first); nothing marched ahead crosses a
                # refresh, so whatever is still queued is stale
                self.rewind_marches()
                k1_reserved = self.k1_calls
                self.k1_calls += 1
                cb()

        rays_o = data['rays_o'].contiguous().float()
        rays_d = data['rays_d'].contiguous().float()
        if 'bg_color' in data:
            data['bg_color'] = data['bg_color'].to(torch.float32).contiguous()
        n_rays = rays_o.shape[0]
        aabb = (float(self.aabb_range[0]), float(self.aabb_range[1]))
        # the reference sizes the buffer n_rays_per_batch*1024 rows (117 MB zero-filled per call,
        # samplers/utils/rays_sampler.py:20-21) whatever the batch; nothing reads rows past the counter, so it is not
        # cleared here, and a batch of more than 65536 rays gets 64 rows per ray (the reference's K1 would silently
        # drop the samples that do not fit; `reference_buffer_rows = True` reproduces exactly that)
        if is_training:
            max_samples = self.train_max_samples(n_rays)
        else:
            # test / render: the worst case is 1024 rows per ray (18 GB for an 800x800 frame marched in one call); the
            # buffer is sized from the previous test launch (first time: 48 rows per ray) and the launch is repeated with
            # the exact size -- same RNG call index, identical result -- in the rare case it overflowed
            est = max(n_rays * 48, int(getattr(self, '_test_rows_seen', 0) * 1.25) + 1024)
            max_samples = min(n_rays * self.MAX_STEP, est)
        # marches issued ahead (march_window: the rest of a refresh window) wait in iteration order.  A test / render launch in between
        # takes them back first: the reference's hidden generator is shared by training and test launches (one `static pcg32` per
        # translation unit), so a frame rendered between iterations j - 1 and j moves the jitter of iteration j on -- the marches of j..
        # are dropped and their RNG / batch counters rewound, and the trainer marches what is left of the window again afterwards
        q = self.__dict__.get('_prefetched_q')
        if q and not is_training:
            self.rewind_marches()
        # (marches for LATER iterations -- issued a moment ago from on_refreshed -- stay queued)
        pf = q.pop(0) if (is_training and q and q[0].get('iter', self.iter_n) <= self.iter_n) else None
        if pf is not None and pf.matches(data['rays_o'], max_samples):
            # K1 of this batch already ran (march_window, right behind the window's grid refresh): order this stream after it -- once
            # per window -- and adopt its outputs
            self._wait_march(pf)
            coords, rays_index, rays_numsteps, counter = pf['out']
            xyz = pf.get('xyz')
            clipped = pf.get('clipped')
            self._pending_counts.append(pf['host'])
            # (K1's outputs and the batch tensors are views of the sampler's persistent window buffers: nothing was allocated on the
            # side stream, nothing has to be handed between the streams' allocator pools)
        else:
            clipped = None
            if pf is not None:
                # a march issued ahead that does not belong to this batch is dropped (its RNG call stays consumed, like
                # any other launch); its side-stream writes into the shared buffers must have finished before this
                # stream's launch touches them
                self._wait_march(pf)
                for other in q:                     # (whatever was marched behind it is as stale as it is)
                    self._wait_march(other)
                del q[:]
            slot = (self.iter_n % self.WINDOW) if is_training else self.TEST_SLOT
            if is_training:
                self.window_for(n_rays, max_samples)
            k1_index = self.k1_calls if k1_reserved is None else k1_reserved
            async_test = (not is_training) and getattr(self, '_async_test', None) is not None and self._streams()
            if async_test:
                # a frame marched in chunks (batchify_forward): no read-back per chunk.  The buffer is sized from the rows per
                # ray the previous frame needed (x1.5; first frame: 48), the valid row count stays on the device, and the
                # chunk's counter is kept for ONE check at the end of the frame (end_async_test: a chunk that overflowed its
                # buffer is marched again with the exact size and the same RNG call index -- identical samples)
                max_samples = min(n_rays * self.MAX_STEP, max(n_rays * 48, int(getattr(self, '_test_rows_per_ray', 0.0) * n_rays * 1.5) + 1024))
            xyz = self._xyz_buffer(max_samples, slot)
            # test mode, `frame_chunk` set by the network: this launch stands for ceil(n_rays / frame_chunk) launches of the
            # reference's chunked frame loop -- same jitter per ray, the hidden generator moves on by as many launches
            rng_chunk = 0 if is_training else int(getattr(self, 'frame_chunk', 0) or 0)
            # ... and `frame_ray0`: these rays are rays frame_ray0.. of the frame (a rank's band of image rows)
            rng_ray0 = int(getattr(self, 'frame_ray0', 0) or 0) if rng_chunk else 0
            coords, rays_index, rays_numsteps, counter = ops.rays_sampler(
                rays_o, rays_d, self.density_grid_bitfield, aabb, self.near_distance, self.cone_angle_constant,
                max_samples, k1_index, coords_out=self._coords_buffer(max_samples, slot),
                small_out=None if async_test else self._small_buffers(n_rays, slot), xyz_out=xyz, rng_chunk=rng_chunk, rng_ray0=rng_ray0,
                # 8 lanes per ray where nothing runs beside the march: the iteration of a grid refresh (every other in-place march may
                # run beside a collective or the K6 prefetch: one ray per lane, same samples)
                wide=is_training and self.iter_n % self.update_grid_freq == 0 and getattr(self, 'wide_in_place', True))
            # (a band: the launches of the chunk series its rays fall into; the network puts the whole frame's count back afterwards)
            if k1_reserved is None:
                self.k1_calls += ((rng_ray0 + n_rays + rng_chunk - 1) // rng_chunk - rng_ray0 // rng_chunk) if rng_chunk else 1
            if async_test:
                self._async_test.append((counter, max_samples, k1_index, n_rays))
                self.rays_index = rays_index
                self.coords = coords[:max_samples]
                self.xyz = xyz
                self.rays_numsteps = rays_numsteps
                data['pts'], data['viewdirs'] = self.coords[..., :3], self.coords[..., 4:]
                data['n_valid_dev'] = counter[1:2]            # rows behind it are stale; an overflowing count is clamped by the kernels
                if xyz is not None:
                    data['pts_planes'] = xyz[:, :max_samples]
                return data
            if not is_training:
                n_valid, samples = counter.tolist()      # one host read-back per call (rays_sampler.py:72)
                if samples > max_samples:
                    max_samples = min(n_rays * self.MAX_STEP, samples)
                    xyz = self._xyz_buffer(max_samples, slot)
                    coords, rays_index, rays_numsteps, counter = ops.rays_sampler(
                        rays_o, rays_d, self.density_grid_bitfield, aabb, self.near_distance, self.cone_angle_constant,
                        max_samples, k1_index, coords_out=self._coords_buffer(max_samples, slot),
                        small_out=self._small_buffers(n_rays, slot), xyz_out=xyz, rng_chunk=rng_chunk, rng_ray0=rng_ray0)
                    n_valid, samples = counter.tolist()
                self._test_rows_seen = samples
            elif self._streams():
                # the counter's device-to-host copy goes to the side stream: on the compute stream its
                # system-scope completion would sit between K1 and the first kernel that uses K1's output
                e = torch.cuda.Event()
                e.record(torch.cuda.current_stream())
                side = self.side_stream()
                with torch.cuda.stream(side):
                    side.wait_event(e)
                    self._pending_counts.append(self._count_to_host(counter))
            else:
                self._pending_counts.append((None, counter.clone()))       # host tensors (kernels under tests/hip_emu)
        self.rays_index = rays_index
        if not is_training:
            coords = coords[:min(samples, max_samples)]
            self.coords = coords
            self.xyz = xyz
            self.rays_numsteps = rays_numsteps
            data['pts'], data['viewdirs'] = coords[..., :3], coords[..., 4:]
            if xyz is not None:
                data['pts_planes'] = xyz[:, :coords.shape[0]]      # the same positions as planes [3, n] (K1 wrote both)
            return data

        # K2.  K1's bases are the ray-ordered prefix sums, i.e. exactly what K2 would assign, so the
        # compacted coordinates are the first min(S, target) rows of K1's buffer IN PLACE and only the
        # clipped per-ray counts have to be produced.  Like the reference, the sample buffer handed to the
        # MLP has a fixed target_batch_size rows (compacted_coords.py:20-21 pads with zeros); the number of
        # valid rows stays on the device (`n_valid_dev`) so that no host read-back is needed here.
        if clipped is None:
            clipped = ops.clip_numsteps(rays_numsteps, counter, self.target_batch_size, out=self._clip_buffers(n_rays, slot))
        rays_numsteps_compacted, n_valid_dev = clipped
        # the pre-clip counter the reference accumulates in `measured_batch_size` (ngp_grid_sampler.py:252) is
        # accumulated on the host from the asynchronous pinned copies (_pending_counts)
        self.update_batch_rays(is_training, max_samples)
        coords_compacted = coords[:self.target_batch_size]
        self.coords = coords_compacted
        self.xyz = xyz
        if xyz is not None:
            data['pts_planes'] = xyz[:, :coords_compacted.shape[0]]
        self.rays_numsteps = rays_numsteps
        self.rays_numsteps_compacted = rays_numsteps_compacted
        self.n_valid_dev = n_valid_dev[0:1]
        data['pts'], data['viewdirs'] = coords_compacted[..., :3], coords_compacted[..., 4:]
        data['n_valid_dev'] = n_valid_dev[0:1]
        cb = getattr(self, 'on_sampled', None)
        if cb is not None:
            cb()       # e.g. the trainer issues the NEXT batch's march on a side stream right here
        return data

    # K1's output buffers are persistent and owned by the sampler.  Training: the XR_NGP_WINDOW chunks of ONE window allocation
    # (ops.MarchWindow) -- iteration `it` lives in chunk it % WINDOW, whether it was marched ahead with the rest of its refresh window
    # (march_window) or in place.  Test / render launches, which may come in between, have a slot of their own.
    WINDOW = ops.MarchWindow.N               # = XR_NGP_WINDOW (include/xrnerf_mi355.h)
    TEST_SLOT = WINDOW

    def train_max_samples(self, n_rays):
        """rows of a training launch's sample buffer.  The reference sizes it n_rays_per_batch(ctor) * 1024 rows whatever the batch
        (117 MB zero-filled per call, samplers/utils/rays_sampler.py:20-21); nothing reads rows past the counter, so it is not cleared
        here, and a batch of more than 65536 rays gets 64 rows per ray (the reference's K1 would silently drop the samples that do not
        fit; `reference_buffer_rows = True` reproduces exactly that)"""
        m = self.num_coords_elements if getattr(self, 'reference_buffer_rows', False) else max(self.num_coords_elements, n_rays * 64)
        return min(m, n_rays * self.MAX_STEP)

    def window_for(self, n_rays, max_samples):
        """the window allocation, grown when a batch size / row capacity does not fit (marches issued ahead keep their old views)"""
        w = self.__dict__.get('_window')
        if w is None or w.ray_stride < n_rays or w.coords_stride < max_samples or w.device != self.device:
            w = self._window = ops.MarchWindow(self.device, max((n_rays + 127) // 128 * 128, 1 << 15), max_samples, planes=self._streams())
        return w

    def _coords_buffer(self, rows, slot):
        if slot < self.WINDOW:
            return self._window.coords[slot][:rows]
        buf = getattr(self, '_test_coords', None)
        if buf is None or buf.shape[0] < rows or buf.device != self.device:
            buf = self._test_coords = torch.empty((rows, 7), dtype=torch.float32, device=self.device)
        return buf[:rows]

    def begin_async_test(self):
        """batchify_forward(is_test=True) over several chunks: no host read-back per chunk until end_async_test"""
        self._async_test = []

    def end_async_test(self):
        """one read-back for the whole frame -> [(chunk index, K1 call index)] of the chunks whose samples did not fit their
        buffer (the caller renders those again through the synchronous path)"""
        pend, self._async_test = self._async_test, None
        if not pend:
            return []
        counts = torch.stack([c for c, _, _, _ in pend]).cpu().numpy()
        self._test_rows_per_ray = max(float(counts[i, 1]) / max(pend[i][3], 1) for i in range(len(pend)))
        return [(i, pend[i][2]) for i in range(len(pend)) if int(counts[i, 1]) > pend[i][1]]

    def _xyz_buffer(self, rows, slot):
        """[3, rows] planes holding the positions of the slot's coordinate rows once more (K1 writes both): the encoder's
        coalesced input.  Device only (the host build of the kernels reads the rows)."""
        if not self._streams():
            return None
        if slot < self.WINDOW:
            return self._window.xyz[slot]
        buf = getattr(self, '_test_xyz', None)
        if buf is None or buf.shape[1] < rows or buf.device != self.device:
            buf = self._test_xyz = torch.empty((3, rows), dtype=torch.float32, device=self.device)
        return buf

    def _small_buffers(self, n_rays, slot):
        if slot < self.WINDOW:
            w = self._window
            return w.rays_index[slot][:n_rays], w.numsteps[slot][:n_rays], w.counter2[slot]
        b = getattr(self, '_test_small', None)
        if b is None or b[0].shape[0] < n_rays or b[0].device != self.device:
            cap = max(n_rays, 1 << 15)
            b = self._test_small = (torch.empty((cap, 1), dtype=torch.int32, device=self.device),
                                    torch.empty((cap, 2), dtype=torch.int32, device=self.device),
                                    torch.empty((2,), dtype=torch.int32, device=self.device))
        return b[0][:n_rays], b[1][:n_rays], b[2]

    def _clip_buffers(self, n_rays, slot):
        """persistent outputs of K2's clip"""
        w = self._window
        return w.clipped[slot][:n_rays], w.n_valid[slot]

    # ------------------------------------------------------------------ the window's marches
    def can_march_ahead(self, next_iter):
        """K1 depends on the rays and the bitfield only -- not on the parameters -- so an iteration can be marched ahead of its step
        unless it refreshes the occupancy grid first"""
        return hasattr(self, 'density_grid') and next_iter % self.update_grid_freq != 0

    def _streams(self):
        """side streams / events / pinned staging only exist on the device (the kernels' host build under tests/hip_emu
        runs everything in program order)"""
        return self.device is not None and self.device.type == 'cuda'

    def side_stream(self):
        if getattr(self, '_side', None) is None:
            # (a high- or a low-priority stream changes nothing -- 0.439 ms/step at priority -1, 0, +1, round 4 -- and a CU-masked stream doubles the
            # iteration: profiles/r04_side_stream_cu_mask_ab.txt)
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _wait_march(self, pf):
        """order the current stream behind the march `pf` came from -- once per series of marches (they share one event)"""
        ev, gate = pf.get('event'), pf.get('gate')
        if ev is not None and not (gate is not None and gate['waited']):
            ev.wait()                                   # the current stream waits (no torch.cuda.current_stream(): ~10 us of Python)
            if gate is not None:
                gate['waited'] = True

    def march_window(self, rows_table, cur_ray, batch_call_index, first_iter, n_iters, n_rays, batches_ready=0, on_side=True):
        """Draw and march iterations first_iter .. first_iter + n_iters - 1 -- the rest of a refresh window: same bitfield, same batch
        size (ngp_grid_sampler.py:194-197,268-281) -- as ONE series of launches (xr_ngp_window_march).  rows_table = the device-resident
        [N, 11] ray table, cur_ray / batch_call_index = HashBatchSample's cursor and the batch generator's call index for the first
        batch drawn here (the first `batches_ready` iterations already hold their batch in their chunk).  on_side: on the side stream,
        behind the last refresh (beside the refresh iteration's own step); else on the current stream.
        -> (cursor behind the last batch, [the queued entry per iteration: `.batch()` gives its batch dict])"""
        n = int(n_rays)
        max_samples = self.train_max_samples(n)
        win = self.window_for(n, max_samples)
        W = self.WINDOW
        c0 = first_iter % W
        if n_iters < 1 or c0 + n_iters > W:
            raise ValueError('iterations %d..%d are not inside one window of %d' % (first_iter, first_iter + n_iters - 1, W))
        aabb = (float(self.aabb_range[0]), float(self.aabb_range[1]))
        k1_0 = self.k1_calls
        # the cursor of every batch drawn here (what a rewind puts back: rewind_marches)
        cursors, cur = [], int(cur_ray)
        for j in range(n_iters):
            if j >= batches_ready and cur + n > rows_table.shape[0]:
                cur = 0
            cursors.append(cur)
            if j >= batches_ready:
                cur += n
        on_side = bool(on_side and self._streams())
        ev = host_ev = None
        if on_side:
            side, main_ev = self.side_stream(), getattr(self, '_bitfield_event', None)
            if main_ev is None:
                main_ev = torch.cuda.Event()
                main_ev.record()
            with torch.cuda.stream(side):
                # behind the last refresh: the last writer of the bitfield, and behind every step of the window before (the last readers
                # of the chunks this series overwrites)
                side.wait_event(main_ev)
                end = ops.ngp_window_march(win, c0, n_iters, batches_ready, n, rows_table, cur_ray, batch_call_index, self.density_grid_bitfield,
                                           aabb, self.near_distance, self.cone_angle_constant, max_samples, k1_0, self.target_batch_size)
                ev = torch.cuda.Event()
                ev.record(side)
                host_ev = ev
        else:
            end = ops.ngp_window_march(win, c0, n_iters, batches_ready, n, rows_table, cur_ray, batch_call_index, self.density_grid_bitfield,
                                       aabb, self.near_distance, self.cone_angle_constant, max_samples, k1_0, self.target_batch_size)
            if self._streams():
                host_ev = torch.cuda.Event()
                host_ev.record()
        assert end == cur
        self.k1_calls = k1_0 + n_iters
        gate = {'waited': ev is None}
        q = self.__dict__.setdefault('_prefetched_q', [])
        made = [_Marched(window=win, slot=c0 + j, n=n, max_samples=max_samples, iter=first_iter + j, k1_index=k1_0 + j, cur_ray=cursors[j],
                         batch_index=batch_call_index + max(j - batches_ready, 0), event=ev, gate=gate, host_ev=host_ev) for j in range(n_iters)]
        q.extend(made)
        return end, made

    def rewind_marches(self):
        """Take back the marches issued ahead and not consumed yet: the hidden generator's call counter goes back to the first of them
        (`on_rewind(entry)` lets the owner of the batch cursor do the same), the current stream is ordered behind their launches."""
        q = self.__dict__.get('_prefetched_q')
        if not q:
            return
        for pf in q:
            self._wait_march(pf)
        first = q[0]
        if 'k1_index' in first:
            self.k1_calls = first['k1_index']
        cb = getattr(self, 'on_rewind', None)
        del q[:]
        if cb is not None:
            cb(first)

    def _count_to_host(self, counter):
        """asynchronous copy of K1's (rays, samples) counter to pinned host memory, right behind the launch: by
        the time the batch-size feedback needs the numbers (every 16th iteration) they have long arrived, so
        reading them does not drain the compute stream the way `measured_batch_size.item()` does"""
        # pinned staging buffers come from a small ring allocated once: a fresh pin_memory allocation per iteration
        # goes through the host allocator (and, on a miss, hipHostMalloc)
        ring = getattr(self, '_pinned_ring', None)
        if ring is None:
            ring = self._pinned_ring = [torch.empty((2,), dtype=torch.int32, pin_memory=True) for _ in range(64)]
            self._pinned_next = 0
        host = ring[self._pinned_next % len(ring)]
        self._pinned_next += 1
        host.copy_(counter, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()                                  # on the current stream
        return ev, host

    def _drain_counts(self):
        """fold the arrived per-iteration counters into `measured_batch_size` (a HOST tensor here; the reference
        keeps it on the device and pays a stream drain to read it, ngp_grid_sampler.py:252,271)"""
        for ev, host in self._pending_counts:
            if ev is not None:
                ev.synchronize()
            c = int(host[1])
            self.measured_batch_size += c
            self.samples_marched += min(c, self.target_batch_size)      # rows that actually went through the MLP
        self._pending_counts = []

    def total_valid_samples(self):
        """samples evaluated by training steps so far (host-side statistic; waits only for counter copies)"""
        self._drain_counts()
        return self.samples_marched

    def poll_mlp_range(self):
        """The default MLP arithmetic (XR_MLP_F16X2) saturates operands above fp16's range and counts them in a device word
        (ops.mlp_range_word).  Once per refresh window, where the host waits for the sample counters anyway: look at the copy of the
        word issued one window ago (arrived long since -- no stall), warn the first time it is not zero, issue the next copy.
        `mlp_range_events` = the count as of one window ago."""
        if not self._streams():
            return
        word = ops.mlp_range_word(self.device)
        pend = self.__dict__.get('_range_pending')
        if pend is not None:
            ev, host = pend
            ev.synchronize()
            v = int(host[0])
            if v and not getattr(self, 'mlp_range_events', 0):
                import warnings
                warnings.warn('the fused MLP met operands above the fp16 range (%d wave events so far): they were saturated at 65504; '
                              'ops.set_f32_forward("mfma") / XR_MLP_BWD_DW=b2x run the fp32 MFMA kernels instead' % v)
            self.mlp_range_events = v
        host = self.__dict__.get('_range_host')
        if host is None:
            host = self._range_host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        host.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._range_pending = (ev, host)

    def update_batch_rays(self, is_training, max_samples=None):
        if is_training and self.iter_n % self.update_grid_freq == (self.update_grid_freq - 1):
            self._drain_counts()
            self.poll_mlp_range()
            total = int(self.measured_batch_size)                # host tensor: no device read-back (:271)
            if max_samples is not None and total > 16 * max_samples:
                raise RuntimeError('ray marcher overflowed its %d-row sample buffer' % max_samples)
            measured_batch_size = max(total / 16, 1)
            rays_per_batch = int(self.n_rays_per_batch * self.target_batch_size / measured_batch_size)
            self.n_rays_per_batch = int(min(self.div_round_up(int(rays_per_batch), 128) * 128, self.target_batch_size))
            self.measured_batch_size.zero_()

    def div_round_up(self, val, divisor):
        return (val + divisor - 1) // divisor
