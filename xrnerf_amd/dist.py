"""Multi-GPU: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU protocol tests).

The Instant-NGP path shards over RAYS:
  * training  = data-parallel replicas, each rank marches its own rays; the one real exchange per
    iteration is the gradient reduction (12.2 M hash-grid + 10 K MLP floats).  The reference gets
    this implicitly from MMDistributedDataParallel (core/apis/train.py:28-36); here it is one
    explicit flat-bucket all-reduce per parameter tensor (3 collectives, the hash-grid one 48.8 MB:
    few, large messages suit xGMI's point-to-point links).
  * rendering = image-space shard: rank r renders a contiguous band of rows, then ONE all-gather of
    the RGBA tiles (the reference renders on rank 0 only, networks/hashnerf.py:58-59,97-98).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl' and torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def allreduce_grads(params, world_size):
    """average gradients across ranks (DDP semantics), one collective per parameter tensor"""
    if world_size <= 1:
        return
    works = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True) for p in params]
    for w in works:
        w.wait()
    for p in params:
        p.grad.mul_(1.0 / world_size)


class _ExposedTimer:
    """How long the compute stream WAITS for the collectives of a step: one event in front of the waits and one behind them, on the
    compute stream -- the span between the two is what the exchange adds to the step after everything that overlapped (measured,
    per rank; `dist.comm_model` is the model beside it).  Host tensors (gloo protocol tests on CPU): wall clock around the waits.
    A ring of RING event pairs, re-recorded in turn (like the native exchange's, csrc/xr_dist.hip): the record covers the last RING
    steps, and the number of live timing events stays bounded (round 5: with several hundred timing events alive, event records took
    ~100 us each -- profiles/NOTES in DESIGN.md section 0, item 8)."""

    RING = 64

    def __init__(self):
        self.on = False
        self._pairs, self._host_ms = [], []
        self._n = 0

    def begin(self, device_is_cuda):
        if not self.on:
            return None
        if device_is_cuda:
            slot = self._n % self.RING
            if slot >= len(self._pairs):
                self._pairs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            a = self._pairs[slot][0]
            a.record()
            return slot
        import time
        return time.perf_counter()

    def end(self, token):
        if token is None:
            return
        if isinstance(token, float):
            import time
            self._host_ms.append((time.perf_counter() - token) * 1e3)
        else:
            self._pairs[token][1].record()
            self._n += 1

    def summary(self):
        """-> {'steps', 'mean_ms', 'max_ms'} (call after a device synchronisation) over the last <= RING device-side steps (all host-side
        ones); 'steps' counts every step since the last call; clears the record"""
        k = min(self._n, len(self._pairs))
        ms = [a.elapsed_time(b) for a, b in self._pairs[:k]] + self._host_ms
        steps = self._n + len(self._host_ms)
        self._host_ms, self._n = [], 0
        return {'steps': steps, 'mean_ms': sum(ms) / len(ms) if ms else None, 'max_ms': max(ms) if ms else None}


class BucketedGradSync:
    """Gradient reduction overlapped with the backward pass that produces the gradients.

    The fused training step hands over each gradient bucket as soon as the kernels writing it are enqueued
    (`ready`): MLP gradients after the MLP backward, the fine half of the hash-grid gradient (levels 8-15,
    32 MB) after its scatter launch -- its all-reduce then runs on RCCL's stream under the scatter of the
    coarse half -- and the coarse half (16.8 MB) last.  `finish` orders the compute stream after all of
    them and returns the factor (1/world) the caller folds into its own gradient scaling.  Three
    collectives per iteration, 10 KB + 32 MB + 16.8 MB: few and large, as xGMI's point-to-point rings want."""

    def __init__(self, world_size, wire_dtype=None):
        """wire_dtype=torch.bfloat16 (XRNERF_DP=allreduce_bf16): buckets above 1 MB cross the links as bf16 -- half the bytes of the
        48.8-MB table gradient; the reference's tcnn gradients are fp16 (hashnerf_mlp.py:76-77 casts its half outputs up).  Each rank
        rounds its gradient to bf16 (2^-9 relative), the sum is taken in bf16 by the collective and widened back to fp32 in place:
        the replicas stay bit-identical (every rank gets the same sum), the trajectory differs from the fp32 exchange by that rounding
        (tests/test_capi_and_host.py::test_bf16_gradient_exchange_two_ranks bounds it).
        NOT a parity mode: tcnn's fp16 gradients are accumulated in fp32 on one GPU and the reference's DDP sums fp32 across ranks, while
        here the collective itself adds in bf16 -- each rank's rounding plus up to world - 1 sequential roundings on a ring, 2^-8 (half an
        ulp of bf16) each, so the error of the sum grows with the world size (test_bf16_wire_sum_error_by_world_size: <= world * 2^-8 of
        sum|g|, i.e. 2^-7 at 2 ranks, 2^-5 at 8).  It is an
        opt-in bandwidth trade on the per-iteration path only; the native loop serves the two exact exchanges."""
        self.world_size = int(world_size)
        self.wire_dtype = wire_dtype
        self._works, self._staged = [], []
        self.exposed = _ExposedTimer()
        self.bytes_on_wire = 0

    def ready(self, bucket):
        if self.world_size > 1:
            # `.data`: the bucket is a slice of a buffer other outputs of the fused step are views of (the loss scalar);
            # the in-place reduction must not bump THEIR autograd version counter ("a view ... has been modified inplace")
            b = bucket.data
            self._on_device = b.is_cuda
            if self.wire_dtype is not None and b.numel() * 4 >= (1 << 20):
                wire = b.to(self.wire_dtype)
                self._works.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True))
                self._staged.append((b, wire))
                self.bytes_on_wire += wire.numel() * wire.element_size()
            else:
                self._works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))
                self.bytes_on_wire += 4 * b.numel()

    def finish(self):
        tok = self.exposed.begin(self._on_device) if self._works else None
        for w in self._works:
            w.wait()
        self.exposed.end(tok)
        for b, wire in self._staged:
            b.copy_(wire)                         # widened back in place: .grad holds the (bf16-rounded) sum as fp32
        self._works, self._staged = [], []
        return 1.0 / self.world_size

    _on_device = False


class Zero1GradSync:
    """SURVEY.md section 8e's gradient exchange as it says it: reduce-scatter -> fused Adam on the rank's shard -> all-gather of
    the updated parameters (the reference's equivalent is DDP's all-reduce + a replicated optimiser, core/apis/train.py:28-38).
    Selected with XRNERF_DP=zero1.  Same bytes on the links as the all-reduce (a ring all-reduce IS reduce-scatter + all-gather),
    but the 71-us Adam + EMA pass over the 12.2 M table parameters -- 12 % of a step, run N times over by the replicated form
    -- shrinks to 1/N per rank, and so do its m / v / EMA states.

    Geometry: the table's storage is padded to `world * shard` floats (shard a multiple of 4: 16-byte aligned for the fused
    Adam's float4 accesses); rank r owns [r * shard, (r + 1) * shard).  `attach` re-seats the parameter (and `pad_grad` the
    step's gradient buffers) on padded storage once, so the collectives run in place: no staging copies.
    Protocol per step, same interface as BucketedGradSync: `ready(mlp gradients)` -> async all-reduce (41 KB, replicated
    Adam); `ready(table gradient)` -> async reduce-scatter into `shard_grad`; `finish()` waits and returns 1 / world;
    the trainer then steps its optimiser on (`shard_param`, wd, wc) and calls `gather_params()`."""

    split_levels = False           # one table bucket: shards cut across the level slices

    def __init__(self, world_size, rank):
        self.world_size, self.rank = int(world_size), int(rank)
        self._works = []
        self.shard = self.n = 0
        self.bytes_reduced = self.bytes_gathered = 0
        self.exposed = _ExposedTimer()
        self._on_device = False

    def attach(self, table_param):
        """pad the table parameter's storage to world * shard floats, in place -> the nn.Parameter of this rank's shard (a view)"""
        n = table_param.numel()
        self.n = n
        self.shard = (-(-n // self.world_size) + 3) // 4 * 4
        buf = torch.zeros(self.shard * self.world_size, dtype=table_param.dtype, device=table_param.device)
        buf[:n] = table_param.data
        table_param.data = buf[:n]
        self.param_padded = buf
        self.shard_param = torch.nn.Parameter(buf[self.rank * self.shard:(self.rank + 1) * self.shard])
        self.shard_grad = torch.zeros_like(self.shard_param.data)
        return self.shard_param

    def pad_grad(self, device):
        """storage for one table-gradient buffer of the fused step: (padded [world * shard], its leading [n] view).  Nothing is kept
        here: `ready` finds the padded extent through the view's own storage, so a buffer set the step replaces is freed with it."""
        buf = torch.zeros(self.shard * self.world_size, dtype=torch.float32, device=device)
        return buf, buf[:self.n]

    def _padded_of(self, bucket):
        """the whole padded buffer behind a table-gradient view made by pad_grad (None: not one of those)"""
        total = self.shard * self.world_size
        if bucket.numel() != self.n or bucket.storage_offset() != 0 or bucket.dtype != torch.float32:
            return None
        st = bucket.untyped_storage()
        if st.nbytes() < 4 * total:
            return None
        return torch.empty(0, dtype=torch.float32, device=bucket.device).set_(st, 0, (total,))

    def ready(self, bucket):
        if self.world_size <= 1:
            return
        self._on_device = bucket.is_cuda
        padded = self._padded_of(bucket)
        if padded is None:                                   # the MLP gradients: replicated
            self._works.append(dist.all_reduce(bucket.data, op=dist.ReduceOp.SUM, async_op=True))
            self.bytes_reduced += 4 * bucket.numel()
            return
        self.bytes_reduced += 4 * padded.numel()
        if dist.get_backend() == 'gloo':                     # gloo has no reduce-scatter: all-reduce, keep this rank's shard
            self._works.append(dist.all_reduce(padded, op=dist.ReduceOp.SUM, async_op=True))
            self._take = padded
        else:
            self._works.append(dist.reduce_scatter_tensor(self.shard_grad, padded, op=dist.ReduceOp.SUM, async_op=True))
            self._take = None

    def finish(self):
        tok = self.exposed.begin(self._on_device) if self._works else None
        for w in self._works:
            w.wait()
        self.exposed.end(tok)
        self._works = []
        if getattr(self, '_take', None) is not None:
            self.shard_grad.copy_(self._take[self.rank * self.shard:(self.rank + 1) * self.shard])
            self._take = None
        self.shard_param.grad = self.shard_grad
        return 1.0 / self.world_size

    def gather_params(self):
        """all-gather of the updated shards, in place in the padded parameter storage"""
        if self.world_size <= 1:
            return
        self.bytes_gathered += 4 * self.param_padded.numel()
        if dist.get_backend() == 'gloo':
            parts = [torch.empty_like(self.shard_param.data) for _ in range(self.world_size)]
            dist.all_gather(parts, self.shard_param.data.contiguous())
            for r, t in enumerate(parts):
                if r != self.rank:
                    self.param_padded[r * self.shard:(r + 1) * self.shard].copy_(t)
        else:
            dist.all_gather_into_tensor(self.param_padded, self.shard_param.data)


class CallbackExchange:
    """xr_grad_exchange (include/xrnerf_mi355.h) served by torch.distributed through ctypes callbacks: the native loop calls back for
    each collective (~30 us of interpreter time each instead of the ~360 us of a whole Python-driven iteration).  Any backend: this is
    what the gloo tests use (two ranks sharing one GPU); RCCL jobs take `RcclExchange`, which never enters the interpreter.
    Buffers are found by address among the registered base tensors (the native loop hands slices of them)."""

    def __init__(self, world_size, rank):
        import ctypes as C
        from . import _lib
        self.world_size, self.rank = int(world_size), int(rank)
        self._bases, self._works = [], []
        self.exposed = _ExposedTimer()
        self._on_device = False
        self.error = None

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:      # noqa: BLE001  (an exception must not unwind through the C frames)
                    self.error = e
                    return -5
            return run

        def all_reduce(ctx, buf, n, stream):
            t = self._view(buf, n)
            self._on_device = t.is_cuda
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

        def reduce_scatter(ctx, send, recv, n_recv, stream):
            src, dst = self._view(send, n_recv * self.world_size), self._view(recv, n_recv)
            self._on_device = dst.is_cuda
            if dist.get_backend() == 'gloo':                 # gloo has no reduce-scatter: all-reduce, keep this rank's shard
                dist.all_reduce(src, op=dist.ReduceOp.SUM)
                dst.copy_(src[self.rank * n_recv:(self.rank + 1) * n_recv])
            else:
                self._works.append(dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, async_op=True))

        def all_gather(ctx, send, recv, n_send, stream):
            src, dst = self._view(send, n_send), self._view(recv, n_send * self.world_size)
            if dist.get_backend() == 'gloo':
                parts = [torch.empty_like(src) for _ in range(self.world_size)]
                dist.all_gather(parts, src.contiguous())
                for r, t in enumerate(parts):
                    if r != self.rank:
                        dst[r * n_send:(r + 1) * n_send].copy_(t)
            else:
                self._works.append(dist.all_gather_into_tensor(dst, src, async_op=True))

        def finish(ctx, stream):
            tok = self.exposed.begin(self._on_device) if self._works else None
            for w in self._works:
                w.wait()
            self.exposed.end(tok)
            self._works = []

        self._cbs = (_lib.EX_ALL_REDUCE(guard(all_reduce)), _lib.EX_SCATTER_GATHER(guard(reduce_scatter)),
                     _lib.EX_SCATTER_GATHER(guard(all_gather)), _lib.EX_FINISH(guard(finish)))
        self.c = _lib.GradExchange(self._cbs[0], self._cbs[1], self._cbs[2], self._cbs[3], None, self.world_size, self.rank)

    def register(self, *tensors):
        """base tensors the loop's buckets are slices of (kept alive here)"""
        for t in tensors:
            if t is not None and not any(b.data_ptr() == t.data_ptr() and b.numel() >= t.numel() for b in self._bases):
                self._bases.append(t.detach().reshape(-1) if t.is_contiguous() else t)

    def _view(self, ptr, n):
        ptr, n = int(ptr), int(n)
        for b in self._bases:
            off = ptr - b.data_ptr()
            if 0 <= off and off + 4 * n <= 4 * b.numel() and off % 4 == 0:
                return b[off // 4:off // 4 + n]
        raise RuntimeError('the native loop handed the exchange a buffer that was not registered (%#x, %d floats)' % (ptr, n))


class RcclExchange:
    """xr_grad_exchange served by RCCL driven from native code (csrc/xr_dist.hip): the communicator is created from a unique id made on
    rank 0 and broadcast through torch.distributed's existing process group; the collectives then never enter the interpreter."""

    def __init__(self, world_size, rank):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        self.world_size, self.rank = int(world_size), int(rank)
        lib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')          # the copy torch.distributed itself uses
        path = lib.encode() if os.path.exists(lib) else None
        uid = (C.c_char * 128)()
        # rank 0 makes the id; a failure there is broadcast too (an empty id), so that no rank is left waiting in a collective
        box = [None]
        if self.rank == 0:
            rc = L.xr_rccl_unique_id(path, uid)
            box = [bytes(uid.raw) if rc == 0 else b'']
            self._id_error = None if rc == 0 else L.xr_last_error().decode()
        if self.world_size > 1:
            dist.broadcast_object_list(box, src=0)
        if not box[0]:
            raise _lib.XrError('xr_rccl_unique_id failed on rank 0%s' % (': ' + self._id_error if self.rank == 0 else ''))
        uid = (C.c_char * 128).from_buffer_copy(box[0])
        self.h = L.xr_rccl_create(path, uid, self.world_size, self.rank)
        if not self.h:
            raise _lib.XrError('xr_rccl_create failed: %s' % L.xr_last_error().decode())
        self.c = _lib.GradExchange()
        _lib.check(L.xr_rccl_exchange(self.h, C.byref(self.c)), 'xr_rccl_exchange')
        self.exposed = self                         # (same two calls as _ExposedTimer: `on`, `summary()`; measured in native code)
        self._on = False

    def register(self, *tensors):
        pass

    error = None

    @property
    def on(self):
        return self._on

    @on.setter
    def on(self, v):
        from . import _lib
        self._on = bool(v)
        _lib.check(_lib.load().xr_rccl_exposed_ms(self.h, 1 if v else 0, None, None, None), 'xr_rccl_exposed_ms')

    def summary(self):
        """-> {'steps', 'mean_ms', 'max_ms'} of the waits the compute stream spent in `finish` (call after a device synchronisation);
        clears the record"""
        import ctypes as C
        from . import _lib
        mean, mx, n = C.c_float(), C.c_float(), C.c_int()
        _lib.check(_lib.load().xr_rccl_exposed_ms(self.h, 1 if self._on else 0, C.byref(mean), C.byref(mx), C.byref(n)), 'xr_rccl_exposed_ms')
        return {'steps': int(n.value), 'mean_ms': float(mean.value) if n.value else None, 'max_ms': float(mx.value) if n.value else None}

    def __del__(self):
        try:
            from . import _lib
            _lib.load().xr_rccl_destroy(self.h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def native_exchange(world_size, rank, prefer_rccl=None):
    """the exchange implementation of the native loop for this job: RCCL from native code under backend 'nccl' (prefer_rccl: force the
    attempt / skip it), callbacks into torch.distributed otherwise (gloo).  The ranks AGREE on the outcome: if the native communicator
    cannot be made on any rank (librccl not loadable, ncclCommInitRank failing), every rank takes the callback form -- the same RCCL
    collectives issued by torch.distributed, ~30 us of interpreter time each -- and says so (`fallback_reason`, echoed in bench.py's
    `collective` block).  Nothing here computes on the host."""
    if prefer_rccl is None:
        prefer_rccl = dist.get_backend() == 'nccl'
    if not prefer_rccl:
        return CallbackExchange(world_size, rank)
    ex, err = None, None
    try:
        ex = RcclExchange(world_size, rank)
    except Exception as e:  # noqa: BLE001  (whatever went wrong, the other ranks must hear of it)
        err = e
    if world_size > 1:
        flag = torch.tensor([0 if ex is not None else 1], dtype=torch.int32,
                            device=torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(flag)
        failed = int(flag[0])
    else:
        failed = 0 if ex is not None else 1
    if failed == 0:
        return ex
    import warnings
    reason = 'the native RCCL exchange could not be created on %d of %d ranks%s' % (failed, world_size, ' (here: %s)' % err if err is not None else '')
    warnings.warn(reason + ': the native loop calls back into torch.distributed for its collectives')
    ex = CallbackExchange(world_size, rank)
    ex.fallback_reason = reason
    return ex


def comm_model(world_size, table_floats=12196240, mlp_floats=10240, link_GBs=153.0, links=7, step_ms=0.50, wire_bytes_per_float=4.0):
    """what one training step puts on xGMI, and what it costs under two schedules (one-GPU boxes only: nothing here is measured).
    xGMI is point-to-point, 7 links x ~153 GB/s per GPU.
      ring:   an N-rank ring is bound by ONE link per hop: all-reduce = 2 (N - 1) / N x bytes over one link;
      direct: on a fully connected mesh (N <= links + 1) reduce-scatter and all-gather each send bytes / N to every peer over its
              own link at the same time: 2 x bytes / N over one link -- what the reduce-scatter / all-gather pair of XRNERF_DP=zero1
              maps to when RCCL uses every link."""
    n = int(world_size)
    if n <= 1:
        return {'world_size': n, 'bytes_per_step': 0, 'ring_ms': 0.0, 'direct_ms': 0.0}
    b = wire_bytes_per_float * table_floats + 4.0 * mlp_floats
    wire = 2.0 * (n - 1) / n * b
    ring_ms = wire / (link_GBs * 1e9) * 1e3
    direct_ms = (2.0 * b / n) / (link_GBs * 1e9) * 1e3 if n <= links + 1 else ring_ms
    return {'world_size': n, 'gradient_bytes_per_rank': b, 'bytes_on_each_link_per_step_ring': wire, 'assumed_link_GBs': link_GBs,
            'ring_ms': ring_ms, 'direct_ms': direct_ms, 'step_ms_single_gpu': step_ms,
            'overlappable_ms': 'all-reduce form: the 32-MB fine bucket runs under the coarse levels\' scatter (~0.06 ms); zero1 form: none '
                               '(one bucket), but the table\'s Adam pass drops from 0.070 ms to 0.070 / N',
            'efficiency_if_fully_exposed': {'ring': step_ms / (step_ms + ring_ms), 'direct': step_ms / (step_ms + direct_ms)},
            'note': 'model, not a measurement: RCCL over xGMI has never run in this repository beyond world size 1 (single-GPU boxes). '
                    'The data-parallel step keeps gradient + separate optimiser launches (the fused table update of one GPU needs the '
                    'reduced gradient first): its single-GPU-equivalent step is ~0.015 ms longer than the N = 1 line'}


def row_band(H, rank, world_size):
    """contiguous band of image rows of rank `rank`: (row0, nrows); bands differ by at most one row"""
    base, rem = divmod(H, world_size)
    row0 = rank * base + min(rank, rem)
    return row0, base + (1 if rank < rem else 0)


def gather_image(tile, H, rank, world_size):
    """all-gather of the rendered row bands -> the full [H, W, C] image on every rank.
    Bands may differ by one row: tiles are padded to the widest band for the collective."""
    if world_size <= 1:
        return tile
    max_rows = row_band(H, 0, world_size)[1]
    W, Cn = tile.shape[1], tile.shape[2]
    padded = torch.zeros((max_rows, W, Cn), dtype=tile.dtype, device=tile.device)
    padded[:tile.shape[0]] = tile
    out = torch.empty((world_size, max_rows, W, Cn), dtype=tile.dtype, device=tile.device)
    if dist.get_backend() == 'gloo':      # gloo has no all_gather_into_tensor for device tensors
        parts = [torch.empty_like(padded) for _ in range(world_size)]
        dist.all_gather(parts, padded)
        out = torch.stack(parts, 0)
    else:
        dist.all_gather_into_tensor(out.view(-1), padded.view(-1))
    rows = [out[r, :row_band(H, r, world_size)[1]] for r in range(world_size)]
    return torch.cat(rows, 0)
