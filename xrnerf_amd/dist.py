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


class BucketedGradSync:
    """Gradient reduction overlapped with the backward pass that produces the gradients.

    The fused training step hands over each gradient bucket as soon as the kernels writing it are enqueued
    (`ready`): MLP gradients after the MLP backward, the fine half of the hash-grid gradient (levels 8-15,
    32 MB) after its scatter launch -- its all-reduce then runs on RCCL's stream under the scatter of the
    coarse half -- and the coarse half (16.8 MB) last.  `finish` orders the compute stream after all of
    them and returns the factor (1/world) the caller folds into its own gradient scaling.  Three
    collectives per iteration, 10 KB + 32 MB + 16.8 MB: few and large, as xGMI's point-to-point rings want."""

    def __init__(self, world_size):
        self.world_size = int(world_size)
        self._works = []

    def ready(self, bucket):
        if self.world_size > 1:
            # `.data`: the bucket is a slice of a buffer other outputs of the fused step are views of (the loss scalar);
            # the in-place reduction must not bump THEIR autograd version counter ("a view ... has been modified inplace")
            self._works.append(dist.all_reduce(bucket.data, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for w in self._works:
            w.wait()
        self._works = []
        return 1.0 / self.world_size


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
