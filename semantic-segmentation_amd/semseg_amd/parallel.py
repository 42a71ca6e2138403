"""Data-parallel training over RCCL/xGMI: one process per GPU.

`DistributedDataParallel` stands in for apex.parallel.DistributedDataParallel
(network/__init__.py:37-39 of the reference): gradients are averaged over ranks
with bucketed all-reduces that are launched from autograd hooks while backward
is still running (reverse-registration order, >= `message_size` elements per
bucket, as apex's default of 1e7).  `backend='nccl'` is RCCL on ROCm.
SyncBN lives in semseg_amd.nn.SyncBatchNorm.
"""
import os

import torch
import torch.distributed as dist
from torch import nn

# SSA_FORCE_DIST=1: treat a world of ONE rank like a distributed job (SyncBN exchanges and the
# DDP hooks run, over a 1-rank RCCL communicator).  Lets a one-GPU box exercise the c10d code
# path -- in particular inside hipGraph capture -- that the multi-GPU runs take.
_FORCE = os.environ.get("SSA_FORCE_DIST", "0") == "1"


def sync_world(enabled=True, group=None):
    """World size to synchronise BN statistics over (0 = no synchronisation)."""
    if enabled and dist.is_available() and dist.is_initialized():
        ws = dist.get_world_size(group)
        return ws if (ws > 1 or _FORCE) else 0
    return 0


def allreduce_bn_sums(sums, local_count, group=None):
    """SyncBN exchange (apex.parallel.SyncBatchNorm, config.py:216-222): SUM the
    per-channel fp64 partial sums over ranks in place and return the global
    sample count.  Every rank runs the same crop size, so count = local * world.
    With (sum x, sum x^2) this makes the statistics those of the concatenated
    global batch; with (sum dy, sum dy*xhat) likewise for the backward."""
    world = sync_world(True, group)
    if not world:
        return float(local_count)
    from . import rccl
    if rccl.ENABLED and group is None and sums.is_cuda:
        rccl.comm().all_reduce_sum_(sums)        # one RCCL call on the stream the BN kernels run on
    else:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return float(local_count) * world


class DistributedDataParallel(nn.Module):
    def __init__(self, module, message_size=10_000_000, delay_allreduce=False, process_group=None, **_):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self.delay_allreduce = delay_allreduce
        self.active = self.world > 1 or (_FORCE and dist.is_initialized())
        params = [p for p in module.parameters() if p.requires_grad]
        if self.active:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=self.group)
        # buckets in reverse parameter order: the order backward produces grads
        self.buckets, cur, n = [], [], 0
        for p in reversed(params):
            cur.append(p)
            n += p.numel()
            if n >= message_size:
                self.buckets.append(cur)
                cur, n = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
        self._pending = [0] * len(self.buckets)
        self._inflight = []
        self._callback_queued = False
        if self.active:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *args, **kwargs):
        self._pending = [len(b) for b in self.buckets]
        self._inflight = []
        self._callback_queued = False
        return self.module(*args, **kwargs)

    # -- autograd-thread side
    def _on_grad(self, p):
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self.delay_allreduce:
            self._launch(bi)

    def _launch(self, bi):
        ps = self.buckets[bi]
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._inflight.append((bi, flat, work))

    def _finalize(self):
        launched = {bi for bi, _, _ in self._inflight}
        for bi in range(len(self.buckets)):
            if bi not in launched and all(p.grad is not None for p in self.buckets[bi]):
                self._launch(bi)
        for bi, flat, work in self._inflight:
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in self.buckets[bi]:      # the averaged gradients stay where they are: views of the bucket
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        self._inflight = []
