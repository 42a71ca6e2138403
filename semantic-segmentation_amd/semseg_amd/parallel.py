"""Data-parallel training over RCCL/xGMI: one process per GPU.

`DistributedDataParallel` stands in for apex.parallel.DistributedDataParallel
(network/__init__.py:37-39 of the reference): gradients are averaged over ranks.
On the HIP path the parameter gradients of a step live in the backend's gradient arena
(hip_backend._GradArena: a few contiguous fp32 chunks that fill in backward order).  The exchange is
OVERLAPPED with backward: every SSA_DDP_FLUSH_AT (256) queued layers the weight gradients are flushed and the arena
range that is now final is all-reduced (mean) in place, by a direct RCCL call, on a communication stream of its own
(second communicator), ordered behind the weight-gradient stream by an event -- about three exchanges per step
(~110, ~110, ~65 MB at 1024x1024), the last of which has no later compute to hide behind.  No flattening copy, no
per-parameter hooks; the exchanges are nodes (a parallel branch) of the same captured hipGraph as the single-GPU step.
`SSA_DDP_OVERLAP=0`: the same ranges on the compute stream, one communicator.

Order of collectives across ranks (what keeps two communicators from dead-locking): every rank runs the same
program on the same shapes, so communicator 0's calls (SyncBN sums, level by level; the autograd-bucket all-reduce)
are issued in the same order on its compute stream everywhere, and communicator 1's (arena ranges) in the same order
on its communication stream; no kernel of either stream waits for anything but (a) earlier work of its own stream and
(b) the fork/join events, which point from the compute stream to the communication stream before an exchange and
back only at the end of backward.  All other kernels of the step are finite, so a collective of either communicator
that has been launched on every rank always becomes resident on every rank.

Gradients that reach a parameter through autograd (conv biases; every parameter under the CPU test backend) are
collected by hooks and exchanged as one flat bucket in an end-of-backward callback.
SyncBN lives in semseg_amd.nn.SyncBatchNorm; its exchange is `allreduce_bn_sums`.
`backend='nccl'` is RCCL on ROCm.
"""
import os
import weakref

import torch
import torch.distributed as dist
from torch import nn

# SSA_FORCE_DIST=1: treat a world of ONE rank like a distributed job (SyncBN exchanges and the
# gradient exchange run, over a 1-rank RCCL communicator).  Lets a one-GPU box exercise the code
# path -- in particular inside hipGraph capture -- that the multi-GPU runs take.
_FORCE = os.environ.get("SSA_FORCE_DIST", "0") == "1"


def sync_world(enabled=True, group=None):
    """World size to synchronise BN statistics over (0 = no synchronisation)."""
    if enabled and dist.is_available() and dist.is_initialized():
        ws = dist.get_world_size(group)
        return ws if (ws > 1 or _FORCE) else 0
    return 0


def _all_reduce_(t, average=False, group=None, comm_index=0):
    from . import rccl, p2p
    if not average and p2p.usable(t, group):
        p2p.exchange(group).all_reduce_sum_(t)             # SSA_SYNCBN_P2P=1: one kernel over peer-mapped buffers
    elif rccl.usable(t, group):
        rccl.comm(comm_index).all_reduce_(t, average)      # one RCCL call on the current stream
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if average:
            t.div_(dist.get_world_size(group))
    return t


def allreduce_bn_sums(sums, group=None):
    """SyncBN exchange (apex.parallel.SyncBatchNorm, config.py:216-222): SUM the per-channel fp64
    partial sums over ranks in place.  Every rank runs the same crop size, so the caller's global
    count is local * world.  With (sum x, sum x^2) this makes the statistics those of the
    concatenated global batch; with (sum dy, sum dy*xhat) likewise for the backward.  `sums` may
    span the partial sums of every problem of a grouped BatchNorm level (one collective)."""
    if sync_world(True, group):
        _all_reduce_(sums, False, group)
    return sums


class _Sink:
    """What hip_backend's gradient arena talks to (weak: a dead wrapper exchanges nothing)."""

    def __init__(self, ref):
        self.ref = ref

    def reduce_range(self, buf, lo, hi):
        ddp = self.ref()
        if ddp is not None and hi > lo:
            ddp._reduce_range(buf, lo, hi)

    def finish(self):
        ddp = self.ref()
        if ddp is not None:
            ddp._finish_exchanges()


class DistributedDataParallel(nn.Module):
    def __init__(self, module, message_size=10_000_000, delay_allreduce=False, process_group=None, **_):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (_FORCE and dist.is_initialized())
        self._hooked = []
        self._callback_queued = False
        if self.active:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=self.group)
            for p in module.parameters():
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._on_grad)
            from . import hip_backend
            self.exchanges = 0            # gradient all-reduces issued in the current backward pass
            self.tail_elements = 0        # elements of the LAST exchange: the part no later compute can hide
            self._comm_stream = None
            self._overlap = os.environ.get("SSA_DDP_OVERLAP", "1") != "0"
            hip_backend.set_grad_sink(_Sink(weakref.ref(self)))

    def __del__(self):
        try:
            from . import hip_backend
            sink = hip_backend._GRAD_SINK[0]
            if isinstance(sink, _Sink) and sink.ref() is None:
                hip_backend.set_grad_sink(None)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    def forward(self, *args, **kwargs):
        self._hooked = []
        self._callback_queued = False
        if self.active:
            self.exchanges = 0
        return self.module(*args, **kwargs)

    # -- gradients accumulated by the kernels in the backend's arena: ranges as they complete (hip_backend._GradArena)
    def _reduce_range(self, buf, lo, hi):
        """Mean over ranks of buf[lo:hi], in place.  Device memory: on a communication stream of its own (second RCCL
        communicator: the SyncBN exchanges keep the compute stream's), fenced by an event after the kernels that
        produced the range -- a parallel branch of the captured step, concurrent with the rest of backward."""
        t = buf[lo:hi]
        self.exchanges += 1
        self.tail_elements = hi - lo
        if t.is_cuda and self._overlap:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                _all_reduce_(t, True, self.group, comm_index=1)
        else:
            _all_reduce_(t, True, self.group)

    def _finish_exchanges(self):
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    # -- gradients that arrive through autograd (autograd thread)
    def _on_grad(self, p):
        self._hooked.append(p)
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True

    def _finalize(self):
        ps, self._hooked = self._hooked, []
        self._callback_queued = False
        ps = [p for p in ps if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        _all_reduce_(flat, True, self.group)
        off = 0
        for p in ps:                     # the averaged gradients stay where they are: views of the bucket
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n
