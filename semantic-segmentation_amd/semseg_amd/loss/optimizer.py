"""Optimizer and LR schedule of the training step (loss/optimizer.py:43-98 of the
reference): `get_optimizer(args, net) -> (optimizer,
scheduler)` with the reference's arguments (`--optimizer sgd`, `--lr`,
`--weight_decay`, `--momentum`, `--lr_schedule poly|poly2|scl-poly`, `--poly_exp`,
`--poly_step`, `--max_epoch`, `--rescale`, `--repoly`).

The optimizer is SGD with momentum and weight decay as ONE streaming pass over
all parameter tensors in a handful of launches (ssa_sgd_momentum_step); its
state_dict has torch.optim.SGD's layout ('momentum_buffer'), so the reference's
checkpoints restore into it (`restore_opt`) and vice versa."""
import contextlib
import ctypes
import math

import torch
from torch import optim

from .._lib import lib, check
from ..config import cfg


def _on_gpu(p):
    return p.is_cuda


@contextlib.contextmanager
def _launch_scope(device):
    """Make `device` current and yield (stream handle, is that stream being captured)."""
    with torch.cuda.device(device):
        yield torch.cuda.current_stream().cuda_stream, torch.cuda.is_current_stream_capturing()


class FusedSGD(optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay, nesterov) semantics, dampening 0.

    The learning rate is read by the kernel from a device scalar, so a captured
    hipGraph of the training step follows the LR schedule: after the scheduler
    changed `param_groups[..]['lr']` call `sync_lr()` (step() does it itself when
    it is not being captured)."""

    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid SGD hyper-parameters")
        if dampening != 0.0:
            raise ValueError("dampening is not supported (the reference trains with 0)")
        if nesterov and momentum <= 0.0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self._lr_dev = {}          # group index -> (device scalar, value it holds)
        self.loss_scaler = None    # semseg_amd.amp.LossScaler (fp16 training): un-scale, overflow test, skipped steps
        self._pending_scaler_state = None   # a checkpoint's scaler state loaded before a scaler was attached

    def _lr_scalar(self, gi, group, device, capturing):
        ent = self._lr_dev.get(gi)
        lr = float(group["lr"])
        if ent is None or ent[0].device != device:
            ent = self._lr_dev[gi] = [torch.full((1,), lr, dtype=torch.float32, device=device), lr]
        elif ent[1] != lr and not capturing:
            ent[0].fill_(lr)
            ent[1] = lr
        return ent[0]

    def sync_lr(self):
        """Push the groups' current learning rates to the device scalars the kernels read."""
        for gi, group in enumerate(self.param_groups):
            ent = self._lr_dev.get(gi)
            if ent is not None and ent[1] != float(group["lr"]):
                ent[0].fill_(float(group["lr"]))
                ent[1] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        amp = self.loss_scaler
        amp_ptr = amp.state.data_ptr() if amp is not None else None
        if amp is not None:
            # the overflow test covers EVERY gradient before any parameter moves (apex skips the whole step)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                     for group in self.param_groups for p in group["params"] if p.grad is not None]
            if any(g.device != amp.state.device for g in grads):
                raise RuntimeError("FusedSGD with a loss scaler drives ONE device per process: the overflow record lives "
                                   "on %s, a gradient on another device" % amp.state.device)
            if grads:
                with _launch_scope(amp.state.device) as (stream, _):
                    amp.check(grads, stream)
        for gi, group in enumerate(self.param_groups):
            momentum = float(group["momentum"])
            by_device = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (_on_gpu(p) and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                    raise RuntimeError("FusedSGD updates dense fp32 parameters on the GPU")
                if not p.is_contiguous():
                    raise RuntimeError("FusedSGD needs contiguous parameters")
                if not g.is_contiguous():
                    g = g.contiguous()
                buf = None
                if momentum != 0.0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:          # zeros: m*0 + d == d, torch's first step
                        if p.is_cuda and torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("FusedSGD: run one eager step before capturing the step in a "
                                               "graph (the momentum buffers are created on the first step)")
                        buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                by_device.setdefault(p.device, []).append((p, g, buf))
            for device, items in by_device.items():
                n = len(items)
                with _launch_scope(device) as (stream, capturing):
                    lr_dev = self._lr_scalar(gi, group, device, capturing)
                    P = (ctypes.c_void_p * n)(*[p.data_ptr() for p, _, _ in items])
                    G = (ctypes.c_void_p * n)(*[g.data_ptr() for _, g, _ in items])
                    Bf = (ctypes.c_void_p * n)(*[b.data_ptr() for _, _, b in items]) if momentum != 0.0 else None
                    N = (ctypes.c_int64 * n)(*[p.numel() for p, _, _ in items])
                    check(lib().ssa_sgd_momentum_step(P, G, Bf, N, n, float(group["lr"]), lr_dev.data_ptr(),
                                                      momentum, float(group["weight_decay"]),
                                                      int(bool(group["nesterov"])), amp_ptr, stream),
                          "ssa_sgd_momentum_step")
                # the kernel wrote through raw pointers: tell autograd (saved-tensor checks) and the
                # packed-filter cache (hip_backend.refresh_packed_filters keys on ._version) that
                # the parameters and buffers changed
                _mark_updated([p for p, _, _ in items] + [b for _, _, b in items if b is not None])
        if amp is not None:
            with _launch_scope(amp.state.device) as (stream, _):
                amp.update(stream)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        if self.loss_scaler is not None:
            sd["loss_scaler"] = self.loss_scaler.state_dict()       # (apex keeps amp.state_dict() beside the optimizer's)
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        ls = state_dict.pop("loss_scaler", None)
        super().load_state_dict(state_dict)
        if ls is not None:
            if self.loss_scaler is not None:
                self.loss_scaler.load_state_dict(ls)
            else:                       # restored before amp.initialize: semseg_amd.amp.attach_scaler applies it
                self._pending_scaler_state = dict(ls)


def _mark_updated(tensors):
    torch.autograd.graph.increment_version(tensors)


def poly_schedules(args):
    """The LR multipliers of loss/optimizer.py:67-92, by name."""
    def poly_schd(epoch):
        return math.pow(1 - epoch / args.max_epoch, args.poly_exp)

    def poly2_schd(epoch):
        poly_exp = args.poly_exp if epoch < args.poly_step else 2 * args.poly_exp
        return math.pow(1 - epoch / args.max_epoch, poly_exp)

    def scl_poly_schd(epoch):
        thresh = cfg.get("REDUCE_BORDER_EPOCH", -1)
        if epoch < thresh:
            return math.pow(1 - epoch / args.max_epoch, args.poly_exp)
        return args.rescale * math.pow(1 - (epoch - thresh) / (args.max_epoch - thresh), args.repoly)

    return {"poly": poly_schd, "poly2": poly2_schd, "scl-poly": scl_poly_schd}


def get_optimizer(args, net):
    """loss/optimizer.py:43-98.  SGD runs on the fused HIP step; Adam / RAdam are not on the
    accelerated path (no BASELINE.json recipe uses them)."""
    if args.optimizer != "sgd":
        raise ValueError("Not a valid optimizer on the accelerated path: {}".format(args.optimizer))
    optimizer = FusedSGD(net.parameters(), lr=args.lr, weight_decay=args.weight_decay, momentum=args.momentum,
                         nesterov=False)
    schedules = poly_schedules(args)
    if args.lr_schedule not in schedules:
        raise ValueError("unknown lr schedule {}".format(args.lr_schedule))
    if args.lr_schedule == "scl-poly" and cfg.get("REDUCE_BORDER_EPOCH", -1) == -1:
        raise ValueError("ERROR Cannot Do Scale Poly")
    scheduler = optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=schedules[args.lr_schedule])
    return optimizer, scheduler


def forgiving_state_restore(net, loaded_dict):
    """loss/optimizer.py:134-154: load the entries whose name and size match, keep the rest."""
    net_state_dict = net.state_dict()
    matched = {k: loaded_dict[k] for k in net_state_dict
               if k in loaded_dict and net_state_dict[k].size() == loaded_dict[k].size()}
    net_state_dict.update(matched)
    net.load_state_dict(net_state_dict)
    return net


def restore_opt(optimizer, checkpoint):
    """loss/optimizer.py:124-126"""
    assert "optimizer" in checkpoint, "cant find optimizer in checkpoint"
    optimizer.load_state_dict(checkpoint["optimizer"])


def restore_net(net, checkpoint):
    """loss/optimizer.py:129-131"""
    assert "state_dict" in checkpoint, "cant find state_dict in checkpoint"
    forgiving_state_restore(net, checkpoint["state_dict"])


def restore_snapshot(net, optimizer, snapshot, restore_optimizer_bool):
    """loss/optimizer.py:106-121"""
    checkpoint = torch.load(snapshot, map_location=torch.device("cpu"))
    if optimizer is not None and "optimizer" in checkpoint and restore_optimizer_bool:
        optimizer.load_state_dict(checkpoint["optimizer"])
    net = forgiving_state_restore(net, checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint)
    return net, optimizer


load_weights = restore_snapshot      # loss/optimizer.py:97-103 (the reference's wrapper only adds a log line)
