"""fp16 training: the reference's `--fp16` path (apex.amp, train.py:380-381 `amp.initialize(net, optim, opt_level)`,
train.py:503-505 `with amp.scale_loss(main_loss, optim) as scaled_loss: scaled_loss.backward()`).

With fp16 storage (`SSA_ACT_DTYPE=fp16`, lib/libsemseg_hip_f16.so) the per-pixel loss gradients of a 1024 x 1024 step
(~1e-6) flush to zero on their way back through the 16-bit activations' gradients, so -- as apex does -- the loss is
multiplied by a scale S before backward, the fp32 parameter gradients are divided by S inside the optimizer, a step
whose gradients hold an inf / nan is skipped, and S follows apex's dynamic schedule (2^16, halved on overflow,
doubled after 2,000 clean steps).  Everything lives on the device:

    state = [S, found_inf, clean steps, 1 / S]        (4 floats; + 2 counters: skipped steps in total / in a row)
    loss * state[0]                                   one 0-dim tensor product (the gradient kernels read the
                                                      upstream gradient from device memory already)
    ssa_amp_check_grads -> ssa_sgd_momentum_step(amp_state) -> ssa_amp_update       (csrc/optim.hip)

so the scaled step is capturable and replays as the same hipGraph as the bf16 step (semseg_amd/graphed.py); the scale a
replay uses is the one the previous replay left.  bf16 storage (the default build) has fp32's exponent range and needs
none of this: `scale_loss` is then the identity and no scaler is attached.
"""
import contextlib

import torch

from ._lib import ACT, check, lib


class LossScaler:
    """apex.amp's dynamic LossScaler as a device record (see the module docstring)."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_interval=2000, growth=2.0, backoff=0.5,
                 min_scale=1.0, max_scale=2.0 ** 24):
        self.growth_interval, self.growth, self.backoff = int(growth_interval), float(growth), float(backoff)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        # [S, found_inf, clean steps, 1 / S | skipped steps in total, skipped steps in a row, -, -]: the first four floats
        # are the record the kernels' C ABI names; the counters behind it are what apex prints per skipped step
        # ("Gradient overflow.  Skipping step") -- here a device count the host reads when it logs or checkpoints
        self.state = torch.tensor([init_scale, 0.0, 0.0, 1.0 / init_scale, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32,
                                  device=device)
        self.warn_after = 50          # consecutive skipped steps at which health() calls the run diverged

    # -- what the training step calls
    def scale(self, loss):
        """loss * S (S read from the device when the product runs)."""
        return loss * self.state[0]

    def check(self, grads, stream):
        """found_inf |= any non-finite element of `grads` (dense fp32 tensors)."""
        import ctypes
        n = len(grads)
        if not n:
            return
        G = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
        N = (ctypes.c_int64 * n)(*[g.numel() for g in grads])
        check(lib().ssa_amp_check_grads(G, N, n, self.state.data_ptr(), stream), "ssa_amp_check_grads")

    def update(self, stream):
        check(lib().ssa_amp_update_counted(self.state.data_ptr(), self.state.data_ptr() + 16, self.growth_interval,
                                           self.growth, self.backoff, self.min_scale, self.max_scale, stream),
              "ssa_amp_update_counted")

    # -- host-side views (synchronise: for logging / tests / checkpoints, not for the step)
    def loss_scale(self):
        return float(self.state[0])

    def skipped_steps(self):
        """(skipped steps in total, skipped steps in a row) -- apex logs one line per skipped step; a captured step cannot."""
        s = self.state.detach().cpu().tolist()
        return int(s[4]), int(s[5])

    def health(self, log=None):
        """None, or a sentence saying why this run is not training: the scale sits on its lower bound, or the last
        `warn_after` steps were all skipped (gradients that are nan at EVERY scale: divergence, not overflow).  Passed to
        `log` (e.g. logx.msg / warnings.warn) when given.  Synchronises; call it where the loop logs, not per step."""
        s = self.state.detach().cpu().tolist()
        msg = None
        if int(s[5]) >= self.warn_after:
            msg = ("loss scaler: the last %d steps were skipped (gradient overflow at every scale down to %g): the "
                   "gradients are non-finite, not merely out of fp16's range" % (int(s[5]), s[0]))
        elif s[0] <= self.min_scale and int(s[5]) > 0:
            msg = "loss scaler: the scale reached its lower bound %g and the step still overflows (%d skipped in a row)" % (
                self.min_scale, int(s[5]))
        if msg and log is not None:
            log(msg)
        return msg

    def state_dict(self):
        s = self.state.detach().cpu().tolist()
        return {"loss_scale": s[0], "unskipped": int(s[2]), "skipped_steps": int(s[4]), "skipped_in_a_row": int(s[5])}

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.state.copy_(torch.tensor([float(sd["loss_scale"]), 0.0, float(sd.get("unskipped", 0)),
                                           1.0 / float(sd["loss_scale"]), float(sd.get("skipped_steps", 0)),
                                           float(sd.get("skipped_in_a_row", 0)), 0.0, 0.0], dtype=torch.float32))


def fp16_storage():
    """True in a process that runs the fp16-storage build of the library (SSA_ACT_DTYPE=fp16)."""
    return ACT == "fp16"


def attach_scaler(optimizer, device, **kw):
    """Give `optimizer` (semseg_amd.loss.optimizer.FusedSGD, possibly behind the graphed proxy) a LossScaler and allow
    the criteria to run backward on the fp16 build.  Returns the scaler."""
    from . import hip_backend
    opt = getattr(optimizer, "_optim", optimizer)
    if not hasattr(opt, "loss_scaler"):
        raise TypeError("loss scaling needs semseg_amd.loss.optimizer.FusedSGD (got %s)" % type(opt).__name__)
    if opt.loss_scaler is None:
        opt.loss_scaler = LossScaler(device, **kw)
        pending = getattr(opt, "_pending_scaler_state", None)
        if pending is not None:          # a checkpoint restored BEFORE amp.initialize (FusedSGD.load_state_dict kept it)
            opt.loss_scaler.load_state_dict(pending)
            opt._pending_scaler_state = None
    hip_backend.enable_fp16_training()
    return opt.loss_scaler


def scaler_of(optimizer):
    opt = getattr(optimizer, "_optim", optimizer)
    return getattr(opt, "loss_scaler", None)


def initialize(model, optimizer=None, opt_level="O1", **kw):
    """apex.amp.initialize: on the fp16 build attach the dynamic loss scaler to the optimizer; bf16: nothing to do."""
    if optimizer is not None and fp16_storage() and not isinstance(optimizer, (list, tuple)):
        dev = next((p.device for p in model.parameters() if p.is_cuda), None)
        if dev is not None:
            attach_scaler(optimizer, dev, **{k: v for k, v in kw.items() if k in ("init_scale", "growth_interval")})
    return model, optimizer


@contextlib.contextmanager
def scale_loss(loss, optimizer, **kw):
    """apex.amp.scale_loss: yields loss * S when `optimizer` carries a scaler (un-scaling, the overflow test and the
    scale update happen inside optimizer.step()), else the loss itself."""
    sc = scaler_of(optimizer) if not isinstance(optimizer, (list, tuple)) else None
    yield sc.scale(loss) if sc is not None else loss
