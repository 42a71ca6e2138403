"""TEST-ONLY: the oracle's CPU operators with the product's STORAGE precision
emulated -- every tensor the HIP path keeps in bf16 is rounded to bf16 here
(forward values and the gradients flowing back), arithmetic stays fp32.

Purpose: the end-to-end tests need a principled tolerance.  The network is
~450 tensors deep and some of its layers (BatchNorm over the 19x1 object
proxies, BatchNorm after the near-constant OCR context) amplify relative
perturbations several-fold, so "HIP vs fp32 oracle" has a noise floor set by
bf16 storage alone.  This backend measures that floor on the same weights and
inputs; the e2e tests then require the HIP path to be no further from the fp32
oracle than a stated multiple of it, op by op.  The product never imports it.
"""
import torch

from oracle_backend import OracleBackend


import contextlib

# storage type the emulation rounds to: bf16 = the product's; fp16 = what the reference's apex-O1 path stores
# (`storage(torch.float16)`: the noise floor the REFERENCE'S OWN mixed-precision run has against its fp32 run,
# tests/test_storage_floor_cpu.py)
from util import ACT_DTYPE as _ACT
_STORAGE = [_ACT]     # the product's storage format in this process (SSA_ACT_DTYPE)


@contextlib.contextmanager
def storage(dtype):
    prev = _STORAGE[0]
    _STORAGE[0] = dtype
    try:
        yield
    finally:
        _STORAGE[0] = prev


# channel counts of the tensors the HIP path keeps in fp32 (class logits, attention maps): every tensor here has dtype
# fp32, so the emulation goes by channel count.  A test with another class count adds it (tests/teacher_backend.py
# shares this set).
F32_CHANNELS = {1, 19}


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(_STORAGE[0]).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(_STORAGE[0]).to(g.dtype)


R = _Round.apply


class Bf16EmuBackend(OracleBackend):
    name = "oracle-cpu-bf16-storage"

    def image_to_nhwc(self, images, out_hw=None):
        return R(super().image_to_nhwc(images, out_hw))

    def _conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False):
        y = super()._conv2d(x, R(weight), bias, stride, padding, dilation, out_f32)
        return y if out_f32 else R(y)

    def _batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        return R(super()._batch_norm_act(x, bn, residual, relu, post))

    def _sum_act(self, tensors, relu=True):
        return R(super()._sum_act(tensors, relu))

    def _bilinear(self, x, size, out_f32=False):
        if tuple(x.shape[1:3]) == tuple(size):
            return x
        y = super()._bilinear(x, size, out_f32)
        # class logits ([..,19]) and attention maps ([..,1]) are fp32 tensors on the
        # HIP path too (everything here has dtype fp32, so go by channel count)
        return y if (out_f32 or x.shape[3] in F32_CHANNELS) else R(y)

    def max_pool3x3s2(self, x):
        return R(super().max_pool3x3s2(x))

    def global_avg_pool(self, x):
        return R(super().global_avg_pool(x))

    def ocr_attention(self, q, k, v, scale):
        return R(super().ocr_attention(q, k, v, scale))

    def to_act(self, x):
        return R(x)


# the public, list-aware operator surface: the granularity every backend shares (the HIP backend runs a
# whole residual block, or conv+BN, as one call; the CPU backends round at the same internal points)
TRACED = ("image_to_nhwc", "conv2d", "conv_bn_act", "batch_norm_act", "basic_block", "sum_act", "bilinear",
          "ocr_gather", "ocr_attention", "max_pool3x3s2", "global_avg_pool")


def traced(backend, sink):
    """Wrap `backend` so that every activation-producing op calls
    sink(index, name, output) for each tensor it returns (lists = independent problems, in order).
    Works for the HIP backend and the CPU ones."""
    counter = [0]

    class Traced(type(backend)):
        pass

    def mk(name):
        f = getattr(type(backend), name)

        def g(self, *a, **k):
            y = f(self, *a, **k)
            for t in (y if isinstance(y, (list, tuple)) else (y,)):
                sink(counter[0], name, t)
                counter[0] += 1
            return y
        return g

    for name in TRACED:
        setattr(Traced, name, mk(name))
    from semseg_amd.ops import BackendBase
    if type(backend).upsample_cat is not BackendBase.upsample_cat:
        # a backend with a fused resize + concatenate reports the resized slices of its output where the others
        # report the outputs of their bilinear() call: same tensors, same order
        fused = type(backend).upsample_cat

        def upsample_cat(self, groups):
            ys = fused(self, groups)
            for g, y in zip(groups, ys):
                off = g[0].shape[3]
                for t in g[1:]:
                    sink(counter[0], "bilinear", y[..., off:off + t.shape[3]])
                    counter[0] += 1
                    off += t.shape[3]
            return ys
        Traced.upsample_cat = upsample_cat
    backend.__class__ = Traced
    return backend
