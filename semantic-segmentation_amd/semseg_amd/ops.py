"""Operator surface used by semseg_amd.network / semseg_amd.loss.

All functions take and return NHWC tensors ([B,H,W,C]); activations in the
backend's `act_dtype` (bf16 on the HIP backend), logits/losses fp32.  The
default -- and only shipped -- backend is the HIP one (hip_backend.py over
libsemseg_hip.so); it is created on first use and raises if the library is
missing.  `_set_backend_for_tests` exists so the CPU test-suite can check the
module wiring against the reference with the oracle's operators; product code
never calls it.
"""
import torch

_BACKEND = None


class HipBackend:
    name = "hip"

    def __init__(self):
        from . import hip_backend as hb
        hb.lib()   # fail loudly now if the extension is absent
        self.hb = hb
        self.act_dtype = hb.ACT_DTYPE

    def begin_step(self, device=None):
        self.hb.begin_step(device)

    def image_to_nhwc(self, images, out_hw=None):
        return self.hb.image_to_nhwc(images, out_hw)

    def conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False, want_stats=False):
        return self.hb.Conv2dFn.apply(x, weight, bias, stride, padding, dilation, out_f32, want_stats)

    def conv_bn_act(self, conv, bn, x, residual=None, relu=False, post=None):
        """conv -> BatchNorm (+residual, ReLU, mask).  In training the conv epilogue
        accumulates the batch statistics where the kernel supports it, and the
        normalisation picks them up instead of re-reading the conv output."""
        y = self.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0], False,
                        bool(bn.training))
        return self.batch_norm_act(y, bn, residual, relu, post)

    def batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        return self.hb.BatchNormActFn.apply(
            x, bn.weight, bn.bias, residual, post, bn.running_mean, bn.running_var,
            bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None,
            0.1 if bn.momentum is None else bn.momentum, bn.eps, bn.training, relu,
            getattr(bn, "sync", False))

    def sum_act(self, tensors, relu=True):
        return self.hb.SumActFn.apply(relu, *tensors)

    def bilinear(self, x, size, out_f32=False):
        if tuple(x.shape[1:3]) == tuple(size) and (x.dtype == torch.float32) == bool(out_f32 or x.dtype == torch.float32):
            return x
        return self.hb.BilinearFn.apply(x, int(size[0]), int(size[1]), bool(out_f32))

    def cat(self, tensors):
        return torch.cat(tensors, dim=3)      # pure data movement

    def to_act(self, x):
        return x.to(self.act_dtype)           # dtype cast only

    def ocr_gather(self, feats, logits):
        return self.hb.OcrGatherFn.apply(feats, logits)

    def ocr_attention(self, q, k, v, scale):
        return self.hb.OcrAttnFn.apply(q, k, v, scale)

    def sigmoid(self, x):
        return self.hb.SigmoidFn.apply(x)

    def bcast_mul(self, a, x):
        return self.hb.BcastMulFn.apply(a, x)

    def attn_blend(self, lo, a, hi):
        return self.hb.AttnBlendFn.apply(lo, a, hi)

    def cross_entropy(self, logits, labels, ignore_index):
        return self.hb.CrossEntropyFn.apply(logits, labels, ignore_index)

    def bce_rmi(self, logits, labels, do_rmi, weight_lambda=0.5):
        return self.hb.BceRmiFn.apply(logits, labels, bool(do_rmi), weight_lambda)


def backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def _set_backend_for_tests(b):
    """Test-suite hook (tests/test_wiring_cpu.py).  Not used by product code."""
    global _BACKEND
    _BACKEND = b
