"""TEST-ONLY operator backend: the oracle's CPU operators (oracle/ops.py) behind
the semseg_amd.ops interface, NHWC fp32.  Lets the CPU suite check the module
wiring of semseg_amd.network against the reference's golden vectors.  It is
injected with ops._set_backend_for_tests(); the product never imports it."""
import torch

from oracle import ops as O
from semseg_amd.ops import BackendBase


def _to_nchw(x):
    return x.permute(0, 3, 1, 2)


def _to_nhwc(x):
    return x.permute(0, 2, 3, 1)


class OracleBackend(BackendBase):
    name = "oracle-cpu"
    act_dtype = torch.float32

    def begin_step(self, device=None):
        pass

    def end_forward(self):
        pass

    def image_to_nhwc(self, images, out_hw=None):
        x = images if images.dtype == torch.float64 else images.float()
        if out_hw is not None and tuple(out_hw) != tuple(x.shape[2:]):
            x = O.bilinear(x, tuple(out_hw))
        x = _to_nhwc(x)
        return torch.nn.functional.pad(x, (0, 16 - x.shape[3]))

    def _conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False):
        xin = _to_nchw(x)[:, :weight.shape[1]]
        return _to_nhwc(O.conv2d(xin, weight, bias, stride, padding, dilation))

    def _conv_bn_act(self, conv, bn, x, residual=None, relu=False, post=None):
        y = self._conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0])
        return self._batch_norm_act(y, bn, residual, relu, post)

    def _batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        y = O.batch_norm(_to_nchw(x), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                         0.1 if bn.momentum is None else bn.momentum, bn.eps)
        y = _to_nhwc(y)
        if residual is not None:
            y = y + residual
        if relu:
            y = torch.relu(y)
        if post is not None:
            y = y * post[:, None, None, :]
        return y

    def _sum_act(self, tensors, relu=True):
        y = tensors[0]
        for t in tensors[1:]:
            y = y + t
        return torch.relu(y) if relu else y

    def _bilinear(self, x, size, out_f32=False):
        if tuple(x.shape[1:3]) == tuple(size):
            return x
        return _to_nhwc(O.bilinear(_to_nchw(x), tuple(size)))

    def max_pool3x3s2(self, x):
        return _to_nhwc(O.max_pool_3x3_s2(_to_nchw(x)))

    def global_avg_pool(self, x):
        return _to_nhwc(O.global_avg_pool(_to_nchw(x)))

    def cat(self, tensors):
        return torch.cat(tensors, dim=3)

    def to_act(self, x):
        return x

    def ocr_gather(self, feats, logits):
        ctx = O.spatial_gather(_to_nchw(feats).contiguous(), _to_nchw(logits).contiguous())  # [B,C,K,1]
        return ctx[..., 0].permute(0, 2, 1)

    def ocr_attention(self, q, k, v, scale):
        B, H, W, D = q.shape
        out = O.object_attention(q.reshape(B, H * W, D), k.permute(0, 2, 1), v, D)
        assert abs(scale - D ** -0.5) < 1e-12
        return out.reshape(B, H, W, D)

    def sigmoid(self, x):
        return torch.sigmoid(x)

    def bcast_mul(self, a, x):
        return a * x

    def attn_blend(self, lo, a, hi):
        return lo + (1 - a) * hi

    def ewise(self, op, a, b):
        return a + b if op == "add" else (a * b if op == "mul" else a / b)

    def relu(self, x):
        return self.sum_act([x], relu=True)

    def cross_entropy(self, logits, labels, ignore_index):
        return O.cross_entropy(_to_nchw(logits), labels, ignore_index)

    def bce_rmi(self, logits, labels, do_rmi, weight_lambda=0.5):
        return O.rmi_loss(_to_nchw(logits), labels, logits.shape[3], do_rmi=do_rmi, weight_lambda=weight_lambda)
