"""OCR (object-contextual representations) head on the HIP operator surface.
Module tree = network/ocr_utils.py of the reference (same state_dict keys)."""
import torch
from torch import nn

from .. import ops
from ..nn import Conv2d, BNReLU, conv_bn


class SpatialGather_Module(nn.Module):
    """network/ocr_utils.py:17-46.  feats [B,H,W,C] bf16, probs (aux logits)
    [B,H,W,K] fp32 -> object-region context [B,K,1,C] (the reference's
    [B,C,K,1], channels-last)."""

    def __init__(self, cls_num=0, scale=1):
        super().__init__()
        self.cls_num = cls_num
        self.scale = scale
        assert scale == 1

    def forward(self, feats, probs):
        """feats / probs: tensors, or lists of tensors (the scale passes)."""
        B = ops.backend()
        if isinstance(feats, (list, tuple)):
            # a pass is a chain of eight launches of a few workgroups: all but the last pass on the forked stream
            n = len(feats)
            joins = [B.fork(lambda f=f, p=p: self.forward(f, p), tag="ocr", has_bn=False)
                     for f, p in zip(feats[:n - 1], probs[:n - 1])]
            last = self.forward(feats[n - 1], probs[n - 1])
            return [j() for j in joins] + [last]
        ctx = B.ocr_gather(feats, probs)            # [B,K,C] fp32
        return B.to_act(ctx).unsqueeze(2)           # [B,K,1,C]


def _conv_bnrelu_stack(cin, cout, n):
    layers = []
    for i in range(n):
        layers += [Conv2d(cin if i == 0 else cout, cout, kernel_size=1, stride=1, padding=0, bias=False),
                   BNReLU(cout)]
    return nn.Sequential(*layers)


def _run_stack(stack, x, out=None):
    """conv-BN-ReLU stack; out: where the LAST layer's result goes (ops.cat_slots), default a new tensor."""
    for i in range(0, len(stack), 2):
        x = conv_bn(stack[i], stack[i + 1][0], x, relu=True, out=out if i + 2 >= len(stack) else None)
    return x


class ObjectAttentionBlock(nn.Module):
    """network/ocr_utils.py:49-119"""

    def __init__(self, in_channels, key_channels, scale=1):
        super().__init__()
        assert scale == 1
        self.scale = scale
        self.in_channels = in_channels
        self.key_channels = key_channels
        self.pool = nn.MaxPool2d(kernel_size=(scale, scale))
        self.f_pixel = _conv_bnrelu_stack(in_channels, key_channels, 2)
        self.f_object = _conv_bnrelu_stack(in_channels, key_channels, 2)
        self.f_down = _conv_bnrelu_stack(in_channels, key_channels, 1)
        self.f_up = _conv_bnrelu_stack(key_channels, in_channels, 1)

    def forward(self, x, proxy, out=None):
        """x / proxy: tensors, or lists of tensors (the scale passes; the 1x1 conv stacks then run
        as grouped launches).  out: placement of the result (ops.cat_slots)."""
        B = ops.backend()
        # key and value come from the K object regions (launches of 2-8 workgroups): a parallel branch of the query stack
        join_kv = B.fork(lambda: (_run_stack(self.f_object, proxy),     # [B,K,1,D]
                                  _run_stack(self.f_down, proxy)), tag="ocr")
        q = _run_stack(self.f_pixel, x)                     # [B,H,W,D]
        k, v = join_kv()
        scale = self.key_channels ** -0.5
        if isinstance(x, (list, tuple)):
            ctx = [B.ocr_attention(qi, ki.squeeze(2), vi.squeeze(2), scale) for qi, ki, vi in zip(q, k, v)]
        else:
            ctx = B.ocr_attention(q, k.squeeze(2), v.squeeze(2), scale)
        return _run_stack(self.f_up, ctx, out=out)


class SpatialOCR_Module(nn.Module):
    """network/ocr_utils.py:122-158.  Dropout2d(p) is applied as a per-(image,
    channel) multiplier fused into the BN+ReLU pass."""

    def __init__(self, in_channels, key_channels, out_channels, scale=1, dropout=0.1):
        super().__init__()
        self.object_context_block = ObjectAttentionBlock(in_channels, key_channels, scale)
        self.conv_bn_dropout = nn.Sequential(
            Conv2d(2 * in_channels, out_channels, kernel_size=1, padding=0, bias=False),
            BNReLU(out_channels),
            nn.Dropout2d(dropout))

    def _mask(self, x):
        drop = self.conv_bn_dropout[2]
        if not (self.training and drop.p > 0):
            return None
        n, c = x.shape[0], self.conv_bn_dropout[0].out_channels
        keep = 1.0 - drop.p
        return (torch.rand(n, c, device=x.device) < keep).to(torch.float32) / keep

    def forward(self, feats, proxy_feats, feats_cat=None, context_out=None):
        """feats / proxy_feats: tensors, or lists of tensors (the scale passes).  feats_cat: a second handle of
        `feats` for the concatenation (ops.fan_out, see OCR_block.forward); default: feats itself.  context_out: the
        slot in front of feats' in a shared buffer (ops.cat_slots): torch.cat([context, feats], 1) of
        network/ocr_utils.py:151 is then the buffer itself."""
        B = ops.backend()
        context = self.object_context_block(feats, proxy_feats, out=context_out)
        if feats_cat is None:
            feats_cat = feats
        if isinstance(feats, (list, tuple)):
            x = [B.cat([c, f]) for c, f in zip(context, feats_cat)]
            post = [self._mask(xi) for xi in x]
        else:
            x = B.cat([context, feats_cat])
            post = self._mask(x)
        return conv_bn(self.conv_bn_dropout[0], self.conv_bn_dropout[1][0], x, relu=True, post=post)
