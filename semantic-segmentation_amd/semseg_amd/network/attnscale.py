"""Attention-to-scale DeepLabV3+ heads (network/attnscale.py of the reference): `ASDV3P`, whose
attention over ALL scales is predicted jointly from the concatenated multi-scale decoder features
(`_forward_fused`, attnscale.py:127-183), and `ASDV3P_Paired`, which predicts a two-channel attention
for every pair of neighbouring scales and chains them (`_forward_paired`, attnscale.py:298-362).
Same factories (`DeepV3R50`, `DeepV3R50B`, `DeepV3R50BP`), module tree and state_dict keys, on the HIP
operator surface.  (`DeepV3W38` needs the wrn38 trunk, which no BASELINE.json configuration uses.)

Faithful to the reference including its call contract: `ASDV3P.forward` returns
`{'pred': _forward_fused(inputs)}` -- in training that dict holds the LOSS, in eval the tuple
(output, attn) -- which the reference's train.py cannot consume (train.py:491 expects the loss
itself).  `_forward_fused` / `_forward_paired` are the well-defined functions, pinned against the
real reference in tests/test_attnscale_cpu.py; `ASDV3P_Paired.forward` returns the loss in training
as the reference's does.
"""
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, conv_bn, initialize_weights
from .deepv3 import get_aspp
from .mynn import Upsample, resized_hw
from .ocrnet import _Base, _nchw
from .utils import get_trunk


def _scale_attn_head(num_scales, bn_head, sigmoid):
    """attnscale.py:78-95 / 241-258: children indexed as in the reference's nn.Sequential."""
    if bn_head:
        layers = [Conv2d(num_scales * (256 + 48), 256, kernel_size=3, padding=1, bias=False), Norm2d(256),
                  nn.ReLU(inplace=True),
                  Conv2d(256, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
                  Conv2d(256, num_scales, kernel_size=1, bias=False)]
        if sigmoid:
            layers.append(nn.Sigmoid())
        return nn.Sequential(*layers)
    # the reference's 1x1 conv has padding=1: its output is two pixels larger than its input
    return nn.Sequential(Conv2d(num_scales * (256 + 48), 512, kernel_size=3, padding=1, bias=False),
                         nn.ReLU(inplace=True),
                         Conv2d(512, num_scales, kernel_size=1, padding=1, bias=False))


class _ASBase(_Base):
    def _build(self, num_classes, trunk, criterion, num_scales, bn_head, sigmoid):
        self.criterion = criterion
        self.backbone, s2_ch, _s4_ch, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=256, output_stride=8)
        self.bot_fine = Conv2d(s2_ch, 48, kernel_size=1, bias=False)
        self.bot_aspp = Conv2d(aspp_out_ch, 256, kernel_size=1, bias=False)
        self.final = nn.Sequential(
            Conv2d(256 + 48, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, num_classes, kernel_size=1, bias=False))
        self.bn_head = bool(cfg.MODEL.get("ATTNSCALE_BN_HEAD", False) or bn_head)
        self.scale_attn = _scale_attn_head(num_scales, self.bn_head, sigmoid)
        if cfg.OPTIONS.INIT_DECODER:
            initialize_weights(self.bot_fine, self.bot_aspp, self.scale_attn, self.final)
        else:
            initialize_weights(self.final)

    def _fwd(self, x, size):
        """attnscale.py:103-125: (fp32 logits at `size`, decoder features at stride 4)."""
        B = ops.backend()
        s2_features, _, final_features = self.backbone(x)
        conv_aspp = self.bot_aspp(self.aspp(final_features))
        conv_s2 = self.bot_fine(s2_features)
        cat_s4 = B.cat([conv_s2, B.bilinear(conv_aspp, s2_features.shape[1:3])])
        h = conv_bn(self.final[0], self.final[1], cat_s4, relu=True)
        h = conv_bn(self.final[3], self.final[4], h, relu=True)
        return Upsample(self.final[6](h, out_f32=True), size), cat_s4

    def _attn(self, feats):
        """The scale-attention head on concatenated decoder features -> fp32 [B,h,w,S]."""
        B = ops.backend()
        sa = self.scale_attn
        if self.bn_head:
            h = conv_bn(sa[0], sa[1], feats, relu=True)
            h = conv_bn(sa[3], sa[4], h, relu=True)
            a = sa[6](h, out_f32=True)
            return B.sigmoid(a) if len(sa) == 8 else a
        return sa[2](B.relu(sa[0](feats)), out_f32=True)

    def _passes(self, inputs, scales):
        """Every scale's pass: logits resampled to the 1x image, decoder features resampled to the 1x
        features (attnscale.py:136-151)."""
        B = ops.backend()
        x_1x, size = self._images(inputs)
        p_1x, feats_1x = self._fwd(x_1x, size)
        ps, feats = {1.0: p_1x}, {1.0: feats_1x}
        for s in scales:
            if s == 1.0:
                continue
            x, ssize = self._images(inputs, s)
            p, f = self._fwd(x, ssize)
            ps[s] = B.bilinear(p, size, out_f32=True)
            feats[s] = B.bilinear(f, feats_1x.shape[1:3])
        return ps, feats, size

    @staticmethod
    def _weighted_sum(terms):
        """sum_s p_s * attn_s with the attention broadcast over the classes (fp32)."""
        B = ops.backend()
        out = None
        for p, a in terms:
            t = B.bcast_mul(a, p)
            out = t if out is None else B.ewise("add", out, t)
        return out


class ASDV3P(_ASBase):
    """network/attnscale.py:39-183"""

    def __init__(self, num_classes, trunk="resnet-50", criterion=None, use_dpc=False, fuse_aspp=False,
                 attn_2b=False, bn_head=False):
        super().__init__()
        assert not use_dpc
        assert cfg.MODEL.N_SCALES is not None
        self.fuse_aspp, self.attn_2b = fuse_aspp, attn_2b
        self.scales = sorted(cfg.MODEL.N_SCALES)
        self._build(num_classes, trunk, criterion, len(self.scales), bn_head, sigmoid=False)

    def _forward_fused(self, inputs):
        B = ops.backend()
        assert 1.0 in self.scales, "expected one of scales to be 1.0"
        ps, feats, size = self._passes(inputs, self.scales)
        order = [1.0] + [s for s in self.scales if s != 1.0]        # concatenation order of attnscale.py:141-152
        attn_tensor = self._attn(B.cat([feats[s] for s in order]))
        terms, attn = [], None
        for idx, s in enumerate(self.scales):
            attn = attn_tensor[..., idx:idx + 1]
            terms.append((ps[s], B.bilinear(attn, size, out_f32=True)))
        output = self._weighted_sum(terms)
        if self.training:
            gts = inputs["gts"]
            loss = self.criterion(_nchw(output), gts)
            if cfg.LOSS.SUPERVISED_MSCALE_WT:
                for s in self.scales:
                    loss = loss + cfg.LOSS.SUPERVISED_MSCALE_WT * self.criterion(_nchw(ps[s]), gts, do_rmi=False)
            return loss
        return _nchw(output), _nchw(attn)

    def forward(self, inputs):
        B = ops.backend()
        B.begin_step(inputs["images"].device)
        out = self._forward_fused(inputs)
        B.end_forward()
        return {"pred": out}


class ASDV3P_Paired(_ASBase):
    """network/attnscale.py:199-368"""

    def __init__(self, num_classes, trunk="resnet-50", criterion=None, use_dpc=False, fuse_aspp=False,
                 attn_2b=False, bn_head=False):
        super().__init__()
        assert not use_dpc
        assert cfg.MODEL.N_SCALES is not None
        self.fuse_aspp, self.attn_2b = fuse_aspp, attn_2b
        self.trn_scales = (0.5, 1.0)
        self.inf_scales = sorted(cfg.MODEL.N_SCALES)
        self._build(num_classes, trunk, criterion, 2, bn_head, sigmoid=True)

    def _forward_paired(self, inputs, scales):
        B = ops.backend()
        assert 1.0 in scales, "expected one of scales to be 1.0"
        scales = list(scales)
        ps, feats, size = self._passes(inputs, scales)
        attn = {}
        for idx in range(len(scales) - 1):
            lo, hi = scales[idx], scales[idx + 1]
            attn[lo] = B.bilinear(self._attn(B.cat([feats[lo], feats[hi]])), size, out_f32=True)   # [B,H,W,2]
        norm_attn, last_attn = {}, None
        for idx in range(len(scales) - 1):
            lo, hi = scales[idx], scales[idx + 1]
            a_lo, a_hi = attn[lo][..., 0:1], attn[lo][..., 1:2]
            if last_attn is None:
                norm_attn[lo], norm_attn[hi] = a_lo, a_hi
            else:
                norm = B.ewise("div", last_attn, B.ewise("add", a_lo, a_hi))
                norm_attn[lo] = B.ewise("mul", a_lo, norm)
                norm_attn[hi] = B.ewise("mul", a_hi, norm)
            last_attn = a_hi
        last = None
        terms = []
        for s in scales:
            last = B.bilinear(norm_attn[s], size, out_f32=True)
            terms.append((ps[s], last))
        output = self._weighted_sum(terms)
        if self.training:
            return self.criterion(_nchw(output), inputs["gts"])
        return _nchw(output), _nchw(last)

    def forward(self, inputs):
        B = ops.backend()
        B.begin_step(inputs["images"].device)
        if self.training:
            out = self._forward_paired(inputs, self.trn_scales)
        else:
            out = {"pred": self._forward_paired(inputs, self.inf_scales)}
        B.end_forward()
        return out


def DeepV3R50(num_classes, criterion):
    return ASDV3P(num_classes, trunk="resnet-50", criterion=criterion)


def DeepV3R50B(num_classes, criterion):
    """Batch-norm head"""
    return ASDV3P(num_classes, trunk="resnet-50", criterion=criterion, bn_head=True)


def DeepV3R50BP(num_classes, criterion):
    """Batch-norm head with paired attention"""
    return ASDV3P_Paired(num_classes, trunk="resnet-50", criterion=criterion, bn_head=True)
