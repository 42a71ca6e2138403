"""Multi-scale model whose attention is predicted from the features of BOTH
scales (network/mscale2.py of the reference): `mscale2.DeepV3R50`.  Same
factory, call contract and state_dict keys.  (`mscale2.HRNet` cannot be
constructed in the reference: Basic.__init__ passes `bot_ch=` to
make_seg_head/make_attn_head, mscale2.py:239-242, which take no such argument.)"""
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, initialize_weights
from .deepv3 import get_aspp
from .mynn import Upsample
from .ocrnet import _Base, _nchw
from .utils import SegHead, get_trunk


class PairAttnHead(SegHead):
    """3x3-BN-ReLU-3x3-BN-ReLU-1x1-Sigmoid with children 0..7 (mscale2.py:191-199)."""

    def forward(self, x):
        return ops.backend().sigmoid(super().forward(x))


class MscaleBase(_Base):
    """network/mscale2.py:44-162.  `_fwd(x, size)` returns (fp32 logits at `size`,
    decoder features at stride 4)."""

    def _pair_attn(self, feats, other, size):
        """attention from [feats, other resampled to feats] (mscale2.py:96-101)."""
        B = ops.backend()
        other = B.bilinear(other, feats.shape[1:3])
        return Upsample(self.scale_attn(B.cat([feats, other])), size)

    def nscale_forward(self, inputs, scales):
        """mscale2.py:56-126"""
        B = ops.backend()
        assert 1.0 in scales, "expected 1.0 to be the target scale"
        order = sorted(scales, reverse=True)

        def one_scale(s):
            x, size = self._images(inputs, s)
            return self._fwd(x, size)

        passes = B.parallel([(lambda s=s: one_scale(s)) for s in order])
        pred = attn = last_feats = None
        for idx, (s, (p, feats)) in enumerate(zip(order, passes)):
            if idx > 0:
                attn = self._pair_attn(feats, last_feats, p.shape[1:3])
            if pred is None:
                pred = p
            elif s >= 1.0:
                pred = B.attn_blend(B.bcast_mul(attn, p), attn, B.bilinear(pred, p.shape[1:3]))
            else:
                tgt = pred.shape[1:3]
                p_up = B.bilinear(B.bcast_mul(attn, p), tgt)
                attn = B.bilinear(attn, tgt)
                pred = B.attn_blend(p_up, attn, pred)
            last_feats = feats
        if self.training:
            return self.criterion(_nchw(pred), inputs["gts"])
        return {"pred": _nchw(pred), "attn_10x": _nchw(attn)}

    def two_scale_forward(self, inputs):
        """mscale2.py:128-156"""
        B = ops.backend()

        def lo_pass():
            x_lo, lo_size = self._images(inputs, cfg.MODEL.MSCALE_LO_SCALE)
            return self._fwd(x_lo, lo_size)

        def hi_pass():
            x_1x, size = self._images(inputs)
            return self._fwd(x_1x, size), size

        ((p_1x, feats_hi), size), (p_lo, feats_lo) = B.parallel([hi_pass, lo_pass])
        logit_attn = self._pair_attn(feats_lo, feats_hi, p_lo.shape[1:3])
        p_lo = B.bilinear(B.bcast_mul(logit_attn, p_lo), size)
        logit_attn = B.bilinear(logit_attn, size)
        joint_pred = B.attn_blend(p_lo, logit_attn, p_1x)
        if self.training:
            return self.criterion(_nchw(joint_pred), inputs["gts"])
        return {"pred": _nchw(joint_pred), "attn_10x": _nchw(logit_attn)}

    def forward(self, inputs):
        B = ops.backend()
        B.begin_step(inputs["images"].device)
        if cfg.MODEL.N_SCALES and not self.training:
            out = self.nscale_forward(inputs, cfg.MODEL.N_SCALES)
        else:
            out = self.two_scale_forward(inputs)
        B.end_forward()
        return out


class MscaleV3Plus(MscaleBase):
    """mscale2.py:165-224"""

    def __init__(self, num_classes, trunk="resnet-50", criterion=None):
        super().__init__()
        self.criterion = criterion
        self.backbone, s2_ch, _s4_ch, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=256, output_stride=8)
        self.bot_fine = Conv2d(s2_ch, 48, kernel_size=1, bias=False)
        self.bot_aspp = Conv2d(aspp_out_ch, 256, kernel_size=1, bias=False)
        self.final = SegHead(
            Conv2d(256 + 48, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, num_classes, kernel_size=1, bias=False))
        self.scale_attn = PairAttnHead(
            Conv2d(2 * (256 + 48), 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, 1, kernel_size=1, bias=False), nn.Sigmoid())
        if cfg.OPTIONS.INIT_DECODER:
            initialize_weights(self.bot_fine, self.bot_aspp, self.scale_attn, self.final)
        else:
            initialize_weights(self.final)

    def _fwd(self, x, size):
        B = ops.backend()
        s2_features, _, final_features = self.backbone(x)
        conv_aspp = self.bot_aspp(self.aspp(final_features))
        conv_s2 = self.bot_fine(s2_features)
        cat_s4 = B.cat([conv_s2, B.bilinear(conv_aspp, s2_features.shape[1:3])])
        return Upsample(self.final(cat_s4), size), cat_s4


def DeepV3R50(num_classes, criterion):
    return MscaleV3Plus(num_classes, trunk="resnet-50", criterion=criterion)
