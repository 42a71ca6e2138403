"""Multi-scale-attention models without the OCR head (network/mscale.py of the
reference): `mscale.HRNet` (MscaleBasic), `mscale.HRNet_ASP` (ASPP) and
`mscale.DeepV3R50` (MscaleV3Plus) -- the sibling `--arch` names of SURVEY.md 8f
rank 4 -- on the same HIP operator surface as ocrnet.MscaleOCR.  Same factories,
call contract and state_dict keys; NHWC bf16 inside.

Not carried over: `nscale_fused_forward` (mscale.py:95-122) -- the reference's
`recurse_fuse_fwd` passes `attn_lo=` to `_fwd` (mscale.py:77), which none of its
models accepts, so that entry point raises TypeError there too; MscaleDeeper and
the wrn38 / xception71 / efficientnet factories need trunks outside
BASELINE.json's configs (get_trunk lists the supported ones)."""
from .. import ops
from ..config import cfg
from ..nn import Conv2d, initialize_weights
from .deepv3 import get_aspp
from .mynn import Upsample
from .ocrnet import _Base, _nchw, fmt_scale
from .utils import get_trunk, make_attn_head, make_seg_head


class MscaleBase(_Base):
    """network/mscale.py:41-229.  `_fwd(x, size, aspp_lo, aspp_attn)` returns
    (logits, attention, aspp attention, aspp features); logits and attention are
    fp32 [B,H,W,*] at `size` (the resolution of the pass's input)."""
    fuse_aspp = False

    def _fwd(self, x, size, aspp_lo=None, aspp_attn=None):
        raise NotImplementedError

    def nscale_forward(self, inputs, scales):
        """Hierarchical attention, evaluated high to low (mscale.py:124-186)."""
        B = ops.backend()
        assert 1.0 in scales, "expected 1.0 to be the target scale"
        order = sorted(scales, reverse=True)

        def one_scale(s):
            x, size = self._images(inputs, s)
            return self._fwd(x, size)

        passes = B.parallel([(lambda s=s: one_scale(s)) for s in order])
        pred = None
        out = {}
        for s, (p, attn, _, _) in zip(order, passes):
            out[fmt_scale("pred", s)] = _nchw(p)
            if s != 2.0:
                out[fmt_scale("attn", s)] = _nchw(attn)
            if pred is None:
                pred = p
            elif s >= 1.0:
                pred = B.attn_blend(B.bcast_mul(attn, p), attn, B.bilinear(pred, p.shape[1:3]))
            else:
                tgt = pred.shape[1:3]
                pred = B.attn_blend(B.bilinear(B.bcast_mul(attn, p), tgt), B.bilinear(attn, tgt), pred)
        if self.training:
            return self.criterion(_nchw(pred), inputs["gts"])
        out["pred"] = _nchw(pred)
        return out

    def two_scale_forward(self, inputs):
        """mscale.py:188-228"""
        B = ops.backend()

        def lo_pass():
            x_lo, lo_size = self._images(inputs, cfg.MODEL.MSCALE_LO_SCALE)
            return self._fwd(x_lo, lo_size)

        def hi_pass(aspp_lo=None, aspp_attn=None):
            x_1x, size = self._images(inputs)
            return self._fwd(x_1x, size, aspp_lo=aspp_lo, aspp_attn=aspp_attn), size

        if self.fuse_aspp:      # the 1.0x pass consumes the 0.5x pass's ASPP features: sequential
            lo = lo_pass()
            hi, size = hi_pass(aspp_lo=lo[3], aspp_attn=lo[2])
        else:                   # independent passes: concurrent streams
            (hi, size), lo = B.parallel([hi_pass, lo_pass])
        pred_05x, attn_05x = lo[0], lo[1]
        p_1x = hi[0]
        p_lo = B.bilinear(B.bcast_mul(attn_05x, pred_05x), size)
        logit_attn = B.bilinear(attn_05x, size)
        joint_pred = B.attn_blend(p_lo, logit_attn, p_1x)
        if self.training:
            gts = inputs["gts"]
            loss = self.criterion(_nchw(joint_pred), gts)
            wt = cfg.LOSS.SUPERVISED_MSCALE_WT
            if wt:
                loss_lo = self.criterion(_nchw(B.bilinear(pred_05x, size)), gts, do_rmi=False)
                loss_hi = self.criterion(_nchw(p_1x), gts, do_rmi=False)
                loss = loss + wt * loss_lo + wt * loss_hi
            return loss
        return {"pred": _nchw(joint_pred), "pred_05x": _nchw(pred_05x), "pred_10x": _nchw(p_1x),
                "attn_05x": _nchw(attn_05x)}

    def forward(self, inputs):
        B = ops.backend()
        B.begin_step(inputs["images"].device)
        if cfg.MODEL.N_SCALES and not self.training:
            if self.fuse_aspp:
                raise NotImplementedError("nscale_fused_forward: unusable in the reference as well (mscale.py:77)")
            out = self.nscale_forward(inputs, cfg.MODEL.N_SCALES)
        else:
            out = self.two_scale_forward(inputs)
        B.end_forward()
        return out


class MscaleV3Plus(MscaleBase):
    """DeepLabV3+-based multi-scale model (mscale.py:232-328)."""

    def __init__(self, num_classes, trunk="resnet-50", criterion=None, use_dpc=False, fuse_aspp=False,
                 attn_2b=False):
        super().__init__()
        self.criterion = criterion
        self.fuse_aspp = fuse_aspp
        self.attn_2b = attn_2b
        self.backbone, s2_ch, _s4_ch, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=256, output_stride=8, dpc=use_dpc)
        self.bot_fine = Conv2d(s2_ch, 48, kernel_size=1, bias=False)
        self.bot_aspp = Conv2d(aspp_out_ch, 256, kernel_size=1, bias=False)
        self.final = make_seg_head(256 + 48, num_classes)
        self.scale_attn = make_attn_head(in_ch=256 + 48, out_ch=2 if attn_2b else 1)
        if cfg.OPTIONS.INIT_DECODER:
            initialize_weights(self.bot_fine, self.bot_aspp, self.scale_attn, self.final)
        else:
            initialize_weights(self.final)

    def _fwd(self, x, size, aspp_lo=None, aspp_attn=None):
        B = ops.backend()
        s2_features, _, final_features = self.backbone(x)
        aspp = self.aspp(final_features)
        if self.fuse_aspp and aspp_lo is not None and aspp_attn is not None:
            tgt = aspp.shape[1:3]
            a = B.bilinear(aspp_attn, tgt)
            aspp = B.to_act(B.attn_blend(B.bcast_mul(a, B.bilinear(aspp_lo, tgt)), a, aspp))
        conv_aspp = self.bot_aspp(aspp)
        conv_s2 = self.bot_fine(s2_features)
        conv_aspp = B.bilinear(conv_aspp, s2_features.shape[1:3])
        cat_s4 = B.cat([conv_s2, conv_aspp])
        out = Upsample(self.final(cat_s4), size)
        scale_attn = Upsample(self.scale_attn(cat_s4), size)
        if self.attn_2b:
            return out, scale_attn[..., 0:1], scale_attn[..., 1:], aspp
        return out, scale_attn, scale_attn, aspp


def DeepV3R50(num_classes, criterion):
    return MscaleV3Plus(num_classes, trunk="resnet-50", criterion=criterion)


class MscaleBasic(MscaleBase):
    """Trunk + segmentation head + attention head (mscale.py:450-471)."""

    def __init__(self, num_classes, trunk="hrnetv2", criterion=None):
        super().__init__()
        self.criterion = criterion
        self.backbone, _, _, high_level_ch = get_trunk(trunk_name=trunk, output_stride=8)
        self.cls_head = make_seg_head(in_ch=high_level_ch, out_ch=num_classes)
        self.scale_attn = make_attn_head(in_ch=high_level_ch, out_ch=1)

    def _fwd(self, x, size, aspp_lo=None, aspp_attn=None):
        _, _, final_features = self.backbone(x)
        attn = self.scale_attn(final_features)
        pred = self.cls_head(final_features)
        return Upsample(pred, size), Upsample(attn, size), None, None


def HRNet(num_classes, criterion, s2s4=None):
    return MscaleBasic(num_classes=num_classes, criterion=criterion, trunk="hrnetv2")


class ASPP(MscaleBase):
    """ASPP-based multi-scale model (mscale.py:479-511)."""

    def __init__(self, num_classes, trunk="hrnetv2", criterion=None):
        super().__init__()
        self.criterion = criterion
        self.backbone, _s2_ch, _s4_ch, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=cfg.MODEL.ASPP_BOT_CH, output_stride=8)
        self.bot_aspp = Conv2d(aspp_out_ch, 256, kernel_size=1, bias=False)
        self.final = make_seg_head(in_ch=256, out_ch=num_classes)
        self.scale_attn = make_attn_head(in_ch=256, out_ch=1)
        initialize_weights(self.final)

    def _fwd(self, x, size, aspp_lo=None, aspp_attn=None):
        _, _, final_features = self.backbone(x)
        aspp = self.bot_aspp(self.aspp(final_features))
        out = Upsample(self.final(aspp), size)
        scale_attn = Upsample(self.scale_attn(aspp), size)
        return out, scale_attn, scale_attn, aspp


def HRNet_ASP(num_classes, criterion, s2s4=None):
    return ASPP(num_classes=num_classes, criterion=criterion, trunk="hrnetv2")
