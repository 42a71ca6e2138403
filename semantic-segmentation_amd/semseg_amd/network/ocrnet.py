"""HRNet-OCR and the hierarchical multi-scale-attention model (MscaleOCR).

Drop-in for the reference's network/ocrnet.py: same factories
(`HRNet`, `HRNet_Mscale`), same call contract (`net({'images': NCHW fp32,
'gts': [B,H,W] int64})` -> scalar loss in training, dict of NCHW-shaped
predictions in eval), same state_dict.  Internally everything is NHWC bf16 on
the HIP kernels; the NCHW tensors handed back are zero-copy permuted views.
"""
import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, BNReLU, conv_bn, initialize_weights
from .mynn import Upsample, resized_hw
from .ocr_utils import SpatialGather_Module, SpatialOCR_Module
from .utils import get_trunk, make_attn_head


def fmt_scale(prefix, scale):
    """utils/misc.py:503-513"""
    return "{}_{}x".format(prefix, str(float(scale)))


def _nchw(t):
    return t.permute(0, 3, 1, 2)


class OCR_block(nn.Module):
    """network/ocrnet.py:42-91"""

    def __init__(self, high_level_ch):
        super().__init__()
        mid = cfg.MODEL.OCR.MID_CHANNELS
        key = cfg.MODEL.OCR.KEY_CHANNELS
        num_classes = cfg.DATASET.NUM_CLASSES
        self.conv3x3_ocr = nn.Sequential(
            Conv2d(high_level_ch, mid, kernel_size=3, stride=1, padding=1), BNReLU(mid))
        self.ocr_gather_head = SpatialGather_Module(num_classes)
        self.ocr_distri_head = SpatialOCR_Module(in_channels=mid, key_channels=key, out_channels=mid,
                                                 scale=1, dropout=0.05)
        self.cls_head = Conv2d(mid, num_classes, kernel_size=1, stride=1, padding=0, bias=True)
        self.aux_head = nn.Sequential(
            Conv2d(high_level_ch, high_level_ch, kernel_size=1, stride=1, padding=0),
            BNReLU(high_level_ch),
            Conv2d(high_level_ch, num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        if cfg.OPTIONS.INIT_DECODER:
            initialize_weights(self.conv3x3_ocr, self.ocr_gather_head, self.ocr_distri_head,
                               self.cls_head, self.aux_head)

    def forward(self, high_level_features):
        """high_level_features: a tensor, or a list of tensors (the scale passes in lockstep);
        every output is a list for a list."""
        # Tensors with several consumers go through ops.fan_out: the HIP backend then sums their gradients in ONE
        # grouped launch per tensor set instead of autograd's k - 1 element-wise adds (0.4 ms of torch kernels per
        # step in round 3: the trunk output has 2 consumers, feats 3, ocr_feats 2)
        B = ops.backend()
        multi = isinstance(high_level_features, (list, tuple))
        hl = list(high_level_features) if multi else [high_level_features]
        n = len(hl)
        pick = (lambda hs, j: [h[j] for h in hs]) if multi else (lambda hs, j: hs[0][j])
        hh = B.fan_out(hl, [2] * n)
        # context and feats are written side by side into one buffer per scale pass: their concatenation in front of the
        # 1x1 bottleneck (network/ocr_utils.py:151) is then that buffer (HIP backend; None elsewhere)
        mid = self.conv3x3_ocr[0].out_channels
        slots = B.cat_slots(hl, (mid, mid))
        ctx_out = feats_out = None
        if slots is not None:
            ctx_out = [s_[0] for s_ in slots] if multi else slots[0][0]
            feats_out = [s_[1] for s_ in slots] if multi else slots[0][1]
        # (the auxiliary head as a parallel branch of the 3x3 conv -- ops.fork -- measured +0.5 ms per step: two
        # MFMA-bound GEMMs sharing the chip, profiles/r06_notes.md call S)
        feats = conv_bn(self.conv3x3_ocr[0], self.conv3x3_ocr[1][0], pick(hh, 0), relu=True, out=feats_out)
        aux = conv_bn(self.aux_head[0], self.aux_head[1][0], pick(hh, 1), relu=True)
        aux_out = self.aux_head[2](aux, out_f32=True)              # [B,H,W,K] fp32
        fh = B.fan_out(list(feats) if multi else [feats], [3] * n)
        context = self.ocr_gather_head(pick(fh, 0), aux_out)
        ocr_feats = self.ocr_distri_head(pick(fh, 1), context, feats_cat=pick(fh, 2), context_out=ctx_out)
        oh = B.fan_out(list(ocr_feats) if multi else [ocr_feats], [2] * n)
        cls_out = self.cls_head(pick(oh, 0), out_f32=True)         # [B,H,W,K] fp32
        return cls_out, aux_out, pick(oh, 1)


class _Base(nn.Module):
    def _images(self, inputs, scale=None):
        assert "images" in inputs
        x = inputs["images"]
        h, w = x.shape[2], x.shape[3]
        size = (h, w) if scale is None or scale == 1.0 else resized_hw(h, w, scale)
        return ops.backend().image_to_nhwc(x, size), size


class OCRNet(_Base):
    """network/ocrnet.py:94-122"""

    def __init__(self, num_classes, trunk="hrnetv2", criterion=None):
        super().__init__()
        self.criterion = criterion
        self.backbone, _, _, high_level_ch = get_trunk(trunk)
        self.ocr = OCR_block(high_level_ch)

    def forward(self, inputs):
        ops.backend().begin_step(inputs["images"].device)
        x, size = self._images(inputs)
        _, _, feats = self.backbone(x)
        cls_out, aux_out, _ = self.ocr(feats)
        aux_out = Upsample(aux_out, size)
        cls_out = Upsample(cls_out, size)
        ops.backend().end_forward()
        if self.training:
            gts = inputs["gts"]
            aux_loss = self.criterion(_nchw(aux_out), gts, do_rmi=cfg.LOSS.OCR_AUX_RMI)
            main_loss = self.criterion(_nchw(cls_out), gts)
            return cfg.LOSS.OCR_ALPHA * aux_loss + main_loss
        return {"pred": _nchw(cls_out)}


class OCRNetASPP(_Base):
    """OCR head on ASPP features (network/ocrnet.py:125-155)."""

    def __init__(self, num_classes, trunk="hrnetv2", criterion=None):
        super().__init__()
        from .deepv3 import get_aspp
        self.criterion = criterion
        self.backbone, _, _, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=256, output_stride=8)
        self.ocr = OCR_block(aspp_out_ch)

    def forward(self, inputs):
        ops.backend().begin_step(inputs["images"].device)
        x, size = self._images(inputs)
        _, _, feats = self.backbone(x)
        cls_out, aux_out, _ = self.ocr(self.aspp(feats))
        aux_out = Upsample(aux_out, size)
        cls_out = Upsample(cls_out, size)
        ops.backend().end_forward()
        if self.training:
            gts = inputs["gts"]
            return cfg.LOSS.OCR_ALPHA * self.criterion(_nchw(aux_out), gts) + self.criterion(_nchw(cls_out), gts)
        return {"pred": _nchw(cls_out)}


class MscaleOCR(_Base):
    """network/ocrnet.py:158-334"""

    def __init__(self, num_classes, trunk="hrnetv2", criterion=None):
        super().__init__()
        self.criterion = criterion
        self.backbone, _, _, high_level_ch = get_trunk(trunk)
        self.ocr = OCR_block(high_level_ch)
        self.scale_attn = make_attn_head(in_ch=cfg.MODEL.OCR.MID_CHANNELS, out_ch=1)

    def _fwd(self, xs, sizes):
        """Trunk + heads for every scale pass in `xs`, in lockstep (the passes are problems of the
        same grouped launches); per pass the outputs are bilinearly resampled to sizes[p] in fp32
        (network/ocrnet.py:170-183)."""
        _, _, feats = self.backbone(list(xs))
        cls_out, aux_out, mid = self.ocr(feats)
        attn = self.scale_attn(mid)
        cls_up = Upsample(cls_out, sizes)
        aux_up = Upsample(aux_out, sizes)
        attn_up = Upsample(attn, sizes)
        return [{"cls_out": c, "aux_out": a, "logit_attn": t} for c, a, t in zip(cls_up, aux_up, attn_up)]

    def nscale_forward(self, inputs, scales):
        """Hierarchical attention over N scales, high to low (network/ocrnet.py:185-262)."""
        B = ops.backend()
        assert 1.0 in scales, "expected 1.0 to be the target scale"
        pred = aux = None
        out = {}
        order = sorted(scales, reverse=True)
        # the per-scale passes are independent (only the fusion below is sequential): lockstep
        imgs = [self._images(inputs, s) for s in order]
        passes = self._fwd([x for x, _ in imgs], [size for _, size in imgs])
        for s, o in zip(order, passes):
            cls_out, attn_out, aux_out = o["cls_out"], o["logit_attn"], o["aux_out"]
            out[fmt_scale("pred", s)] = _nchw(cls_out)
            if s != 2.0:
                out[fmt_scale("attn", s)] = _nchw(attn_out)
            if pred is None:
                pred, aux = cls_out, aux_out
            elif s >= 1.0:
                tgt = cls_out.shape[1:3]
                pred = B.attn_blend(B.bcast_mul(attn_out, cls_out), attn_out, B.bilinear(pred, tgt))
                aux = B.attn_blend(B.bcast_mul(attn_out, aux_out), attn_out, B.bilinear(aux, tgt))
            else:
                tgt = pred.shape[1:3]
                cls_up = B.bilinear(B.bcast_mul(attn_out, cls_out), tgt)
                aux_up = B.bilinear(B.bcast_mul(attn_out, aux_out), tgt)
                attn_up = B.bilinear(attn_out, tgt)
                pred = B.attn_blend(cls_up, attn_up, pred)
                aux = B.attn_blend(aux_up, attn_up, aux)
        if self.training:
            gts = inputs["gts"]
            return cfg.LOSS.OCR_ALPHA * self.criterion(_nchw(aux), gts) + self.criterion(_nchw(pred), gts)
        out["pred"] = _nchw(pred)
        return out

    def two_scale_forward(self, inputs):
        """Training path: 0.5x and 1.0x passes fused by the 0.5x attention
        (network/ocrnet.py:264-327).  The two passes share nothing but the weights: they run in
        lockstep, low scale first (the reference's order: BatchNorm running statistics)."""
        B = ops.backend()
        x_lo, lo_size = self._images(inputs, cfg.MODEL.MSCALE_LO_SCALE)
        x_1x, size = self._images(inputs)
        lo, hi = self._fwd([x_lo, x_1x], [lo_size, size])
        pred_05x, aux_lo, attn_05x = lo["cls_out"], lo["aux_out"], lo["logit_attn"]
        pred_10x, aux_1x = hi["cls_out"], hi["aux_out"]

        p_lo = B.bilinear(B.bcast_mul(attn_05x, pred_05x), size)
        a_lo = B.bilinear(B.bcast_mul(attn_05x, aux_lo), size)
        attn_up = B.bilinear(attn_05x, size)
        joint_pred = B.attn_blend(p_lo, attn_up, pred_10x)
        joint_aux = B.attn_blend(a_lo, attn_up, aux_1x)

        if self.training:
            gts = inputs["gts"]
            # the RMI term is a chain of small launches (pool -> Gram -> 9x9 solves): a parallel branch next to the
            # three BCE-only terms (ops.fork)
            join_main = B.fork(lambda: self.criterion(_nchw(joint_pred), gts, do_rmi=True), tag="loss", has_bn=False)
            aux_loss = self.criterion(_nchw(joint_aux), gts, do_rmi=cfg.LOSS.OCR_AUX_RMI)
            wt = cfg.LOSS.SUPERVISED_MSCALE_WT
            if wt:
                loss_lo = self.criterion(_nchw(B.bilinear(pred_05x, size)), gts, do_rmi=False)
                loss_hi = self.criterion(_nchw(pred_10x), gts, do_rmi=False)
            loss = cfg.LOSS.OCR_ALPHA * aux_loss + join_main()
            if wt:
                loss = loss + wt * loss_lo + wt * loss_hi
            return loss
        return {"pred": _nchw(joint_pred), "pred_05x": _nchw(pred_05x), "pred_10x": _nchw(pred_10x),
                "attn_05x": _nchw(attn_05x)}

    def forward(self, inputs):
        ops.backend().begin_step(inputs["images"].device)
        if cfg.MODEL.N_SCALES and not self.training:
            out = self.nscale_forward(inputs, cfg.MODEL.N_SCALES)
        else:
            out = self.two_scale_forward(inputs)
        ops.backend().end_forward()       # deferred BN running-statistics updates
        return out


def HRNet(num_classes, criterion):
    return OCRNet(num_classes, trunk="hrnetv2", criterion=criterion)


def HRNet_Mscale(num_classes, criterion):
    return MscaleOCR(num_classes, trunk="hrnetv2", criterion=criterion)
