"""HRNet-OCR and the hierarchical multi-scale-attention model (MscaleOCR).

Drop-in for the reference's network/ocrnet.py: same factories
(`HRNet`, `HRNet_Mscale`), same call contract (`net({'images': NCHW fp32,
'gts': [B,H,W] int64})` -> scalar loss in training, dict of NCHW-shaped
predictions in eval), same state_dict.  Internally everything is NHWC bf16 on
the HIP kernels; the NCHW tensors handed back are zero-copy permuted views.
"""
import torch
from torch import nn
from torch.nn.utils import stateless

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
        feats = conv_bn(self.conv3x3_ocr[0], self.conv3x3_ocr[1][0], high_level_features, relu=True)
        aux = conv_bn(self.aux_head[0], self.aux_head[1][0], high_level_features, relu=True)
        aux_out = self.aux_head[2](aux, out_f32=True)              # [B,H,W,K] fp32
        context = self.ocr_gather_head(feats, aux_out)
        ocr_feats = self.ocr_distri_head(feats, context)
        cls_out = self.cls_head(ocr_feats, out_f32=True)           # [B,H,W,K] fp32
        return cls_out, aux_out, ocr_feats


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

    def _fwd(self, x, size):
        """One trunk+heads pass; outputs bilinearly resampled to `size` in fp32
        (network/ocrnet.py:170-183)."""
        _, _, feats = self.backbone(x)
        cls_out, aux_out, mid = self.ocr(feats)
        attn = self.scale_attn(mid)
        return {"cls_out": Upsample(cls_out, size), "aux_out": Upsample(aux_out, size),
                "logit_attn": Upsample(attn, size)}

    def nscale_forward(self, inputs, scales):
        """Hierarchical attention over N scales, high to low (network/ocrnet.py:185-262)."""
        B = ops.backend()
        assert 1.0 in scales, "expected 1.0 to be the target scale"
        pred = aux = None
        out = {}
        order = sorted(scales, reverse=True)

        def one_scale(s):
            x, size = self._images(inputs, s)
            return self._fwd(x, size)

        # the per-scale passes are independent (only the fusion below is sequential): issue them
        # on concurrent streams, the target scale on the calling stream
        passes = B.parallel([(lambda s=s: one_scale(s)) for s in order])
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
        (network/ocrnet.py:264-327)."""
        B = ops.backend()

        shadow = None
        if self.training and torch.is_grad_enabled() and getattr(B, "use_shadow_pass", lambda: False)():
            shadow = self._shadow_parameters()

        def lo_pass():
            x_lo, lo_size = self._images(inputs, cfg.MODEL.MSCALE_LO_SCALE)
            if shadow is None:
                return self._fwd(x_lo, lo_size)
            with stateless._reparametrize_module(self, shadow):
                return self._fwd(x_lo, lo_size)

        def hi_pass():
            x_1x, size = self._images(inputs)
            return self._fwd(x_1x, size), size

        # the two scale passes share nothing but the weights: run them concurrently
        (hi, size), lo = B.parallel([hi_pass, lo_pass])
        pred_05x, aux_lo, attn_05x = lo["cls_out"], lo["aux_out"], lo["logit_attn"]
        pred_10x, aux_1x = hi["cls_out"], hi["aux_out"]

        p_lo = B.bilinear(B.bcast_mul(attn_05x, pred_05x), size)
        a_lo = B.bilinear(B.bcast_mul(attn_05x, aux_lo), size)
        attn_up = B.bilinear(attn_05x, size)
        joint_pred = B.attn_blend(p_lo, attn_up, pred_10x)
        joint_aux = B.attn_blend(a_lo, attn_up, aux_1x)

        if self.training:
            gts = inputs["gts"]
            aux_loss = self.criterion(_nchw(joint_aux), gts, do_rmi=cfg.LOSS.OCR_AUX_RMI)
            main_loss = self.criterion(_nchw(joint_pred), gts, do_rmi=True)
            loss = cfg.LOSS.OCR_ALPHA * aux_loss + main_loss
            wt = cfg.LOSS.SUPERVISED_MSCALE_WT
            if wt:
                loss_lo = self.criterion(_nchw(B.bilinear(pred_05x, size)), gts, do_rmi=False)
                loss_hi = self.criterion(_nchw(pred_10x), gts, do_rmi=False)
                loss = loss + wt * loss_lo + wt * loss_hi
            if shadow is not None and loss.requires_grad:
                loss.register_hook(self._arm_shadow_merge)
            return loss
        return {"pred": _nchw(joint_pred), "pred_05x": _nchw(pred_05x), "pred_10x": _nchw(pred_10x),
                "attn_05x": _nchw(attn_05x)}

    # -- gradients of the 0.5x pass ------------------------------------------------------
    # Both passes use every parameter, so autograd would add the two contributions with one
    # `add` kernel per parameter (955 launches per step).  Instead the 0.5x pass runs on detached
    # aliases (same storage, separate autograd leaves) and ONE multi-tensor add merges their
    # gradients into the parameters' at the end of backward.
    def _shadow_parameters(self):
        cur = getattr(self, "_shadow", None)
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        if cur is None or len(cur[0]) != len(named) or any(
                s.data_ptr() != p.data_ptr() or s.shape != p.shape for (_, p), s in zip(named, cur[0].values())):
            cur = ({n: p.detach().requires_grad_(True) for n, p in named}, [p for _, p in named])
            object.__setattr__(self, "_shadow", cur)
        return cur[0]

    def _arm_shadow_merge(self, grad):
        from torch.autograd import Variable
        Variable._execution_engine.queue_callback(self._merge_shadow_grads)
        return None

    def _merge_shadow_grads(self):
        shadows, params = self._shadow
        getattr(ops.backend(), "flush_backward", lambda: None)()    # deferred weight-gradient reduces first
        main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        if main is not None:
            for st in getattr(ops.backend(), "side_streams", lambda: [])():
                main.wait_stream(st)             # the 0.5x pass's gradients were produced there
        dst, src = [], []
        for p, s in zip(params, shadows.values()):
            if s.grad is None:
                continue
            if p.grad is None:
                p.grad = s.grad
            else:
                dst.append(p.grad)
                src.append(s.grad)
            s.grad = None
        if dst:
            torch._foreach_add_(dst, src)

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
