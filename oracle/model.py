"""ORACLE -- test infrastructure only.  Never imported by the product package.

Functional CPU restatement (plain torch, fp32, NCHW) of the reference's
HRNet-OCR-MScale network: `network.ocrnet.MscaleOCR` over `network.hrnetv2`
with `network.ocr_utils` and `network.utils.make_attn_head`.  It is driven by a
state_dict with the REFERENCE's parameter names (1,903 keys), so a reference
checkpoint runs unchanged.  Pinned against the real reference by
tests/golden/make_golden.py + tests/test_oracle_golden.py.
"""
import torch

from . import ops as O

# cfg.MODEL.OCR_EXTRA, config.py:161-190: (num_modules, num_blocks per branch, channels per branch)
STAGES = {
    "stage2": (1, [4, 4], [48, 96]),
    "stage3": (4, [4, 4, 4], [48, 96, 192]),
    "stage4": (3, [4, 4, 4, 4], [48, 96, 192, 384]),
}
OCR_ALPHA = 0.4        # cfg.LOSS.OCR_ALPHA, config.py:151
KEY_CHANNELS = 256     # cfg.MODEL.OCR.KEY_CHANNELS, config.py:158


class Net:
    """Holds the parameters and the mode; every method restates one reference block."""

    def __init__(self, sd, num_classes=19, training=True, dropout_mask=None, mscale_wt=0.0,
                 aux_rmi=False, criterion="rmi", ignore_index=255, used=None, bn_momentum=0.1):
        self.sd = sd
        self.nc = num_classes
        self.training = training
        self.dropout_mask = dropout_mask   # dict: pass index -> [B,512] multiplier, or None (p=0)
        self.mscale_wt = mscale_wt
        self.aux_rmi = aux_rmi
        self.criterion_kind = criterion
        self.ignore_index = ignore_index
        self.used = used if used is not None else set()
        self._pass = 0
        self.bn_momentum = bn_momentum

    # -- primitives
    def p(self, name):
        self.used.add(name)
        return self.sd[name]

    def conv(self, x, name, stride=1, padding=0, bias=False):
        b = self.p(name + ".bias") if bias else None
        return O.conv2d(x, self.p(name + ".weight"), b, stride, padding)

    def bn(self, x, name, relu=False):
        self.used.add(name + ".num_batches_tracked")
        y = O.batch_norm(x, self.p(name + ".weight"), self.p(name + ".bias"),
                         self.p(name + ".running_mean"), self.p(name + ".running_var"),
                         self.training, self.bn_momentum, 1e-5)
        return torch.relu(y) if relu else y

    # -- network/hrnetv2.py:37-66
    def basic_block(self, x, pre):
        out = self.bn(self.conv(x, pre + ".conv1", 1, 1), pre + ".bn1", relu=True)
        out = self.bn(self.conv(out, pre + ".conv2", 1, 1), pre + ".bn2")
        return torch.relu(out + x)

    # -- network/hrnetv2.py:69-106
    def bottleneck(self, x, pre, downsample):
        out = self.bn(self.conv(x, pre + ".conv1"), pre + ".bn1", relu=True)
        out = self.bn(self.conv(out, pre + ".conv2", 1, 1), pre + ".bn2", relu=True)
        out = self.bn(self.conv(out, pre + ".conv3"), pre + ".bn3")
        res = x
        if downsample:
            res = self.bn(self.conv(x, pre + ".downsample.0"), pre + ".downsample.1")
        return torch.relu(out + res)

    # -- network/hrnetv2.py:109-254
    def hr_module(self, xs, pre, num_blocks):
        nb = len(xs)
        xs = list(xs)
        for i in range(nb):
            for k in range(num_blocks[i]):
                xs[i] = self.basic_block(xs[i], "%s.branches.%d.%d" % (pre, i, k))
        outs = []
        for i in range(nb):
            y = None
            for j in range(nb):
                fp = "%s.fuse_layers.%d.%d" % (pre, i, j)
                if j == i:
                    t = xs[j]
                elif j > i:
                    t = self.bn(self.conv(xs[j], fp + ".0"), fp + ".1")
                    t = O.bilinear(t, xs[i].shape[-2:])
                else:
                    t = xs[j]
                    for k in range(i - j):
                        last = k == i - j - 1
                        t = self.bn(self.conv(t, "%s.%d.0" % (fp, k), 2, 1), "%s.%d.1" % (fp, k), relu=not last)
                y = t if y is None else y + t
            outs.append(torch.relu(y))
        return outs

    # -- network/hrnetv2.py:399-449
    def backbone(self, x):
        bb = "backbone."
        x = self.bn(self.conv(x, bb + "conv1", 2, 1), bb + "bn1", relu=True)
        x = self.bn(self.conv(x, bb + "conv2", 2, 1), bb + "bn2", relu=True)
        for i in range(4):
            x = self.bottleneck(x, bb + "layer1.%d" % i, downsample=(i == 0))
        # transition1 (hrnetv2.py:319-350)
        t = bb + "transition1"
        xs = [self.bn(self.conv(x, t + ".0.0", 1, 1), t + ".0.1", relu=True),
              self.bn(self.conv(x, t + ".1.0.0", 2, 1), t + ".1.0.1", relu=True)]
        for stage, trans, new_idx in (("stage2", None, None), ("stage3", "transition2", 2),
                                      ("stage4", "transition3", 3)):
            nmod, nblocks, _ = STAGES[stage]
            if trans is not None:
                tp = "%s%s.%d.0" % (bb, trans, new_idx)
                xs = xs + [self.bn(self.conv(xs[-1], tp + ".0", 2, 1), tp + ".1", relu=True)]
            for m in range(nmod):
                xs = self.hr_module(xs, "%s%s.%d" % (bb, stage, m), nblocks)
        size = xs[0].shape[-2:]
        return torch.cat([xs[0]] + [O.bilinear(t, size) for t in xs[1:]], 1)

    def bnrelu_seq(self, x, conv_name, bn_name, **kw):
        return self.bn(self.conv(x, conv_name, **kw), bn_name, relu=True)

    # -- network/ocrnet.py:85-91 + network/ocr_utils.py
    def ocr(self, feats_in):
        o = "ocr."
        feats = self.bnrelu_seq(feats_in, o + "conv3x3_ocr.0", o + "conv3x3_ocr.1.0", stride=1, padding=1, bias=True)
        aux = self.bnrelu_seq(feats_in, o + "aux_head.0", o + "aux_head.1.0", bias=True)
        aux = self.conv(aux, o + "aux_head.2", bias=True)
        context = O.spatial_gather(feats, aux)                       # [B,512,K,1]
        ob = o + "ocr_distri_head.object_context_block."
        B, _, H, W = feats.shape
        q = self.bnrelu_seq(feats, ob + "f_pixel.0", ob + "f_pixel.1.0")
        q = self.bnrelu_seq(q, ob + "f_pixel.2", ob + "f_pixel.3.0")
        k = self.bnrelu_seq(context, ob + "f_object.0", ob + "f_object.1.0")
        k = self.bnrelu_seq(k, ob + "f_object.2", ob + "f_object.3.0")
        v = self.bnrelu_seq(context, ob + "f_down.0", ob + "f_down.1.0")
        query = q.view(B, KEY_CHANNELS, -1).permute(0, 2, 1)
        key = k.view(B, KEY_CHANNELS, -1)
        value = v.view(B, KEY_CHANNELS, -1).permute(0, 2, 1)
        ctx = O.object_attention(query, key, value, KEY_CHANNELS)
        ctx = ctx.permute(0, 2, 1).contiguous().view(B, KEY_CHANNELS, H, W)
        ctx = self.bnrelu_seq(ctx, ob + "f_up.0", ob + "f_up.1.0")
        cat = torch.cat([ctx, feats], 1)
        out = self.bnrelu_seq(cat, o + "ocr_distri_head.conv_bn_dropout.0", o + "ocr_distri_head.conv_bn_dropout.1.0")
        if self.training and self.dropout_mask is not None:
            out = out * self.dropout_mask[self._pass][:, :, None, None]
        cls = self.conv(out, o + "cls_head", bias=True)
        return cls, aux, out

    # -- network/utils.py:343-367
    def attn_head(self, x):
        a = "scale_attn."
        x = self.bn(self.conv(x, a + "conv0", 1, 1), a + "bn0", relu=True)
        x = self.bn(self.conv(x, a + "conv1", 1, 1), a + "bn1", relu=True)
        return torch.sigmoid(self.conv(x, a + "conv2"))

    # -- network/ocrnet.py:170-183
    def fwd(self, x):
        size = x.shape[-2:]
        feats = self.backbone(x)
        cls, aux, mid = self.ocr(feats)
        attn = self.attn_head(mid)
        self._pass += 1
        return {"cls_out": O.bilinear(cls, size), "aux_out": O.bilinear(aux, size),
                "logit_attn": O.bilinear(attn, size)}

    def criterion(self, logits, gts, do_rmi=None):
        if self.criterion_kind == "rmi":
            return O.rmi_loss(logits, gts, self.nc, do_rmi=True if do_rmi is None else do_rmi)
        return O.cross_entropy(logits, gts, self.ignore_index)

    # -- network/ocrnet.py:264-327
    def two_scale_forward(self, images, gts=None, lo_scale=0.5):
        self._pass = 0
        x_lo = O.resize_x(images, lo_scale)
        lo = self.fwd(x_lo)
        pred_05x, aux_lo, attn = lo["cls_out"], lo["aux_out"], lo["logit_attn"]
        hi = self.fwd(images)
        pred_10x, aux_1x = hi["cls_out"], hi["aux_out"]
        size = pred_10x.shape[-2:]
        p_lo = O.bilinear(attn * pred_05x, size)
        a_lo = O.bilinear(attn * aux_lo, size)
        attn_up = O.bilinear(attn, size)
        joint_pred = p_lo + (1 - attn_up) * pred_10x
        joint_aux = a_lo + (1 - attn_up) * aux_1x
        if not self.training:
            return {"pred": joint_pred, "pred_05x": pred_05x, "pred_10x": pred_10x, "attn_05x": attn}
        aux_loss = self.criterion(joint_aux, gts, do_rmi=self.aux_rmi)
        main_loss = self.criterion(joint_pred, gts, do_rmi=True)
        loss = OCR_ALPHA * aux_loss + main_loss
        if self.mscale_wt:
            loss = loss + self.mscale_wt * self.criterion(O.bilinear(pred_05x, size), gts, do_rmi=False)
            loss = loss + self.mscale_wt * self.criterion(pred_10x, gts, do_rmi=False)
        return loss

    # -- network/ocrnet.py:185-262 (eval only)
    def nscale_forward(self, images, scales):
        assert 1.0 in scales
        self._pass = 0
        pred = aux = None
        out = {}
        for s in sorted(scales, reverse=True):
            x = O.resize_x(images, s) if s != 1.0 else O.resize_x(images, 1.0)
            o = self.fwd(x)
            cls, attn, aux_out = o["cls_out"], o["logit_attn"], o["aux_out"]
            out["pred_%sx" % s] = cls
            if s != 2.0:
                out["attn_%sx" % s] = attn
            if pred is None:
                pred, aux = cls, aux_out
            elif s >= 1.0:
                pred = O.bilinear(pred, cls.shape[-2:])
                pred = attn * cls + (1 - attn) * pred
                aux = O.bilinear(aux, cls.shape[-2:])
                aux = attn * aux_out + (1 - attn) * aux
            else:
                size = pred.shape[-2:]
                cls, aux_out = O.bilinear(attn * cls, size), O.bilinear(attn * aux_out, size)
                attn = O.bilinear(attn, size)
                pred = cls + (1 - attn) * pred
                aux = aux_out + (1 - attn) * aux
        out["pred"] = pred
        return out


def seeded_state_dict(shapes, seed=0):
    """Deterministic parity weights, identical for the reference, the oracle and
    the HIP model: per key (in the given order) conv weights ~ N(0, 2/fan_in)
    (kaiming scale -- the reference's own trunk init N(0, 1e-3) makes every
    logit ~0 and hides errors), BN gamma in
    [0.5, 1.5], small biases/betas, running stats near (0, 1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes:
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = torch.randn(shape, generator=g) * 0.05
        elif name.endswith("running_var"):
            sd[name] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif name.endswith(".weight"):
            sd[name] = torch.rand(shape, generator=g) + 0.5
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.05
    return sd
