"""Trunk selection and the scale-attention head (network/utils.py:102-141,
343-367 of the reference)."""
from collections import OrderedDict

from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, BNReLU, conv_bn  # noqa: F401  (BNReLU re-exported like the reference)


def get_trunk(trunk_name, output_stride=8):
    """network/utils.py:102-141 -- the trunks of BASELINE.json's configs: HRNetV2-W48
    (hot path) and ResNet-50 (DeepLabV3+ plumbing config)."""
    assert output_stride == 8, "Only stride8 supported right now"
    if trunk_name == "hrnetv2":
        from . import hrnetv2
        backbone = hrnetv2.get_seg_model()
        return backbone, -1, -1, backbone.high_level_ch
    if trunk_name == "resnet-50":
        from .resnet import get_resnet
        return get_resnet(trunk_name, output_stride=output_stride), 256, -1, 2048
    raise ValueError("unsupported trunk {} (supported: hrnetv2, resnet-50)".format(trunk_name))


class AttnHead(nn.Sequential):
    """3x3 -> BN -> ReLU -> 3x3 -> BN -> ReLU -> 1x1 -> sigmoid; children are
    named conv0/bn0/re0/conv1/bn1/re1/conv2/sig as in network/utils.py:348-363."""

    def forward(self, x):
        B = ops.backend()
        x = conv_bn(self.conv0, self.bn0, x, relu=True)
        if hasattr(self, "conv1"):
            x = conv_bn(self.conv1, self.bn1, x, relu=True)
        y = self.conv2(x, out_f32=True)                    # [B,H,W,1] fp32 (a list for a list)
        return [B.sigmoid(t) for t in y] if isinstance(y, (list, tuple)) else B.sigmoid(y)


def make_attn_head(in_ch, out_ch):
    bot_ch = cfg.MODEL.SEGATTN_BOT_CH
    od = OrderedDict([("conv0", Conv2d(in_ch, bot_ch, kernel_size=3, padding=1, bias=False)),
                      ("bn0", Norm2d(bot_ch)),
                      ("re0", nn.ReLU(inplace=True))])
    if cfg.MODEL.MSCALE_INNER_3x3:
        od["conv1"] = Conv2d(bot_ch, bot_ch, kernel_size=3, padding=1, bias=False)
        od["bn1"] = Norm2d(bot_ch)
        od["re1"] = nn.ReLU(inplace=True)
    od["conv2"] = Conv2d(bot_ch, out_ch, kernel_size=1, bias=False)
    od["sig"] = nn.Sigmoid()
    return AttnHead(od)


class SegHead(nn.Sequential):
    """3x3 -> BN -> ReLU -> 3x3 -> BN -> ReLU -> 1x1 (children 0..6 as in
    network/utils.py:320-329); returns fp32 logits [B,H,W,classes]."""

    def forward(self, x):
        x = conv_bn(self[0], self[1], x, relu=True)
        x = conv_bn(self[3], self[4], x, relu=True)
        return self[6](x, out_f32=True)


def make_seg_head(in_ch, out_ch):
    """network/utils.py:320-329"""
    bot_ch = cfg.MODEL.SEGATTN_BOT_CH
    return SegHead(Conv2d(in_ch, bot_ch, kernel_size=3, padding=1, bias=False), Norm2d(bot_ch), nn.ReLU(inplace=True),
                   Conv2d(bot_ch, bot_ch, kernel_size=3, padding=1, bias=False), Norm2d(bot_ch), nn.ReLU(inplace=True),
                   Conv2d(bot_ch, out_ch, kernel_size=1, bias=False))
