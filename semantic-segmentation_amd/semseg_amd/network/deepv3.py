"""DeepLabV3+ (network/deepv3.py:44-93 of the reference) with the ResNet-50 trunk:
the `--arch deepv3.DeepV3PlusR50` of BASELINE.json configs[0].  Same factory
name, call contract and state_dict (363 keys); NHWC bf16 on the HIP kernels."""
import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, conv_bn, initialize_weights
from .mynn import Upsample
from .utils import get_trunk


class AtrousSpatialPyramidPoolingModule(nn.Module):
    """network/utils.py:162-218: image pooling + 1x1 + three dilated 3x3 branches,
    concatenated (image-pooling branch first)."""

    def __init__(self, in_dim, reduction_dim=256, output_stride=16, rates=(6, 12, 18)):
        super().__init__()
        if output_stride == 8:
            rates = [2 * r for r in rates]
        elif output_stride != 16:
            raise ValueError("output stride of {} not supported".format(output_stride))
        feats = [nn.Sequential(Conv2d(in_dim, reduction_dim, kernel_size=1, bias=False), Norm2d(reduction_dim),
                               nn.ReLU(inplace=True))]
        for r in rates:
            feats.append(nn.Sequential(Conv2d(in_dim, reduction_dim, kernel_size=3, dilation=r, padding=r, bias=False),
                                       Norm2d(reduction_dim), nn.ReLU(inplace=True)))
        self.features = nn.ModuleList(feats)
        self.img_pooling = nn.AdaptiveAvgPool2d(1)
        self.img_conv = nn.Sequential(Conv2d(in_dim, reduction_dim, kernel_size=1, bias=False), Norm2d(reduction_dim),
                                      nn.ReLU(inplace=True))

    def forward(self, x):
        B = ops.backend()
        size = x.shape[1:3]
        img = B.global_avg_pool(x)
        img = conv_bn(self.img_conv[0], self.img_conv[1], img, relu=True)
        outs = [B.bilinear(img, size)]
        for f in self.features:
            outs.append(conv_bn(f[0], f[1], x, relu=True))
        return B.cat(outs)


def get_aspp(high_level_ch, bottleneck_ch, output_stride, dpc=False):
    """network/utils.py:301-311"""
    assert not dpc, "DPC is not on the supported path"
    return AtrousSpatialPyramidPoolingModule(high_level_ch, bottleneck_ch, output_stride=output_stride), 5 * bottleneck_ch


class DeepV3Plus(nn.Module):
    """DeepLabV3+ with various trunks supported; always stride 8 (network/deepv3.py:44-93)."""

    def __init__(self, num_classes, trunk="resnet-50", criterion=None, use_dpc=False, init_all=False):
        super().__init__()
        self.criterion = criterion
        self.backbone, s2_ch, _s4_ch, high_level_ch = get_trunk(trunk)
        self.aspp, aspp_out_ch = get_aspp(high_level_ch, bottleneck_ch=256, output_stride=8, dpc=use_dpc)
        self.bot_fine = Conv2d(s2_ch, 48, kernel_size=1, bias=False)
        self.bot_aspp = Conv2d(aspp_out_ch, 256, kernel_size=1, bias=False)
        self.final = nn.Sequential(
            Conv2d(256 + 48, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, 256, kernel_size=3, padding=1, bias=False), Norm2d(256), nn.ReLU(inplace=True),
            Conv2d(256, num_classes, kernel_size=1, bias=False))
        if init_all:
            initialize_weights(self.aspp, self.bot_aspp, self.bot_fine, self.final)
        else:
            initialize_weights(self.final)

    def forward(self, inputs):
        assert "images" in inputs
        B = ops.backend()
        B.begin_step(inputs["images"].device)
        images = inputs["images"]
        size = (images.shape[2], images.shape[3])
        x = B.image_to_nhwc(images, size)
        s2_features, _, final_features = self.backbone(x)
        aspp = self.aspp(final_features)
        conv_aspp = self.bot_aspp(aspp)
        conv_s2 = self.bot_fine(s2_features)
        conv_aspp = B.bilinear(conv_aspp, s2_features.shape[1:3])
        cat_s4 = B.cat([conv_s2, conv_aspp])
        x = conv_bn(self.final[0], self.final[1], cat_s4, relu=True)
        x = conv_bn(self.final[3], self.final[4], x, relu=True)
        out = Upsample(self.final[6](x, out_f32=True), size)       # [B,H,W,classes] fp32
        B.end_forward()
        out = out.permute(0, 3, 1, 2)
        if self.training:
            assert "gts" in inputs
            return self.criterion(out, inputs["gts"])
        return {"pred": out}


def DeepV3PlusR50(num_classes, criterion):
    return DeepV3Plus(num_classes, trunk="resnet-50", criterion=criterion)
