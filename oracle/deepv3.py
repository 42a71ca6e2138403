"""ORACLE -- test infrastructure only.  Never imported by the product package.

Functional CPU restatement (plain torch, NCHW fp32) of the reference's
DeepLabV3+/ResNet-50 (`network.deepv3.DeepV3PlusR50`, BASELINE.json configs[0]):
network/deepv3.py:44-93, network/utils.py:48-99 (get_resnet, output stride 8),
network/utils.py:162-218 (AtrousSpatialPyramidPoolingModule),
network/Resnet.py:94-192 (Bottleneck, ResNet).  Driven by a state_dict with the
reference's 363 parameter names.  Pinned against the real reference by
tests/golden/make_golden_deepv3.py + tests/test_oracle_golden.py."""
import torch

from . import ops as O

LAYERS = (("layer1", 3, 64, 1, 1), ("layer2", 4, 128, 2, 1),      # name, blocks, planes, stride, dilation
          ("layer3", 6, 256, 1, 2), ("layer4", 3, 512, 1, 4))     # stride-8 surgery: utils.py:71-81
ASPP_RATES = (12, 24, 36)                                          # 2 x (6, 12, 18) at output stride 8


class DeepV3PlusNet:
    def __init__(self, sd, num_classes=19, training=True, ignore_index=255, bn_momentum=0.1):
        self.sd, self.nc, self.training = sd, num_classes, training
        self.ignore_index, self.bn_momentum = ignore_index, bn_momentum

    def conv(self, x, name, stride=1, padding=0, dilation=1):
        return O.conv2d(x, self.sd[name + ".weight"], None, stride, padding, dilation)

    def bn(self, x, name, relu=False):
        y = O.batch_norm(x, self.sd[name + ".weight"], self.sd[name + ".bias"], self.sd[name + ".running_mean"],
                         self.sd[name + ".running_var"], self.training, self.bn_momentum, 1e-5)
        return torch.relu(y) if relu else y

    # network/Resnet.py:113-133; conv2 carries the stride and (after the surgery) the dilation
    def bottleneck(self, x, pre, stride, dilation, downsample):
        out = self.bn(self.conv(x, pre + ".conv1"), pre + ".bn1", relu=True)
        out = self.bn(self.conv(out, pre + ".conv2", stride, dilation, dilation), pre + ".bn2", relu=True)
        out = self.bn(self.conv(out, pre + ".conv3"), pre + ".bn3")
        res = x
        if downsample:
            res = self.bn(self.conv(x, pre + ".downsample.0", stride), pre + ".downsample.1")
        return torch.relu(out + res)

    # network/utils.py:91-99
    def backbone(self, x):
        b = "backbone."
        x = self.bn(self.conv(x, b + "layer0.0", 2, 3), b + "layer0.1", relu=True)
        x = O.max_pool_3x3_s2(x)
        s2 = None
        for name, blocks, _planes, stride, dil in LAYERS:
            for i in range(blocks):
                # the first block of a layer carries the stride and the downsample branch; the
                # surgery sets every conv2 of layer3/4 to dilation d, stride 1
                x = self.bottleneck(x, "%s%s.%d" % (b, name, i), stride if i == 0 else 1, dil, downsample=(i == 0))
            if name == "layer1":
                s2 = x
        return s2, x

    # network/utils.py:205-218
    def aspp(self, x):
        size = x.shape[-2:]
        img = O.global_avg_pool(x)
        img = self.bn(self.conv(img, "aspp.img_conv.0"), "aspp.img_conv.1", relu=True)
        outs = [O.bilinear(img, size)]
        outs.append(self.bn(self.conv(x, "aspp.features.0.0"), "aspp.features.0.1", relu=True))
        for i, r in enumerate(ASPP_RATES):
            outs.append(self.bn(self.conv(x, "aspp.features.%d.0" % (i + 1), 1, r, r),
                                "aspp.features.%d.1" % (i + 1), relu=True))
        return torch.cat(outs, 1)

    # network/deepv3.py:73-93
    def forward(self, images, gts=None):
        size = images.shape[-2:]
        s2, feats = self.backbone(images)
        conv_aspp = self.conv(self.aspp(feats), "bot_aspp")
        conv_s2 = self.conv(s2, "bot_fine")
        conv_aspp = O.bilinear(conv_aspp, s2.shape[-2:])
        x = torch.cat([conv_s2, conv_aspp], 1)
        x = self.bn(self.conv(x, "final.0", 1, 1), "final.1", relu=True)
        x = self.bn(self.conv(x, "final.3", 1, 1), "final.4", relu=True)
        out = O.bilinear(self.conv(x, "final.6"), size)
        if self.training:
            return O.cross_entropy(out, gts, self.ignore_index)
        return {"pred": out}
