"""ResNet-50 trunk at output stride 8 for DeepLabV3+ (network/Resnet.py:94-192 and
the dilation surgery of network/utils.py:48-99 of the reference), on the HIP
operator surface.  State-dict keys are the reference's (`layer0.0.weight`,
`layer1.0.conv1.weight`, `layer1.0.downsample.1.running_mean`, ...)."""
from torch import nn

from .. import ops
from ..nn import Conv2d, Norm2d, conv_bn


class Bottleneck(nn.Module):
    """network/Resnet.py:94-133; conv2 carries the stride and, after the stride-8
    surgery, the dilation."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = Norm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                            bias=False)
        self.bn2 = Norm2d(planes)
        self.conv3 = Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = Norm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        res = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        return conv_bn(self.conv3, self.bn3, out, residual=res, relu=True)


class ResNetTrunk(nn.Module):
    """get_resnet('resnet-50', output_stride=8): layer3 / layer4 keep stride 1 with
    dilation 2 / 4 on every conv2 (network/utils.py:71-81).
    forward(x NHWC bf16 [B,H,W,16]) -> (s2 [B,H/4,W/4,256], None, feats [B,H/8,W/8,2048])."""

    def __init__(self, layers=(3, 4, 6, 3), output_stride=8):
        super().__init__()
        assert output_stride == 8, "Only stride8 supported right now"
        self.inplanes = 64
        self.layer0 = nn.Sequential(Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), Norm2d(64),
                                    nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.layer1 = self._make_layer(64, layers[0], stride=1, dilation=1)
        self.layer2 = self._make_layer(128, layers[1], stride=2, dilation=1)
        self.layer3 = self._make_layer(256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(512, layers[3], stride=1, dilation=4)
        for m in self.modules():                       # network/Resnet.py:155-160
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride, dilation):
        # network/Resnet.py:162-177: a downsample branch whenever the block changes the shape;
        # in layer3 / layer4 the surgery resets its stride to 1 but the branch stays
        downsample = nn.Sequential(
            Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
            Norm2d(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, dilation, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, 1, dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = conv_bn(self.layer0[0], self.layer0[1], x, relu=True)
        x = ops.backend().max_pool3x3s2(x)
        x = self.layer1(x)
        s2 = x
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return s2, None, x


def get_resnet(trunk_name, output_stride=8):
    if trunk_name != "resnet-50":
        raise ValueError("Not a valid network arch")
    return ResNetTrunk((3, 4, 6, 3), output_stride)
