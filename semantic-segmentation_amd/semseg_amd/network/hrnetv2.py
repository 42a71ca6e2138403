"""HRNetV2-W48 trunk on the HIP operator surface.

Same module tree (hence the same 1,5xx state_dict keys) as the reference's
network/hrnetv2.py; the forward passes are written against fused operators:
conv -> BN(+residual)(+ReLU) is two kernels, the cross-resolution fuse sum is
one kernel, all tensors are NHWC bf16.
"""
import os

import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, conv_bn, residual_link

BN_MOMENTUM = 0.1   # network/hrnetv2.py:26


def conv3x3(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """network/hrnetv2.py:37-66"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = Norm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = Norm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def forward(self, x):
        res = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        # dataflow hints for the backend's backward fusions: bn1's output feeds conv2 alone, and
        # (without a downsample branch) the block input is consumed by conv1 and the final add only
        link = residual_link() if self.downsample is None else None
        out = conv_bn(self.conv1, self.bn1, x, relu=True, block=link)
        return conv_bn(self.conv2, self.bn2, out, residual=res, relu=True, private_input=True, block=link)


class Bottleneck(nn.Module):
    """network/hrnetv2.py:69-106"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = Norm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = Norm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = Norm2d(planes * 4, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        res = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True, private_input=True)
        return conv_bn(self.conv3, self.bn3, out, residual=res, relu=True, private_input=True)


BLOCKS = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _down_chain(cin, cout_last, steps):
    """(i-j) stride-2 3x3 conv+BN stages; ReLU after all but the last
    (network/hrnetv2.py:203-222)."""
    layers = []
    for k in range(steps):
        last = k == steps - 1
        cout = cout_last if last else cin
        seq = [Conv2d(cin, cout, 3, 2, 1, bias=False), Norm2d(cout, momentum=BN_MOMENTUM)]
        if not last:
            seq.append(nn.ReLU(inplace=True))
        layers.append(nn.Sequential(*seq))
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """network/hrnetv2.py:109-254"""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels,
                 multi_scale_output=True):
        super().__init__()
        assert num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)
        self.num_branches = num_branches
        self.num_inchannels = list(num_inchannels)
        branches = []
        for i in range(num_branches):
            cin, cout = self.num_inchannels[i], num_channels[i] * block.expansion
            down = None
            if cin != cout:
                down = nn.Sequential(Conv2d(cin, cout, kernel_size=1, bias=False),
                                     Norm2d(cout, momentum=BN_MOMENTUM))
            layers = [block(cin, num_channels[i], 1, down)]
            layers += [block(cout, num_channels[i]) for _ in range(1, num_blocks[i])]
            self.num_inchannels[i] = cout
            branches.append(nn.Sequential(*layers))
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = None
        if num_branches > 1:
            ch = self.num_inchannels
            rows = []
            for i in range(num_branches if multi_scale_output else 1):
                row = []
                for j in range(num_branches):
                    if j > i:
                        row.append(nn.Sequential(Conv2d(ch[j], ch[i], 1, 1, 0, bias=False),
                                                 Norm2d(ch[i], momentum=BN_MOMENTUM)))
                    elif j == i:
                        row.append(None)
                    else:
                        row.append(_down_chain(ch[j], ch[i], i - j))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(inplace=True)

    def get_num_inchannels(self):
        return self.num_inchannels

    def forward(self, xs):
        B = ops.backend()
        # the resolution branches are independent until the fuse layers
        xs = B.parallel([(lambda i=i: self.branches[i](xs[i])) for i in range(self.num_branches)], level=2)
        if self.num_branches == 1:
            return xs
        outs = []
        for i, row in enumerate(self.fuse_layers):
            terms = []
            for j in range(self.num_branches):
                if j == i:
                    terms.append(xs[j])
                elif j > i:
                    t = conv_bn(row[j][0], row[j][1], xs[j])
                    terms.append(B.bilinear(t, xs[i].shape[1:3]))
                else:
                    t = xs[j]
                    for stage in row[j]:
                        t = conv_bn(stage[0], stage[1], t, relu=len(stage) == 3)
                    terms.append(t)
            outs.append(B.sum_act(terms, relu=True))
        return outs


class HighResolutionNet(nn.Module):
    """network/hrnetv2.py:263-449.  forward(x NHWC bf16 [B,H,W,16]) ->
    (None, None, feats [B,H/4,W/4,720])."""

    def __init__(self, **kwargs):
        super().__init__()
        extra = cfg.MODEL.OCR_EXTRA
        self.conv1 = Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = Norm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = Norm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)

        s1 = extra["STAGE1"]
        block = BLOCKS[s1["BLOCK"]]
        planes, nblocks = s1["NUM_CHANNELS"][0], s1["NUM_BLOCKS"][0]
        down = nn.Sequential(Conv2d(64, planes * block.expansion, kernel_size=1, bias=False),
                             Norm2d(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(64, planes, 1, down)]
        layers += [block(planes * block.expansion, planes) for _ in range(1, nblocks)]
        self.layer1 = nn.Sequential(*layers)
        pre = [planes * block.expansion]

        for idx in (2, 3, 4):
            sc = extra["STAGE%d" % idx]
            blk = BLOCKS[sc["BLOCK"]]
            chans = [c * blk.expansion for c in sc["NUM_CHANNELS"]]
            setattr(self, "stage%d_cfg" % idx, sc)
            setattr(self, "transition%d" % (idx - 1), self._make_transition(pre, chans))
            stage, pre = self._make_stage(sc, chans)
            setattr(self, "stage%d" % idx, stage)
        self.high_level_ch = int(sum(pre))

    @staticmethod
    def _make_transition(pre, cur):
        """network/hrnetv2.py:319-350"""
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    layers.append(nn.Sequential(Conv2d(pre[i], c, 3, 1, 1, bias=False),
                                                Norm2d(c, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = c if j == i - len(pre) else cin
                    chain.append(nn.Sequential(Conv2d(cin, cout, 3, 2, 1, bias=False),
                                               Norm2d(cout, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    @staticmethod
    def _make_stage(sc, num_inchannels):
        mods = []
        for _ in range(sc["NUM_MODULES"]):
            m = HighResolutionModule(sc["NUM_BRANCHES"], BLOCKS[sc["BLOCK"]], sc["NUM_BLOCKS"],
                                     num_inchannels, sc["NUM_CHANNELS"], True)
            num_inchannels = m.get_num_inchannels()
            mods.append(m)
        return nn.Sequential(*mods), num_inchannels

    @staticmethod
    def _apply_transition(trans, ys, n_prev):
        xs = []
        for i, t in enumerate(trans):
            src = ys[i] if i < n_prev else ys[-1]
            if t is None:
                xs.append(src)
            elif i < n_prev:
                xs.append(conv_bn(t[0], t[1], src, relu=True))
            else:
                for stage in t:
                    src = conv_bn(stage[0], stage[1], src, relu=True)
                xs.append(src)
        return xs

    def forward(self, x):
        B = ops.backend()
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        x = conv_bn(self.conv2, self.bn2, x, relu=True)
        x = self.layer1(x)
        ys = [x]
        ys = self.stage2(self._apply_transition(self.transition1, ys, 1))
        ys = self.stage3(self._apply_transition(self.transition2, ys, self.stage2_cfg["NUM_BRANCHES"]))
        ys = self.stage4(self._apply_transition(self.transition3, ys, self.stage3_cfg["NUM_BRANCHES"]))
        size = ys[0].shape[1:3]
        feats = B.cat([ys[0]] + [B.bilinear(y, size) for y in ys[1:]])
        return None, None, feats

    def init_weights(self, pretrained=None):
        """network/hrnetv2.py:451-477: N(0, 1e-3) convs / unit BN outside the
        heads, then an optional ImageNet checkpoint."""
        if pretrained is None:
            pretrained = cfg.MODEL.HRNET_CHECKPOINT
        for name, m in self.named_modules():
            if any(part in name for part in ("cls", "aux", "ocr")):
                continue
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if pretrained and os.path.isfile(pretrained):
            pre = torch.load(pretrained, map_location="cpu")
            own = self.state_dict()
            pre = {k.replace("last_layer", "aux_head").replace("model.", ""): v for k, v in pre.items()}
            own.update({k: v for k, v in pre.items() if k in own})
            self.load_state_dict(own)
        elif pretrained:
            raise RuntimeError("No such file {}".format(pretrained))


def get_seg_model():
    model = HighResolutionNet()
    model.init_weights()
    return model
