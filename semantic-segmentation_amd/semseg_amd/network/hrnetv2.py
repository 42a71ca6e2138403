"""HRNetV2-W48 trunk on the HIP operator surface.

Same module tree (hence the same 1,5xx state_dict keys) as the reference's
network/hrnetv2.py; the forward passes are written against fused operators:
conv -> BN(+residual)(+ReLU) is two kernels, the cross-resolution fuse sum is
one kernel, all tensors are NHWC bf16.

Lockstep over independent problems.  The trunk takes ONE image or a LIST of images (the
scale passes of MscaleOCR, network/ocrnet.py:264-327); inside a HighResolutionModule the
2-4 resolution branches are independent until the fuse layers.  Every depth level of
(branch x pass) problems is handed to the operator surface as one list, which the HIP
backend turns into one launch per kernel instantiation (csrc/group.h) -- where the
reference issues one cuDNN call per (layer, branch, pass).
"""
import os

import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..nn import Conv2d, Norm2d, conv_bn

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
        """x: a tensor or a list of tensors (the scale passes)."""
        return run_basic_blocks([self] * len(x), x) if isinstance(x, (list, tuple)) else run_basic_blocks([self], [x])[0]


def run_basic_blocks(blocks, xs):
    """blocks[i] applied to xs[i], all problems in lockstep."""
    if all(b.downsample is None for b in blocks):
        return ops.backend().basic_block(blocks, list(xs))
    outs = []
    for b, x in zip(blocks, xs):
        res = x if b.downsample is None else conv_bn(b.downsample[0], b.downsample[1], x)
        out = conv_bn(b.conv1, b.bn1, x, relu=True)
        outs.append(conv_bn(b.conv2, b.bn2, out, residual=res, relu=True))
    return outs


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
        """x: a tensor or a list of tensors (the scale passes)."""
        res = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        return conv_bn(self.conv3, self.bn3, out, residual=res, relu=True)


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
        """xs[i]: branch i's input -- a tensor, or a list of tensors (the scale passes)."""
        B = ops.backend()
        nb = self.num_branches
        multi = isinstance(xs[0], (list, tuple))
        xs = [list(x) if multi else [x] for x in xs]
        P = len(xs[0])
        # ---- branches: every depth level of all (branch, pass) problems in lockstep
        jobs = [(i, p) for i in range(nb) for p in range(P)]
        cur = [xs[i][p] for i, p in jobs]
        for d in range(max(len(br) for br in self.branches)):
            sel = [k for k, (i, p) in enumerate(jobs) if d < len(self.branches[i])]
            blocks = [self.branches[jobs[k][0]][d] for k in sel]
            if all(isinstance(b, BasicBlock) for b in blocks):
                outs = run_basic_blocks(blocks, [cur[k] for k in sel])
            else:
                outs = [b(cur[k]) for b, k in zip(blocks, sel)]
            for k, o in zip(sel, outs):
                cur[k] = o
        ys = [[cur[i * P + p] for p in range(P)] for i in range(nb)]
        if nb == 1:
            return ys if multi else [y[0] for y in ys]
        # ---- fuse layers (network/hrnetv2.py:236-252), level by level over all (i, j, pass)
        rows = self.fuse_layers
        # every branch output feeds every row: one handle per consumer (their gradients are summed by ONE
        # grouped launch in the backward pass, ops.fan_out)
        handles = B.fan_out([ys[i][p] for i, p in jobs], [len(rows)] * len(jobs))
        use = {(i, p): handles[k] for k, (i, p) in enumerate(jobs)}       # use[(j, p)][i]: row i's handle
        term = {}
        up = [(i, j, p) for i in range(len(rows)) for j in range(nb) if j > i for p in range(P)]
        join_up = None
        if up:
            # nothing below needs the upsampled terms before the final sum: a parallel branch (ops.fork)
            def up_branch():
                ts = conv_bn([rows[i][j][0] for i, j, p in up], [rows[i][j][1] for i, j, p in up],
                             [use[(j, p)][i] for i, j, p in up])
                return B.bilinear(ts, [tuple(ys[i][p].shape[1:3]) for i, j, p in up])
            join_up = B.fork(up_branch)
        down = [(i, j, p) for i in range(len(rows)) for j in range(nb) if j < i for p in range(P)]
        state = {k: use[(k[1], k[2])][k[0]] for k in down}
        step = 0
        while True:
            sel = [k for k in down if step < len(rows[k[0]][k[1]])]
            if not sel:
                break
            stages = [rows[i][j][step] for i, j, p in sel]
            outs = conv_bn([st[0] for st in stages], [st[1] for st in stages], [state[k] for k in sel],
                           relu=[len(st) == 3 for st in stages])
            state.update(zip(sel, outs))
            step += 1
        term.update(state)
        if join_up is not None:
            term.update(zip(up, join_up()))
        sums = []
        for i in range(len(rows)):
            for p in range(P):
                sums.append([use[(i, p)][i] if j == i else term[(i, j, p)] for j in range(nb)])
        outs = B.sum_act(sums, relu=True)
        outs = [[outs[i * P + p] for p in range(P)] for i in range(len(rows))]
        return outs if multi else [o[0] for o in outs]


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
        """ys[i]: list over the scale passes.  The convs of one chain depth run as one list."""
        xs = [None] * len(trans)
        chains = []                     # (new branch index, [stages], source)
        for i, t in enumerate(trans):
            src = ys[i] if i < n_prev else ys[-1]
            if t is None:
                xs[i] = src
            elif i < n_prev:
                chains.append((i, [t], src))
            else:
                chains.append((i, list(t), src))
        step = 0
        while True:
            sel = [c for c in chains if step < len(c[1])]
            if not sel:
                break
            convs, bns, ins, owner = [], [], [], []
            for ci, c in enumerate(chains):
                if step < len(c[1]):
                    for x in c[2]:
                        convs.append(c[1][step][0])
                        bns.append(c[1][step][1])
                        ins.append(x)
                        owner.append(ci)
            outs = conv_bn(convs, bns, ins, relu=True)
            new_src = {}
            for ci, o in zip(owner, outs):
                new_src.setdefault(ci, []).append(o)
            chains = [(c[0], c[1], new_src.get(ci, c[2])) for ci, c in enumerate(chains)]
            step += 1
        for i, stages, src in chains:
            xs[i] = src
        return xs

    @staticmethod
    def _run_stage(stage, xs):
        for m in stage:
            xs = m(xs)
        return xs

    def forward(self, x):
        """x: NHWC bf16 image [B,H,W,16], or a list of images (the scale passes, run in lockstep).
        Returns (None, None, feats) -- feats a tensor or a list like x."""
        B = ops.backend()
        multi = isinstance(x, (list, tuple))
        xs = list(x) if multi else [x]
        xs = conv_bn(self.conv1, self.bn1, xs, relu=True)
        xs = conv_bn(self.conv2, self.bn2, xs, relu=True)
        for blk in self.layer1:
            xs = blk(xs)
        ys = [xs]
        ys = self._run_stage(self.stage2, self._apply_transition(self.transition1, ys, 1))
        ys = self._run_stage(self.stage3, self._apply_transition(self.transition2, ys, self.stage2_cfg["NUM_BRANCHES"]))
        ys = self._run_stage(self.stage4, self._apply_transition(self.transition3, ys, self.stage3_cfg["NUM_BRANCHES"]))
        feats = B.upsample_cat([[ys[i][p] for i in range(len(ys))] for p in range(len(xs))])
        return None, None, (feats if multi else feats[0])

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
