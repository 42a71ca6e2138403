"""Sibling architectures (network/mscale.py, mscale2.py,
ocrnet.OCRNetASPP) -- the product's modules on the oracle's operators
(tests/oracle_backend.py) against golden vectors from the REAL reference
(tests/golden/make_golden_siblings.py): state_dict keys/shapes, train loss,
sampled parameter gradients + norms, BN running statistics, two-scale and
N-scale eval outputs, all in fp64."""
import os
import sys

import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)

NAMES = ("mscale.HRNet", "mscale.HRNet_ASP", "mscale.DeepV3R50", "mscale.MscaleV3Plus.fuse2b",
         "mscale2.DeepV3R50", "ocrnet.OCRNetASPP", "ocrnet.HRNet")


def sibling_shapes():
    out, cur = {}, None
    with open(os.path.join(G, "keys_siblings.txt")) as f:
        for line in f:
            line = line.strip()
            if line.startswith("# "):
                cur = out.setdefault(line[2:], [])
            elif line:
                k, _, s = line.partition(" ")
                cur.append((k, tuple(int(v) for v in s.split(",")) if s else ()))
    return out


def sample_idx(n, k=16, seed=0):         # tests/golden/make_golden.py:sample_idx
    g = torch.Generator().manual_seed(seed + n)
    return torch.randint(0, n, (min(k, n),), generator=g)


def build(name, gold, train):
    """The product module for golden configuration `name`, seeded as the fixture was."""
    from semseg_amd.config import cfg
    from semseg_amd.loss import CrossEntropyLoss2d, RMILoss
    from semseg_amd.network import get_model, mscale, ocrnet
    from oracle.model import seeded_state_dict
    cfg.LOSS.SUPERVISED_MSCALE_WT = gold["wt"]
    cfg.MODEL.N_SCALES = None
    crit = RMILoss(num_classes=19, ignore_index=255) if gold["crit"] == "rmi" else CrossEntropyLoss2d(ignore_index=255)
    if name == "mscale.MscaleV3Plus.fuse2b":
        net = mscale.MscaleV3Plus(19, trunk="resnet-50", criterion=crit, fuse_aspp=True, attn_2b=True)
    elif name == "ocrnet.OCRNetASPP":
        net = ocrnet.OCRNetASPP(19, criterion=crit)
    else:
        net = get_model(name, 19, crit)
    shapes = sibling_shapes()[name]
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == shapes, name
    net.load_state_dict(seeded_state_dict(shapes, seed=gold["seed"]))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net.double().train(train)


def calibrate(net, inputs):
    """BN running statistics := batch statistics of this batch (momentum 1.0), as the fixture did."""
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    net.train()
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        net(inputs)
    for m in bns:
        m.momentum = 0.1
    return net.eval()


@pytest.fixture(scope="module")
def gold_all():
    return torch.load(os.path.join(G, "siblings_golden.pt"), map_location="cpu", weights_only=False)


@pytest.fixture()
def oracle_ops():
    from semseg_amd import ops
    from semseg_amd.config import cfg
    from oracle_backend import OracleBackend
    prev = ops._BACKEND
    ops._set_backend_for_tests(OracleBackend())
    yield
    ops._set_backend_for_tests(prev)
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0
    cfg.MODEL.N_SCALES = None


@pytest.mark.parametrize("name", NAMES)
def test_sibling_wiring(name, gold_all, oracle_ops):
    from semseg_amd.config import cfg
    gold = gold_all[name]
    # fp64 on both sides; loss/rmi.py computes parts of the RMI criterion in fp32 whatever the
    # input type, so that configuration is pinned to 1e-5 instead of 1e-8
    f64 = gold["crit"] != "rmi"
    rtol = 1e-8 if f64 else 1e-4
    inputs = {"images": gold["images"].double(), "gts": gold["gts"].long()}

    net = build(name, gold, True)
    loss = net(inputs)
    loss.backward()
    ref = float(gold["train_loss"])
    assert abs(float(loss.detach()) - ref) <= (1e-10 if f64 else 1e-5) * max(1.0, abs(ref)), (float(loss), ref)
    samples = torch.cat([p.grad.flatten()[sample_idx(p.numel())] for _, p in net.named_parameters()])
    norms = torch.stack([p.grad.flatten().norm() for _, p in net.named_parameters()])
    gn = gold["grad_norms"].double()
    live = gn > (1e-12 if f64 else 1e-6) * gn.max()        # conv biases in front of BN: analytically zero
    assert int(live.sum()) > 0.8 * live.numel()
    print("grad norm rel err max %.3g" % float(((norms - gn)[live].abs() / gn[live]).max()))
    assert float(((norms - gn)[live].abs() / gn[live]).max()) < rtol
    # sampled entries: error relative to the parameter's gradient norm
    per = torch.repeat_interleave(torch.arange(len(gn)), torch.tensor(
        [min(16, p.numel()) for _, p in net.named_parameters()]))
    err = (samples - gold["grad_samples"].double()).abs() / gn[per].clamp_min(1e-300)
    print("grad sample err max %.3g" % float(err[live[per]].max()))
    assert float(err[live[per]].max()) < rtol
    rs = torch.cat([v.flatten()[:4] for k, v in net.state_dict().items()
                    if k.endswith("running_mean") or k.endswith("running_var")])
    assert torch.allclose(rs, gold["running_sample"].double(), rtol=1e-9, atol=1e-12)

    net = calibrate(build(name, gold, True), inputs)
    etol = dict(rtol=1e-7, atol=1e-7)
    with torch.no_grad():
        o = net(inputs)
        assert tuple(o["pred"].shape) == (2, 19, 64, 96)
        assert sorted(o) == sorted(gold["eval"])
        for k, v in gold["eval"].items():
            assert torch.allclose(o[k][:, :, ::8, ::8], v.double(), **etol), (k, float((o[k][:, :, ::8, ::8] - v).abs().max()))
        if "eval_nscale" in gold:
            cfg.MODEL.N_SCALES = [0.5, 1.0, 2.0]
            o = net(inputs)
            cfg.MODEL.N_SCALES = None
            assert sorted(o) == sorted(gold["eval_nscale"])
            for k, v in gold["eval_nscale"].items():
                assert torch.allclose(o[k][:, :, ::8, ::8], v.double(), **etol), ("nscale", k)
