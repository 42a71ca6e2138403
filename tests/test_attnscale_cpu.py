"""network/attnscale.py (north_star's "mscale/attnscale scale-fusion head"): the
product's modules on the oracle's operators against golden vectors from the REAL reference
(tests/golden/make_golden_attnscale.py), in fp64: state_dict keys/shapes, the train loss of
`_forward_fused` / `_forward_paired`, sampled parameter gradients + norms, BN running statistics, the
eval prediction and the attention map the reference returns next to it."""
import os
import sys

import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)

NAMES = ("attnscale.DeepV3R50", "attnscale.DeepV3R50B", "attnscale.DeepV3R50BP")


def attn_shapes():
    out, cur = {}, None
    with open(os.path.join(G, "keys_attnscale.txt")) as f:
        for line in f:
            line = line.strip()
            if line.startswith("# "):
                cur = out.setdefault(line[2:], [])
            elif line:
                k, _, s = line.partition(" ")
                cur.append((k, tuple(int(v) for v in s.split(",")) if s else ()))
    return out


def build(name, gold, train):
    from semseg_amd.config import cfg
    from semseg_amd.loss import CrossEntropyLoss2d
    from semseg_amd.network import get_model
    from oracle.model import seeded_state_dict
    cfg.MODEL.N_SCALES = list(gold["scales"])
    cfg.LOSS.SUPERVISED_MSCALE_WT = gold["wt"]
    net = get_model(name, 19, CrossEntropyLoss2d(ignore_index=255))
    shapes = attn_shapes()[name]
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == shapes, name
    net.load_state_dict(seeded_state_dict(shapes, seed=gold["seed"]))
    return net.double().train(train)


def run(net, inputs):
    """What the reference's fixture ran: `_forward_fused` for ASDV3P, the module call for the paired model."""
    out = net(inputs)
    return out["pred"] if isinstance(out, dict) else out


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
def test_attnscale_wiring(name, oracle_ops):
    from test_siblings_cpu import sample_idx, calibrate
    gold = torch.load(os.path.join(G, "attnscale_golden.pt"), map_location="cpu", weights_only=False)[name]
    inputs = {"images": gold["images"].double(), "gts": gold["gts"].long()}
    net = build(name, gold, True)
    loss = run(net, inputs)
    loss.backward()
    ref = float(gold["train_loss"])
    assert abs(float(loss.detach()) - ref) <= 1e-10 * max(1.0, abs(ref)), (float(loss), ref)
    samples = torch.cat([p.grad.flatten()[sample_idx(p.numel())] for _, p in net.named_parameters()])
    norms = torch.stack([p.grad.flatten().norm() for _, p in net.named_parameters()])
    gn = gold["grad_norms"].double()
    live = gn > 1e-12 * gn.max()
    assert int(live.sum()) > 0.8 * live.numel()
    assert float(((norms - gn)[live].abs() / gn[live]).max()) < 1e-8
    per = torch.repeat_interleave(torch.arange(len(gn)), torch.tensor(
        [min(16, p.numel()) for _, p in net.named_parameters()]))
    err = (samples - gold["grad_samples"].double()).abs() / gn[per].clamp_min(1e-300)
    assert float(err[live[per]].max()) < 1e-8
    rs = torch.cat([v.flatten()[:4] for k, v in net.state_dict().items()
                    if k.endswith("running_mean") or k.endswith("running_var")])
    assert torch.allclose(rs, gold["running_sample"].double(), rtol=1e-9, atol=1e-12)

    net = build(name, gold, True)
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        run(net, inputs)
    for m in bns:
        m.momentum = 0.1
    net.eval()
    with torch.no_grad():
        pred, attn = run(net, inputs)
    assert tuple(pred.shape) == (2, 19, 64, 96) and tuple(attn.shape) == tuple(gold["eval_attn_shape"])
    assert torch.allclose(pred[:, :, ::8, ::8], gold["eval_pred"].double(), rtol=1e-7, atol=1e-7)
    assert torch.allclose(attn[:, :, ::4, ::4], gold["eval_attn"].double(), rtol=1e-7, atol=1e-7)
