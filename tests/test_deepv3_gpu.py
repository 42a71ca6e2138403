"""DeepLabV3+/ResNet-50 (BASELINE.json configs[0], rows a14/a15 of SURVEY.md 8a) on
the HIP kernels against the oracle, on the golden inputs generated from the real
reference.  Same tolerance scheme as tests/test_e2e_gpu.py: the measured bf16
storage noise floor (tests/bf16_emu_backend.py) bounds the HIP path op by op."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _shapes():
    out = []
    with open(os.path.join(G, "keys_deepv3.txt")) as f:
        for line in f:
            k, _, s = line.strip().partition(" ")
            out.append((k, tuple(int(v) for v in s.split(",")) if s else ()))
    return out


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _run(backend, sd, images, gts, train, device="cpu"):
    from semseg_amd import ops
    from semseg_amd.loss import CrossEntropyLoss2d
    from semseg_amd.network import get_model
    prev = ops._BACKEND
    ops._set_backend_for_tests(backend)
    try:
        net = get_model("deepv3.DeepV3PlusR50", 19, CrossEntropyLoss2d(ignore_index=255))
        net.load_state_dict(sd)
        net = net.to(device).train(train)
        inputs = {"images": images.to(device), "gts": gts.to(device)}
        if not train:
            with torch.no_grad():
                return net(inputs)["pred"].float().cpu()
        loss = net(inputs)
        loss.backward()
        if device != "cpu":
            torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()}
    finally:
        ops._set_backend_for_tests(prev)


@pytest.fixture(scope="module")
def setup():
    from oracle.model import seeded_state_dict
    gold = torch.load(os.path.join(G, "deepv3_golden.pt"), map_location="cpu", weights_only=False)
    sd = seeded_state_dict(_shapes(), seed=gold["seed"])
    for k in sd:                      # near-identity residual blocks, as in test_e2e_gpu.parity_state_dict
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.2
    # The ASPP image-pooling branch normalises a [B,2048,1,1] tensor: BatchNorm over B = 2 samples,
    # x-hat is +-1 whatever the input, and its backward is chaotic (with gamma ~ 1 two bf16 runs of the
    # same step agree to a gradient cosine of only 0.65-0.80, the emulation to 0.69-0.77 depending on
    # the host's thread count).  Damping that one gamma takes the emulation's median to 0.90.
    sd["aspp.img_conv.1.weight"] = sd["aspp.img_conv.1.weight"] * 0.05
    return gold, sd


def test_deepv3_eval_op_by_op(setup):
    from semseg_amd import ops
    from oracle_backend import OracleBackend
    from bf16_emu_backend import Bf16EmuBackend, traced
    from oracle.deepv3 import DeepV3PlusNet
    gold, sd = setup
    images, gts = gold["images"], gold["gts"]
    sd = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():             # calibrate the running statistics on this batch
        DeepV3PlusNet(sd, 19, training=True, bn_momentum=1.0).forward(images, gts)
    ref_log, emu_err, hip_err, names = [], [], [], []
    ref = _run(traced(OracleBackend(), lambda i, n, y: (ref_log.append(y.detach()), names.append(n))), sd, images,
               gts, False)
    emu = _run(traced(Bf16EmuBackend(), lambda i, n, y: emu_err.append(_rel(y.detach(), ref_log[i]))), sd, images,
               gts, False)
    hip = _run(traced(ops.HipBackend(), lambda i, n, y: hip_err.append(_rel(y.detach().float().cpu(), ref_log[i]))),
               sd, images, gts, False, device="cuda")
    assert len(ref_log) == len(emu_err) == len(hip_err) > 50       # tensors returned by the (list-aware) public ops
    bad = [(i, names[i], tuple(ref_log[i].shape), hip_err[i], emu_err[i]) for i in range(len(hip_err))
           if not hip_err[i] <= 1.5 * emu_err[i] + 5e-3]
    worst = max(range(len(hip_err)), key=lambda i: hip_err[i] - 1.5 * emu_err[i])
    print("ops traced %d; largest excess at op %d (%s): hip %.4f emu %.4f; pred rel err hip %.4f emu %.4f" % (
        len(hip_err), worst, names[worst], hip_err[worst], emu_err[worst], _rel(hip, ref), _rel(emu, ref)))
    assert not bad, bad[:5]
    assert torch.isfinite(hip).all() and _rel(hip, ref) <= 1.5 * _rel(emu, ref) + 5e-3
    ah = (hip.argmax(1) == ref.argmax(1)).float().mean().item()
    ae = (emu.argmax(1) == ref.argmax(1)).float().mean().item()
    print("argmax agreement with the oracle: hip %.4f emu %.4f" % (ah, ae))
    assert ah >= ae - 0.02


def test_deepv3_train_step(setup):
    from semseg_amd import ops
    from oracle_backend import OracleBackend
    from bf16_emu_backend import Bf16EmuBackend
    gold, sd = setup
    images, gts = gold["images"], gold["gts"]
    lr, gr = _run(OracleBackend(), sd, images, gts, True)
    le, ge = _run(Bf16EmuBackend(), sd, images, gts, True)
    lh, gh = _run(ops.HipBackend(), sd, images, gts, True, device="cuda")
    print("deepv3 train loss hip %.6f emu %.6f oracle %.6f" % (lh, le, lr))
    assert abs(lh - lr) <= 2e-3 * abs(lr) + 2 * abs(le - lr)

    def cosines(g):
        return sorted(float((g[n] * r).sum() / (g[n].norm() * r.norm() + 1e-30)) for n, r in gr.items()
                      if float(r.norm()) > 1e-10)
    vh, ve = cosines(gh), cosines(ge)
    print("grad cosine vs oracle: hip min %.4f p10 %.4f median %.4f | emu min %.4f p10 %.4f median %.4f (n=%d)" % (
        vh[0], vh[len(vh) // 10], vh[len(vh) // 2], ve[0], ve[len(ve) // 10], ve[len(ve) // 2], len(vh)))
    assert all(torch.isfinite(g).all() for g in gh.values())
    assert vh[len(vh) // 2] >= ve[len(ve) // 2] - 0.10 and vh[len(vh) // 10] >= ve[len(ve) // 10] - 0.15
    for n, r in gr.items():
        if float(r.norm()) > 1e-10:
            ce = float((ge[n] * r).sum() / (ge[n].norm() * r.norm() + 1e-30))
            ch = float((gh[n] * r).sum() / (gh[n].norm() * r.norm() + 1e-30))
            assert not (ce >= 0.5 and ch < 0.5 * ce), (n, ch, ce)
