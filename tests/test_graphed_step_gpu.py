"""The reference's training loop (train.py:480-533: zero_grad, net(inputs), mean, backward, optim.step) run through
semseg_amd.graph_training -- the product's captured-step helper -- against the same loop run eagerly, at the
benchmarked shape (nullloader: 1 x 3 x 1024 x 1024, datasets/nullloader.py:58-68, with the bench's synthetic batch so
that the RMI covariances are not degenerate), three iterations from identical initial state.  Same kernels, same
order; what may differ is the order of the fp64 / fp32 atomics inside the BatchNorm sums, hence a tolerance."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build():
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.loss.optimizer import FusedSGD
    from semseg_amd.network import ocrnet
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    cfg.LOSS.OCR_AUX_RMI = False
    cfg.MODEL.N_SCALES = None
    cfg.MODEL.BNFUNC = None
    torch.manual_seed(0)
    net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
    net = net.cuda().train()
    # Dropout2d(0.05) of the OCR head draws its mask from torch's generator, whose state a captured graph advances
    # differently from eager launches: off for this comparison (the mask path itself: test_kernels_gpu.py::test_bn_train)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net, FusedSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)


def _reference_loop(net, optim, batches):
    """train.py:480-533, the lines that touch the model."""
    losses = []
    for inputs in batches:
        optim.zero_grad()
        main_loss = net(inputs)
        main_loss = main_loss.mean()
        losses.append(main_loss.detach().clone())
        main_loss.backward()
        optim.step()
    torch.cuda.synchronize()
    return [float(v) for v in losses]


def test_reference_loop_through_the_captured_step_matches_eager():
    import semseg_amd
    import __graft_entry__ as ge
    crop = int(os.environ.get("SSA_PARITY_CROP", "1024"))
    batches = []
    for i in range(3):
        images, gts = ge._synth(1, crop, crop, 40 + i, "cuda")
        batches.append({"images": images, "gts": gts})
    net, optim = _build()
    init = {k: v.clone() for k, v in net.state_dict().items()}
    eager = _reference_loop(net, optim, batches)
    w_eager = net.ocr.cls_head.weight.detach().clone()

    net2, optim2 = _build()
    net2.load_state_dict(init)
    gnet, goptim = semseg_amd.graph_training(net2, optim2)
    graphed = _reference_loop(gnet, goptim, batches)
    w_graphed = net2.ocr.cls_head.weight.detach().clone()
    print("eager", eager, "graphed", graphed)
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-3 * abs(a), (eager, graphed)
    rel = float((w_eager - w_graphed).abs().max() / w_eager.abs().max())
    assert rel < 1e-3, rel
    # evaluation goes straight to the module
    gnet.eval()
    with torch.no_grad():
        out = gnet({"images": batches[0]["images"]})
    assert tuple(out["pred"].shape) == (1, 19, crop, crop)


def _eval_loop(net, images_list, gts):
    """utils/trnval_utils.py:115-160, the lines that touch the model: one `net(inputs)` per image under no_grad, the
    prediction accumulated into a NEW tensor, the assets of the last call read afterwards."""
    outs = []
    with torch.no_grad():
        for images in images_list:
            output = 0.0
            output_dict = net({"images": images, "gts": gts[:, :images.shape[2], :images.shape[3]]})
            output = output + output_dict["pred"]
            outs.append((output, {k: v for k, v in output_dict.items()}))
    torch.cuda.synchronize()
    return outs


def test_validation_loop_through_graph_eval_matches_eager():
    """semseg_amd.graph_eval: the reference's validation loop (utils/trnval_utils.py:134-141 `net(inputs)`) replays one
    captured forward per input signature.  Against the same loop on the bare module: identical outputs (same kernels,
    same order, eval-mode BatchNorm has no atomics); three image sizes through a cache of TWO graphs (least recently
    used evicted, re-captured when it comes back); the outputs handed out are clones (a later replay does not change
    them); parameters changed between two validations reach the replayed forward (packed filters refreshed)."""
    import semseg_amd
    from semseg_amd.config import cfg
    from semseg_amd.graphed import graph_eval
    from test_parity_eval_gpu import _image
    net, _ = _build()
    cfg.MODEL.N_SCALES = [0.5, 1.0]
    try:
        net.eval()
        sizes = [(128, 192), (192, 256), (128, 192), (160, 160), (192, 256)]
        imgs = [_image(h, w, 7 + i).cuda() for i, (h, w) in enumerate(sizes)]
        gts = torch.zeros(1, 256, 256, dtype=torch.long, device="cuda")
        want = _eval_loop(net, imgs, gts)
        g = graph_eval(net, max_graphs=2, capture_after=1)
        assert not g.training and g.wrapped is net
        got = _eval_loop(g, imgs, gts)
        ev = g._eval_stepper
        # (128,192) captured, (192,256) captured, (128,192) replayed, (160,160) evicts (192,256), which is captured again
        assert (ev.captures, ev.evictions, ev.replays) == (4, 2, 5), (ev.captures, ev.evictions, ev.replays)
        for (o0, d0), (o1, d1) in zip(want, got):
            assert sorted(d0) == sorted(d1)
            assert torch.equal(o0, o1)
            for k in d0:
                assert torch.equal(d0[k], d1[k]), k
        # clones: the first call's outputs survived four more replays
        assert torch.equal(got[0][0], want[0][0]) and got[0][0].data_ptr() != got[2][0].data_ptr()
        # training between two validations: a parameter moves -> the replayed forward sees it
        with torch.no_grad():
            net.ocr.cls_head.weight.mul_(1.5)
            net.backbone.conv1.weight.mul_(0.5)
        want2 = _eval_loop(net, imgs[-1:], gts)
        got2 = _eval_loop(g, imgs[-1:], gts)
        assert ev.captures == 4                           # a replay, not a new capture
        assert torch.equal(want2[0][0], got2[0][0])
        assert not torch.equal(want2[0][0], want[-1][0])
        # training-mode calls of the proxy are the module's own (graph_eval captures no training step)
        g.train()
        assert net.training
        loss = g({"images": imgs[0], "gts": gts[:, :128, :192]})
        assert loss.requires_grad and loss.grad_fn is not None
    finally:
        cfg.MODEL.N_SCALES = None
