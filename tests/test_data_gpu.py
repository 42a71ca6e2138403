"""Label resize on the GPU (ssa_resize_nearest_u8 through the C ABI): bit-exact
against the golden Pillow checksums, against the oracle on random sizes, and at
Cityscapes' full 1024x2048 size."""
import numpy as np
import pytest
import torch

from util import ACT_DTYPE

from test_data_cpu import golden_masks, checksums

pytestmark = pytest.mark.gpu


def test_nearest_resize_matches_pillow_golden():
    from semseg_amd.datasets import resize_labels_nearest
    for case, m in golden_masks():
        r = resize_labels_nearest(torch.from_numpy(m).cuda(), tuple(case["dst"])).cpu().numpy()
        assert checksums(r) == (case["sum"], case["weighted"]), case["src"]
        assert r[0].tolist()[:64] == case["first_row"] and r[:, -1].tolist()[:64] == case["last_col"]


def test_nearest_resize_batched_and_full_size():
    from oracle.data import pil_nearest_indices
    from semseg_amd.datasets import resize_labels_nearest
    rng = np.random.default_rng(11)
    for (b, hs, ws), (hd, wd) in [((3, 37, 53), (80, 41)), ((2, 1024, 2048), (717, 1434)), ((1, 64, 64), (64, 64)),
                                  ((2, 300, 200), (1, 1))]:
        m = rng.integers(0, 256, (b, hs, ws), dtype=np.uint8)
        got = resize_labels_nearest(torch.from_numpy(m).cuda(), (hd, wd)).cpu().numpy()
        iy, ix = pil_nearest_indices(hd, hs), pil_nearest_indices(wd, ws)
        want = m[:, iy][:, :, ix]
        assert got.shape == want.shape and np.array_equal(got, want)
    try:
        from PIL import Image
    except ImportError:
        return
    m = rng.integers(0, 20, (1024, 2048), dtype=np.uint8)
    want = np.array(Image.fromarray(m).resize((1434, 717), Image.NEAREST))
    got = resize_labels_nearest(torch.from_numpy(m).cuda(), (717, 1434)).cpu().numpy()
    assert np.array_equal(got, want)


def test_confusion_matrix_on_device():
    """ssa_confusion_matrix (argmax + fast_hist on the GPU) bit-exact against the oracle's
    softmax -> max(1) -> np.bincount chain, with exact ties, ignore labels and negative labels."""
    from oracle.data import fast_hist, eval_predictions
    from semseg_amd.utils import confusion_matrix
    from semseg_amd.utils import fast_hist as fast_hist_dev
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 19, 67, 93
    logits = torch.randn(B, C, H, W, generator=g) * 3
    logits[:, 5] = logits[:, 3]                       # exact ties: the first maximum must win
    logits[0, :, :8] = 0.0                            # all-equal rows -> class 0
    gts = torch.randint(0, C, (B, H, W), generator=g)
    gts[torch.rand(B, H, W, generator=g) < 0.1] = 255
    gts[0, 0, :5] = -1
    pred_ref = eval_predictions(logits)
    assert torch.equal(pred_ref, logits.max(1)[1])    # on this data softmax does not merge distinct logits
    want = fast_hist(pred_ref.numpy().flatten(), gts.numpy().flatten(), C)
    nhwc_view = logits.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2)   # what the network hands back
    hist, pred = confusion_matrix(nhwc_view, gts.cuda(), C, return_predictions=True)
    assert torch.equal(pred.cpu().long(), pred_ref)
    assert np.array_equal(hist.cpu().numpy(), want)
    hist2 = confusion_matrix(logits.cuda(), gts.cuda(), C, hist=hist.clone())         # accumulates; NCHW input
    assert np.array_equal(hist2.cpu().numpy(), 2 * want)
    assert np.array_equal(fast_hist_dev(pred, gts.cuda(), C).cpu().numpy(), want)


def test_pipeline_tail_on_device_is_bit_exact():
    """ssa_image_u8_crop_flip_normalize / ssa_label_u8_crop_flip against the oracle (pinned to PIL +
    torch in tests/test_data_cpu.py) on the golden image and on a full-size 1024x2048 frame: the
    bf16 image equals bf16(oracle fp32) bit for bit, the labels are identical."""
    import os
    import numpy as np
    import torch
    from oracle.data import crop_flip_normalize as oracle
    from semseg_amd.datasets.transforms import crop_flip_normalize
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.pt"),
                   weights_only=False)
    cases = [(g["img"].numpy(), g["lab"].numpy(), c["window"], c["flip"]) for c in g["cases"]]
    rng = np.random.RandomState(1)
    big = rng.randint(0, 256, (1024, 2048, 3)).astype(np.uint8)
    biglab = rng.randint(0, 256, (1024, 2048)).astype(np.uint8)
    cases += [(big, biglab, (512, 0, 1024, 1024), True), (big, biglab, (0, 0, 2048, 1024), False)]
    for img, lab, window, flip in cases:
        out, gts = crop_flip_normalize(torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda(), window, flip,
                                       (g["mean"], g["std"]))
        want_im, want_lab = oracle(img, lab, window, flip, g["mean"], g["std"])
        want = torch.from_numpy(want_im).permute(1, 2, 0).to(ACT_DTYPE)
        got = out[0].cpu()
        assert torch.equal(got[..., :3].view(torch.int16), want.contiguous().view(torch.int16)), window
        assert int(got[..., 3:].abs().max()) == 0
        assert torch.equal(gts[0].cpu(), torch.from_numpy(want_lab)), window


def test_bicubic_scale_step_on_device_is_bit_exact():
    """ssa_resample_u8 (two passes) against the oracle (pinned to Pillow in tests/test_data_cpu.py): the
    image half of RandomSizeAndCrop's scale step, incl. a full-size 1024x2048 frame scaled by 0.9 and 1.7."""
    import numpy as np
    from oracle.data import resize_bicubic_u8
    from semseg_amd.datasets.transforms import resize_image_bicubic
    rng = np.random.default_rng(1)
    cases = [(37, 53, 74, 106), (64, 96, 48, 72), (50, 70, 100, 35), (33, 45, 67, 91), (97, 131, 49, 66),
             (1024, 2048, 921, 1843), (512, 1024, 870, 1740)]
    for h, w, hd, wd in cases:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = resize_image_bicubic(torch.from_numpy(img).cuda(), (hd, wd)).cpu().numpy()
        assert got.shape == (hd, wd, 3)
        assert np.array_equal(got, resize_bicubic_u8(img, (hd, wd))), (h, w, hd, wd)


def test_device_prefetcher_overlaps_and_preserves_batches():
    from semseg_amd.datasets.transforms import DevicePrefetcher
    batches = [(torch.randn(1, 3, 256, 256), torch.randint(0, 19, (1, 256, 256)), "n%d" % i, 0.0) for i in range(4)]
    for i, (img, gts, name, _) in enumerate(DevicePrefetcher(batches)):
        assert img.is_cuda and gts.is_cuda and name == "n%d" % i
        assert torch.equal(img.cpu(), batches[i][0]) and torch.equal(gts.cpu(), batches[i][1])
