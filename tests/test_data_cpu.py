"""Integer-exact boundary pieces on CPU: the oracle
against golden vectors from the real reference sampler / Pillow, and the
product's host logic against the oracle."""
import json
import os
import random

import torch

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_golden.json")


def _gold():
    with open(G) as f:
        return json.load(f)


def golden_masks():
    """The source masks of make_golden_data.py, regenerated from its seed."""
    rng = np.random.default_rng(7)
    for case in _gold()["nearest"]:
        hs, ws = case["src"]
        yield case, rng.integers(0, 256, (hs, ws), dtype=np.uint8)


def checksums(r):
    r = np.asarray(r)
    w = (np.arange(r.size).reshape(r.shape) % 251 + 1)
    return int(r.astype(np.int64).sum()), int((r.astype(np.int64) * w).sum())


def test_oracle_sampler_matches_reference():
    from oracle.data import sampler_indices
    for c in _gold()["sampler"]:
        got = sampler_indices(c["n"], c["epoch"], c["rank"], c["world"], c["pad"], c["consecutive"], c["permutation"])
        assert got == c["indices"], c


def test_product_sampler_matches_reference_and_oracle():
    from oracle.data import sampler_indices
    from semseg_amd.datasets import DistributedSampler, shard_indices
    for c in _gold()["sampler"]:
        s = DistributedSampler(list(range(c["n"])), pad=c["pad"], consecutive_sample=c["consecutive"],
                               permutation=c["permutation"], num_replicas=c["world"], rank=c["rank"])
        s.set_epoch(c["epoch"])
        assert list(s) == c["indices"] and len(s) == len(c["indices"])
    rnd = random.Random(3)
    for _ in range(200):
        n, world = rnd.randint(1, 400), rnd.randint(1, 16)
        args = (n, rnd.randint(0, 50), rnd.randrange(world), world, rnd.random() < 0.5, rnd.random() < 0.5,
                rnd.random() < 0.5)
        assert shard_indices(*args) == sampler_indices(*args), args
    # every sample appears exactly once per epoch when the split is padded and strided
    seen = sorted(i for r in range(8) for i in shard_indices(64, 4, r, 8, True, False, True))
    assert seen == list(range(64))
    # set_num_samples() on a sampler built with pad=False (n % world != 0): the reference's __iter__ then yields the
    # CEILING with wrap-around padding (datasets/sampler.py:78-110); restated here from its text
    for rank in range(4):
        s = DistributedSampler(list(range(10)), pad=False, permutation=True, num_replicas=4, rank=rank)
        s.set_epoch(7)
        assert len(list(s)) == 2
        s.set_num_samples()
        g = torch.Generator()
        g.manual_seed(7)
        order = torch.randperm(10, generator=g).tolist()
        order += order[:12 - 10]
        assert len(s) == 3 and list(s) == order[rank:12:4]


def test_oracle_nearest_matches_pillow():
    from oracle.data import resize_nearest
    for case, m in golden_masks():
        r = np.array(resize_nearest(m.tolist(), tuple(case["dst"])), dtype=np.uint8)
        assert tuple(r.shape) == tuple(case["dst"])
        assert checksums(r) == (case["sum"], case["weighted"]), case["src"]
        assert r[0].tolist()[:64] == case["first_row"] and r[:, -1].tolist()[:64] == case["last_col"]


def test_index_table_is_the_oracles():
    from oracle.data import pil_nearest_indices
    from semseg_amd.datasets import nearest_index_table
    rnd = random.Random(5)
    sizes = [(rnd.randint(1, 2100), rnd.randint(1, 2100)) for _ in range(300)] + [(1024, 2048), (2048, 1024), (1, 1)]
    for n_dst, n_src in sizes:
        assert nearest_index_table(n_dst, n_src).tolist() == pil_nearest_indices(n_dst, n_src), (n_dst, n_src)


def test_oracle_fast_hist_matches_reference():
    """oracle.data.fast_hist vs the confusion matrix of the real utils/misc.py:fast_hist."""
    from oracle.data import fast_hist
    with open(os.path.join(os.path.dirname(G), "fast_hist_golden.json")) as f:
        gold = json.load(f)
    rng = np.random.default_rng(gold["seed"])
    pred = rng.integers(0, 19, gold["n"])
    gt = rng.integers(0, 19, gold["n"])
    gt[rng.random(gold["n"]) < 0.1] = 255
    assert fast_hist(pred, gt, 19).tolist() == gold["hist"]


def test_oracle_pipeline_tail_matches_pil_and_torch():
    """oracle.data.crop_flip_normalize == PIL crop + FLIP_LEFT_RIGHT + ToTensor + Normalize +
    MaskToTensor (tests/golden/make_golden_pipeline.py), bit for bit in fp32."""
    import os
    import numpy as np
    import torch
    from oracle.data import crop_flip_normalize
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.pt"),
                   weights_only=False)
    for c in g["cases"]:
        im, lab = crop_flip_normalize(g["img"].numpy(), g["lab"].numpy(), c["window"], c["flip"], g["mean"], g["std"])
        assert np.array_equal(im, c["image"].numpy()), c["window"]
        assert np.array_equal(lab, c["labels"].numpy()), c["window"]


BICUBIC_CASES = [(37, 53, 74, 106), (64, 96, 48, 72), (50, 70, 100, 35), (33, 45, 67, 91), (128, 256, 205, 410),
                 (20, 20, 20, 31), (97, 131, 49, 66)]


def test_bicubic_oracle_is_pillow_bit_for_bit():
    """oracle.data.resize_bicubic_u8 (restatement of libImaging/Resample.c) against Pillow itself --
    `img.resize((w, h), Image.BICUBIC)`, the image half of the reference's scale step
    (transforms/joint_transforms.py:433-471) -- on up-, down- and mixed scalings."""
    import numpy as np
    from PIL import Image
    from oracle.data import resize_bicubic_u8
    rng = np.random.default_rng(0)
    for h, w, hd, wd in BICUBIC_CASES:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img, "RGB").resize((wd, hd), Image.BICUBIC))
        assert np.array_equal(resize_bicubic_u8(img, (hd, wd)), ref), (h, w, hd, wd)


def test_bicubic_tap_tables_of_the_product_are_the_oracles():
    import numpy as np
    from oracle.data import pil_bicubic_coeffs
    from semseg_amd.datasets.transforms import bicubic_tables
    for n_src, n_dst in [(53, 106), (96, 72), (70, 35), (45, 91), (256, 410), (20, 31), (131, 66), (2048, 1843),
                         (1024, 2049), (1024, 512), (7, 3), (3, 7), (2048, 1024), (1024, 717)]:
        k, b, c = pil_bicubic_coeffs(n_src, n_dst)
        k2, b2, c2 = bicubic_tables(n_dst, n_src)
        assert k == k2 and np.array_equal(np.array(b), b2) and np.array_equal(np.array(c), c2), (n_src, n_dst)


def test_device_prefetcher_passes_batches_through_in_order():
    import torch
    from semseg_amd.datasets.transforms import DevicePrefetcher
    batches = [(torch.full((2, 3), float(i)), torch.full((2,), i), "name%d" % i, 0.5) for i in range(5)]
    got = list(DevicePrefetcher(batches, device="cpu"))
    assert len(got) == 5 and len(DevicePrefetcher(batches, device="cpu")) == 5
    for i, (img, gts, name, scale) in enumerate(got):
        assert float(img[0, 0]) == i and int(gts[0]) == i and name == "name%d" % i and scale == 0.5
    assert list(DevicePrefetcher([], device="cpu")) == []
