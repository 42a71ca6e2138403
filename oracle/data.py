"""ORACLE -- test infrastructure only.  Never imported by the product package.

Plain-Python restatement of the two integer pieces of the reference's input
pipeline that SURVEY.md section 8a lists on the hot path's boundary:
  S1  datasets/sampler.py:78-100   DistributedSampler.__iter__
  S2  transforms/joint_transforms.py:193 (and 267, 290, 319, 339, 364, 466)
      mask.resize(size, Image.NEAREST)  -- Pillow's ImagingScaleAffine
Pinned by tests/golden/data_golden.json (generated from the real reference /
from Pillow by tests/golden/make_golden_data.py)."""
import math

import torch


def sampler_indices(n, epoch, rank, world, pad=False, consecutive_sample=False, permutation=False):
    """datasets/sampler.py:78-100, statement by statement."""
    num_samples = int(math.ceil(n * 1.0 / world)) if pad else int(math.floor(n * 1.0 / world))
    total_size = num_samples * world                       # sampler.py:71-75
    g = torch.Generator()
    g.manual_seed(epoch)                                   # sampler.py:80-81
    if permutation:
        indices = [int(v) for v in torch.randperm(n, generator=g)]   # sampler.py:84
    else:
        indices = [x for x in range(n)]                    # sampler.py:86
    if total_size > len(indices):
        indices += indices[:(total_size - len(indices))]   # sampler.py:89-90
    if consecutive_sample:
        offset = num_samples * rank                        # sampler.py:93-95
        indices = indices[offset:offset + num_samples]
    else:
        indices = indices[rank:total_size:world]           # sampler.py:97
    assert len(indices) == num_samples
    return indices


def pil_nearest_indices(n_dst, n_src):
    """Source index per destination index of Pillow's NEAREST resize
    (libImaging/Geometry.c ImagingScaleAffine: xo = a0*0.5; per pixel
    xin = (int)xo; xo += a0, all in double)."""
    a0 = float(n_src) / float(n_dst)
    xo = a0 * 0.5
    out = []
    for _ in range(n_dst):
        xin = -1 if xo < 0.0 else int(xo)
        out.append(min(max(xin, 0), n_src - 1))
        xo += a0
    return out


def resize_nearest(mask, size):
    """mask: 2-D list/array of ints [Hs][Ws]; size = (Hd, Wd) -> nested lists."""
    hs, ws = len(mask), len(mask[0])
    iy, ix = pil_nearest_indices(size[0], hs), pil_nearest_indices(size[1], ws)
    return [[int(mask[y][x]) for x in ix] for y in iy]


def fast_hist(pred, gtruth, num_classes):
    """utils/misc.py:50-67, statement by statement (numpy)."""
    import numpy as np
    mask = (gtruth >= 0) & (gtruth < num_classes)
    hist = np.bincount(num_classes * gtruth[mask].astype(int) + pred[mask], minlength=num_classes ** 2)
    return hist.reshape(num_classes, num_classes)


def eval_predictions(output):
    """utils/trnval_utils.py:173-174: softmax over classes, then max(1) -> class ids."""
    import torch
    probs = torch.nn.functional.softmax(output, dim=1)
    return probs.max(1)[1]


def crop_flip_normalize(img_u8, labels_u8, window, flip, mean, std):
    """CPU restatement of the tail of the reference's input pipeline for one sample
    (datasets/base_loader.py:120-150): the crop transforms' `img.crop((x1, y1, x1+tw, y1+th))`
    (transforms/joint_transforms.py:88-90, 156-157), RandomHorizontallyFlip
    (joint_transforms.py:276-281), torchvision's ToTensor (uint8 HWC -> float32 CHW / 255) and
    Normalize ((t - mean) / std, fp32), MaskToTensor (transforms/transforms.py: int64 labels).
    img_u8: numpy uint8 [H,W,3]; labels_u8: numpy uint8 [H,W].  Returns (float32 [3,h,w], int64 [h,w])."""
    import numpy as np
    x0, y0, w, h = window
    im = img_u8[y0:y0 + h, x0:x0 + w]
    lab = labels_u8[y0:y0 + h, x0:x0 + w]
    if flip:
        im, lab = im[:, ::-1], lab[:, ::-1]
    t = np.ascontiguousarray(im.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (t - m) / s, np.ascontiguousarray(lab).astype(np.int64)


# ---------------------------------------------------------------------------------------------
# S2': the IMAGE half of the scale step, `img.resize((w, h), Image.BICUBIC)`
# (transforms/joint_transforms.py:433-471 RandomSizeAndCrop -> scale_and_crop, and the Scale /
# ResizeHeight transforms): Pillow 8-bit two-pass resampling, libImaging/Resample.c
# (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc),
# restated scalar by scalar.  Pinned against Pillow itself in tests/test_data_cpu.py.
# ---------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size, out_size):
    """-> (ksize, bounds [(xmin, xmax)], integer coefficients [out_size][ksize]) of Resample.c for the
    whole-image box (in0 = 0, in1 = in_size) and the bicubic filter (support 2.0)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws, ww = [], 0.0
        for x in range(xmax):
            w = _bicubic_filter((x + xmin - center + 0.5) * ss)
            ws.append(w)
            ww += w
        if ww != 0.0:
            ws = [w / ww for w in ws]
        ws += [0.0] * (ksize - xmax)
        kk.append([int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in ws])
        bounds.append((xmin, xmax))
    return ksize, bounds, kk


def _clip8(v):
    v >>= PRECISION_BITS               # arithmetic shift, as the C `in >> PRECISION_BITS` on int
    return 0 if v < 0 else (255 if v > 255 else v)


def resize_bicubic_u8(img, size):
    """img: numpy uint8 [H][W][C]; size = (Hd, Wd).  Horizontal pass into an 8-bit image, then the
    vertical pass over it (ImagingResampleInner)."""
    import numpy as np
    hs, ws, ch = img.shape
    hd, wd = size
    src = img.astype(np.int64)
    if wd != ws:
        _, bounds, kk = pil_bicubic_coeffs(ws, wd)
        tmp = np.empty((hs, wd, ch), dtype=np.int64)
        for xx, ((xmin, xmax), k) in enumerate(zip(bounds, kk)):
            acc = np.full((hs, ch), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for x in range(xmax):
                acc += src[:, xmin + x, :] * k[x]
            tmp[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        src = tmp
    if hd != hs:
        _, bounds, kk = pil_bicubic_coeffs(hs, hd)
        out = np.empty((hd, src.shape[1], ch), dtype=np.int64)
        for yy, ((ymin, ymax), k) in enumerate(zip(bounds, kk)):
            acc = np.full((src.shape[1], ch), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for y in range(ymax):
                acc += src[ymin + y, :, :] * k[y]
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        src = out
    return src.astype(np.uint8)
