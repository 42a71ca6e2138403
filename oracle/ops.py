"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU (device-agnostic torch, fp32/fp64) restatement of every operator on the
HRNet-OCR-MScale hot path of NVIDIA/semantic-segmentation, in the reference's
own NCHW convention.  Each function cites the reference file:line it follows.
It is pinned against the real reference (imported from /root/reference in the
build container) by tests/golden/make_golden.py -> tests/golden/*.pt and
tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.
"""
import torch
import torch.nn.functional as F

_CLIP_MIN = 1e-6   # loss/rmi.py:24
_POS_ALPHA = 5e-4  # loss/rmi.py:26


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1):
    """nn.Conv2d as used at network/hrnetv2.py:31-34, network/ocrnet.py:54-58."""
    return F.conv2d(x, w, b, stride=stride, padding=padding, dilation=dilation)


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """cfg.MODEL.BNFUNC = nn.BatchNorm2d (config.py:224, network/mynn.py:18-24).
    Updates running stats in place when training (unbiased running_var)."""
    return F.batch_norm(x, running_mean, running_var, gamma, beta, training, momentum, eps)


def bilinear(x, size):
    """F.interpolate(..., mode='bilinear', align_corners=False):
    network/mynn.py:42-84, network/hrnetv2.py:246-249,440-445."""
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


def resize_x(x, scale_factor):
    """ResizeX, network/mynn.py:101-114 (recompute_scale_factor=True)."""
    return F.interpolate(x, scale_factor=scale_factor, mode='bilinear', align_corners=False,
                         recompute_scale_factor=True)


def spatial_gather(feats, probs, scale=1):
    """SpatialGather_Module.forward, network/ocr_utils.py:34-46.
    feats [B,C,H,W], probs [B,K,H,W] -> [B,C,K,1]."""
    B, K = probs.size(0), probs.size(1)
    probs = probs.view(B, K, -1)
    feats = feats.view(B, feats.size(1), -1).permute(0, 2, 1)
    probs = F.softmax(scale * probs, dim=2)
    ctx = torch.matmul(probs, feats)
    return ctx.permute(0, 2, 1).unsqueeze(3)


def object_attention(query, key, value, key_channels):
    """Core of ObjectAttentionBlock.forward, network/ocr_utils.py:100-113.
    query [B,HW,D], key [B,D,K], value [B,K,D] -> context [B,HW,D]."""
    sim = torch.matmul(query, key)
    sim = (key_channels ** -.5) * sim
    sim = F.softmax(sim, dim=-1)
    return torch.matmul(sim, value)


def cross_entropy(logits, targets, ignore_index=255):
    """CrossEntropyLoss2d.forward, loss/utils.py:121-134."""
    return F.nll_loss(F.log_softmax(logits, dim=1), targets, ignore_index=ignore_index)


def _map_get_pairs(labels_4D, probs_4D, radius=3):
    """rmi_utils.map_get_pairs(is_combine=0), loss/rmi_utils.py:15-56."""
    h, w = labels_4D.shape[2], labels_4D.shape[3]
    new_h, new_w = h - (radius - 1), w - (radius - 1)
    la_ns, pr_ns = [], []
    for y in range(radius):
        for x in range(radius):
            la_ns.append(labels_4D[:, :, y:y + new_h, x:x + new_w])
            pr_ns.append(probs_4D[:, :, y:y + new_h, x:x + new_w])
    return torch.stack(la_ns, dim=2), torch.stack(pr_ns, dim=2)


def _log_det_by_cholesky(matrix):
    """loss/rmi_utils.py:95-107."""
    chol = torch.linalg.cholesky(matrix)
    return 2.0 * torch.sum(torch.log(torch.diagonal(chol, dim1=-2, dim2=-1) + 1e-8), dim=-1)


def rmi_lower_bound(labels_4D, probs_4D, num_classes, radius=3, pool=4):
    """RMILoss.rmi_lower_bound, loss/rmi.py:139-215 (rmi_pool_way=1, stride 4)."""
    half_d = radius * radius
    labels_4D = F.avg_pool2d(labels_4D, kernel_size=pool, stride=pool, padding=pool // 2)
    probs_4D = F.avg_pool2d(probs_4D, kernel_size=pool, stride=pool, padding=pool // 2)
    n, c = labels_4D.shape[0], labels_4D.shape[1]
    la_vectors, pr_vectors = _map_get_pairs(labels_4D, probs_4D, radius)
    la_vectors = la_vectors.reshape(n, c, half_d, -1).double()
    pr_vectors = pr_vectors.reshape(n, c, half_d, -1).double()
    diag = torch.eye(half_d, dtype=torch.float64, device=labels_4D.device)[None, None]
    la_vectors = la_vectors - la_vectors.mean(dim=3, keepdim=True)
    la_cov = torch.matmul(la_vectors, la_vectors.transpose(2, 3))
    pr_vectors = pr_vectors - pr_vectors.mean(dim=3, keepdim=True)
    pr_cov = torch.matmul(pr_vectors, pr_vectors.transpose(2, 3))
    pr_cov_inv = torch.inverse(pr_cov + diag * _POS_ALPHA)
    la_pr_cov = torch.matmul(la_vectors, pr_vectors.transpose(2, 3))
    appro_var = la_cov - torch.matmul(la_pr_cov.matmul(pr_cov_inv), la_pr_cov.transpose(-2, -1))
    rmi_now = 0.5 * _log_det_by_cholesky(appro_var + diag * _POS_ALPHA)
    rmi_per_class = rmi_now.view(-1, num_classes).mean(dim=0).float()
    rmi_per_class = rmi_per_class / float(half_d)
    return torch.sum(rmi_per_class)


def rmi_loss(logits_4D, labels_4D, num_classes, do_rmi=True, weight_lambda=0.5):
    """RMILoss.forward_sigmoid, loss/rmi.py:82-134 (lambda_way=1)."""
    label_mask_3D = labels_4D < num_classes
    onehot = F.one_hot(labels_4D.long() * label_mask_3D.long(), num_classes=num_classes).float()
    label_mask_3D = label_mask_3D.float()
    label_mask_flat = label_mask_3D.view(-1)
    onehot = onehot * label_mask_3D.unsqueeze(3)
    onehot_flat = onehot.view(-1, num_classes)
    logits_flat = logits_4D.permute(0, 2, 3, 1).contiguous().view(-1, num_classes)
    valid_pixels = torch.sum(label_mask_flat)
    binary_loss = F.binary_cross_entropy_with_logits(
        logits_flat, target=onehot_flat, weight=label_mask_flat.unsqueeze(1), reduction='sum')
    bce_loss = binary_loss / (valid_pixels + 1.0)
    if not do_rmi:
        return bce_loss
    probs_4D = logits_4D.sigmoid() * label_mask_3D.unsqueeze(1) + _CLIP_MIN
    onehot_4D = onehot.permute(0, 3, 1, 2)
    rmi = rmi_lower_bound(onehot_4D, probs_4D, num_classes)
    return weight_lambda * bce_loss + rmi * (1 - weight_lambda)


def max_pool_3x3_s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1), network/Resnet.py:147."""
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d(1), network/utils.py:201 (ASPP image pooling)."""
    return F.adaptive_avg_pool2d(x, 1)
