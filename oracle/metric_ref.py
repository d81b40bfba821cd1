"""CPU restatement of the metric-learning rows of the hot path: fp32, plain
PyTorch ops, each function citing the reference lines it follows.  TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/metric_heads.npz, which tests/golden/gen_golden.py writes by running the reference's
own arcface_head.py / linear_head.py / pairwise.py (and the body of calc_relevance_matrix)."""
import math

import torch
import torch.nn.functional as F


def arcface_defaults(in_channels: int, num_classes: int):
    """arcface_head.py:46-56 — default scale and margin."""
    p = .999
    c_1 = num_classes - 1
    scale = c_1 / num_classes * math.log(c_1 * p / (1 - p)) + 1
    margin = (.9 - math.cos(2 * math.pi / num_classes)) if in_channels == 2 else .5 * num_classes / (num_classes - 1)
    return scale, margin


def arcface_forward(x, weight, target, margin, scale, easy_margin=False, training=True):
    """arcface_head.py:95-131."""
    if not training:
        return F.linear(x, weight)                                   # :120-121
    if target is None:
        raise ValueError('Target is None in training mode.')         # :122-123
    cosine = F.linear(F.normalize(x), F.normalize(weight))           # :125-127
    cos_m, sin_m = math.cos(margin), math.sin(margin)                # :75-78
    th, mm = math.cos(math.pi - margin), math.sin(math.pi - margin) * margin
    sine = torch.sqrt((1.0 - torch.pow(cosine, 2)).clamp(0, 1))      # :96
    phi = (cosine * cos_m - sine * sin_m).type(cosine.dtype)         # :97
    phi = torch.where(cosine > 0, phi, cosine) if easy_margin else torch.where(cosine > th, phi, cosine - mm)
    one_hot = torch.zeros_like(cosine)
    one_hot.scatter_(1, target.view(-1, 1).long(), 1)                # :103-104
    return torch.where(one_hot == 1, phi, cosine) * scale            # :105-106


def linear_head_forward(x, weight, bias, normalize):
    """linear_head.py:27-36 (drop_rate = 0)."""
    y = F.linear(x, weight, bias)
    return F.normalize(y, p=2, dim=-1) if normalize else y


def relevance_matrix(y, num_classes):
    """pairwise_task.py:87-107."""
    if y.ndim == 1:
        y = torch.zeros(y.shape[0], num_classes).scatter_(1, y[:, None], 1)
    inter = torch.matmul(y, y.transpose(1, 0))
    return torch.where(inter > 0, 1., 0.)


def contrastive_loss(emb1, emb2, R, margin, reduction='mean', reg=None, eps=1e-3):
    """pairwise.py:126-136 (calc_loss) + :28-46 (regularize: eps * sum|emb1| or eps * ||emb1||_2 per row) + :48-64."""
    S = torch.cdist(emb1, emb2, p=2)
    L = ((1. - R) * F.relu(margin - S).pow(2) + R * S.pow(2)).sum(1)
    if reg == 'L1':
        L = L + eps * emb1.abs().sum(1)
    elif reg == 'L2':
        L = L + eps * torch.norm(emb1, p=None, dim=1)
    elif reg is not None:
        raise ValueError(f'Unknown regularization type: {reg}')
    return L.mean() if reduction == 'mean' else L.sum()
