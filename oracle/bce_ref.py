"""TEST INFRASTRUCTURE (imported from tests/ only): numpy restatement of the reference's BCEWithLogitsLoss with an ignore
value, ``torchok/losses/classification/binary_cross_entropy.py:50-59``:

    target = target.float(); input = input[target != ignore_index].float(); target = target[target != ignore_index]
    selected count > 0 -> F.binary_cross_entropy_with_logits(input, target, reduction) else 0

with ATen's element formula (1 - t) * x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|)).
Pinned by tests/golden/bce_loss.npz (outputs + input gradients of the reference's own class, tests/golden/gen_golden.py)."""
import numpy as np


def bce_with_logits_ignore(x: np.ndarray, t: np.ndarray, ignore_index=-1, reduction='mean'):
    """Returns (loss, d loss / d x) in float64."""
    x = np.asarray(x, np.float64)
    t = np.asarray(t, np.float64)
    sel = t != ignore_index
    n = int(sel.sum())
    if n == 0:
        return 0.0, np.zeros_like(x)
    el = (1.0 - t) * x - (np.minimum(x, 0.0) - np.log1p(np.exp(-np.abs(x))))
    scale = 1.0 / n if reduction == 'mean' else 1.0
    loss = float(el[sel].sum() * scale)
    grad = np.where(sel, (1.0 / (1.0 + np.exp(-x)) - t) * scale, 0.0)
    return loss, grad
