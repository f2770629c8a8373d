"""Loss terms of the three hot-path models.

Contracts (observable through ``model.get_loss``): ml3d/torch/modules/losses/focal_loss.py:24-63, smooth_L1.py:24-52,
cross_entropy.py:22-46 (PointPillars, point_pillars.py:140-205) and ``filter_valid_label`` of semseg_loss.py:7-38
(RandLA-Net / KPFCNN, randlanet.py:359-380, kpconv.py:315-351).  ``avg_factor`` semantics: None -> mean; > 0 -> sum / factor;
0 -> the focal term stays un-reduced (its caller sums it), the other two fall back to the mean."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reduce(loss, avg_factor, keep_when_zero=False):
    if avg_factor is None:
        return loss.mean()
    if avg_factor > 0:
        return loss.sum() / avg_factor
    return loss if keep_when_zero else loss.mean()


class FocalLoss(nn.Module):
    """Sigmoid focal loss over one-hot targets; a target equal to the number of classes is background (all zeros)."""

    def __init__(self, gamma=2.0, alpha=0.25, loss_weight=1.0):
        super().__init__()
        self.gamma, self.alpha, self.loss_weight = gamma, alpha, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        p = pred.sigmoid()
        if pred.dim() > 1:
            target = (target.unsqueeze(-1) == torch.arange(pred.shape[-1], device=target.device)).float()
        t = target.type_as(pred)
        miss = (1 - p) * t + p * (1 - t)                                # probability mass on the wrong side
        w = (self.alpha * t + (1 - self.alpha) * (1 - t)) * miss.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * w
        if weight is not None:
            loss = loss * weight
        return _reduce(loss * self.loss_weight, avg_factor, keep_when_zero=True)


class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, loss_weight=1.0):
        super().__init__()
        self.beta, self.loss_weight = beta, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, **kwargs):
        assert pred.size() == target.size() and target.numel() > 0
        d = (pred - target).abs()
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        if weight is not None:
            loss = loss * weight
        return _reduce(loss * self.loss_weight, avg_factor if avg_factor else None)


class CrossEntropyLoss(nn.Module):

    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, **kwargs):
        loss = F.cross_entropy(cls_score, label, reduction='none')
        if weight is not None:
            loss = loss * weight
        return _reduce(loss * self.loss_weight, avg_factor if avg_factor else None)


def valid_scores_and_labels(scores, labels, num_classes, ignored_label_inds, device):
    """semseg_loss.py:7-38: drop the points whose label is ignored and renumber the remaining labels into 0..num_classes-1
    (every ignored label id below a label shifts it down by one).  Returns (scores [M, C], labels [M])."""
    s = scores.reshape(-1, num_classes).to(device)
    l = labels.reshape(-1).to(device).long()
    ign = sorted(int(i) for i in ignored_label_inds)
    keep = torch.ones_like(l, dtype=torch.bool)
    for i in ign:
        keep &= l != i
    l = l[keep]
    # the reference inserts a zero into an identity table at every ignored index, in list order: label v maps to
    # v - #(inserted entries at positions <= v in the grown table)
    table = list(range(num_classes))
    for i in ignored_label_inds:
        if i >= 0:
            table = table[:i] + [0] + table[i:]
    lut = torch.tensor(table, dtype=torch.int64, device=l.device)
    return s[keep], lut[l]
