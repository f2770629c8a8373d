"""Anchor <-> ground-truth assignment for the PointPillars training / validation loss.

Contract = ``Anchor3DHead.assign_bboxes`` (point_pillars.py:842-943) with its helpers ``box3d_to_bev2d`` / ``bbox_overlaps`` /
``BBoxCoder.encode`` (objdet_helper.py:102-126, 353-420, 259-284): per sample and per anchor class j with thresholds
(neg_th, pos_th), EVERY ground-truth box of the sample is matched against that class's anchors by the axis-aligned IoU of the
nearest-BEV rectangles; anchors with max IoU >= pos_th are positive, those below neg_th negative, and each ground truth whose
best IoU reaches neg_th makes all anchors tied at that IoU positive ("low-quality matching", later ground truths win the
first-best anchor).  Indices are flat over (sample, y, x, class, rotation).  Vectorised over the ground truths instead of the
reference's Python loop; same results."""
import math

import torch


def nearest_bev_boxes(boxes3d):
    """[n, 7] (x, y, z, w, l, h, yaw) -> [n, 4] axis-aligned (x0, y0, x1, y1): w / l swapped when the yaw, folded into
    [-pi/2, pi/2), is past 45 degrees."""
    yaw = boxes3d[:, 6]
    folded = (yaw - torch.floor(yaw / math.pi + 0.5) * math.pi).abs()
    wl = torch.where((folded > math.pi / 4)[:, None], boxes3d[:, [4, 3]], boxes3d[:, [3, 4]])
    c = boxes3d[:, :2]
    return torch.cat([c - wl / 2, c + wl / 2], dim=-1)


def pairwise_iou_xyxy(a, b, eps=1e-6):
    """[m, 4] x [n, 4] -> [m, n] IoU (union floored at eps)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = (area_a[:, None] + area_b[None, :] - inter).clamp(min=eps)
    return inter / union


def encode_boxes(anchors, targets):
    """Regression deltas that turn ``anchors`` into ``targets`` (the inverse of Anchor3DHead.decode)."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, wg, lg, hg, rg = torch.split(targets, 1, dim=-1)
    diag = torch.sqrt(la ** 2 + wa ** 2)
    return torch.cat([(xg - xa) / diag, (yg - ya) / diag, ((zg + hg / 2) - (za + ha / 2)) / ha, torch.log(wg / wa),
                      torch.log(lg / la), torch.log(hg / ha), rg - ra], dim=-1)


def assign_anchor_targets(anchors, num_classes, rotations, iou_thr, gt_boxes):
    """``anchors``: [H * W * num_classes * rotations, 7] in (y, x, class, rotation) order (Anchor3DHead.grid_anchors);
    ``iou_thr``: [(neg_th, pos_th)] per class (a single pair is shared); ``gt_boxes``: list of [n_i, 7] per sample.
    Returns (deltas [P, 7], gt index [P] into the concatenated ground truths, positive flat indices [P], negative flat
    indices [Q]) ordered by (sample, class, anchor) like the reference."""
    dev = anchors.device
    per_sample = anchors.shape[0]
    R = int(rotations)
    a = anchors.view(-1, num_classes, R, 7)
    thr = list(iou_thr) if len(iou_thr) == num_classes else [iou_thr[0]] * num_classes
    deltas, gt_idx, pos_all, neg_all = [], [], [], []
    off = 0
    empty_l = torch.zeros((0,), dtype=torch.long, device=dev)
    for i, gt in enumerate(gt_boxes):
        n = int(gt.shape[0])
        for j, (neg_th, pos_th) in enumerate(thr):
            if n == 0:
                deltas.append(torch.zeros((0, 7), device=dev))
                gt_idx.append(empty_l), pos_all.append(empty_l), neg_all.append(empty_l)
                continue
            aj = a[:, j].reshape(-1, 7)
            iou = pairwise_iou_xyxy(nearest_bev_boxes(gt), nearest_bev_boxes(aj))          # [n, A]
            best, who = iou.max(dim=0)                                                     # per anchor
            gt_best, gt_where = iou.max(dim=1)                                             # per ground truth
            pos = best >= pos_th
            neg = (best >= 0) & (best < neg_th)
            rescued = gt_best >= neg_th
            pos = pos | ((iou == gt_best[:, None]) & rescued[:, None]).any(dim=0)
            # the LAST rescued ground truth that names an anchor as its best keeps it
            k = torch.arange(n, device=dev)
            claim = torch.full((aj.shape[0],), -1, dtype=torch.long, device=dev)
            claim.scatter_reduce_(0, gt_where[rescued], k[rescued], reduce='amax', include_self=True)
            who = torch.where(claim >= 0, claim, who)
            p = torch.nonzero(pos).squeeze(-1)
            q = torch.nonzero(neg).squeeze(-1)
            deltas.append(encode_boxes(aj[p], gt[who[p]]))
            gt_idx.append(who[p] + off)

            def flat(idx):
                return (idx // R) * (num_classes * R) + j * R + idx % R + i * per_sample
            pos_all.append(flat(p))
            neg_all.append(flat(q))
        off += n
    return torch.cat(deltas, 0), torch.cat(gt_idx, 0), torch.cat(pos_all, 0), torch.cat(neg_all, 0)
