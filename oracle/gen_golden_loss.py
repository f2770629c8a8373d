"""tests/golden/losses.npz: the training-side host functions pinned to the REAL reference.
  * ``PointPillars.get_loss`` (point_pillars.py:140-205) with ``Anchor3DHead.assign_bboxes`` (:842-943) on seeded head maps and
    ground-truth boxes, small two-class config: the three loss terms + the assignment (deltas, gt index, positive / negative
    flat indices);
  * ``filter_valid_label`` (modules/losses/semseg_loss.py:7-38) + the class-weighted cross entropy of ``SemSegLoss`` for
    RandLA-Net-shaped scores with ignored labels [0] and [] and [0, 3].
Run from the repo root:  python -m oracle.gen_golden_loss"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
import synth_weights  # noqa: E402


def loss_inputs(cfg, seed, n_gt=(5, 0, 3)):
    """Seeded head maps + ground truth; the test side regenerates exactly this."""
    g = torch.Generator().manual_seed(seed)
    H, W = [s // 2 for s in cfg["scatter"]["output_shape"]]
    nc, na = len(cfg["classes"]), len(cfg["head"]["sizes"]) * len(cfg["head"]["rotations"])
    B = len(n_gt)
    maps = (torch.randn((B, na * nc, H, W), generator=g), torch.randn((B, na * 7, H, W), generator=g) * 0.3,
            torch.randn((B, na * 2, H, W), generator=g))
    r = cfg["point_cloud_range"]
    boxes, labels = [], []
    for n in n_gt:
        u = torch.rand((n, 7), generator=g)
        sizes = torch.tensor(cfg["head"]["sizes"])[torch.randint(0, nc, (n,), generator=g)] if n else torch.zeros((0, 3))
        b = torch.cat([r[0] + u[:, :1] * (r[3] - r[0]), r[1] + u[:, 1:2] * (r[4] - r[1]), -1.5 + 0.5 * u[:, 2:3],
                       sizes * (0.8 + 0.4 * u[:, 3:6]), (u[:, 6:7] - 0.5) * 6.0], 1)
        boxes.append(b.float())
        labels.append(torch.randint(0, nc + 1, (n,), generator=g))          # (nc = a class the head does not model)
    return maps, boxes, labels


def main():
    os.chdir(tempfile.mkdtemp())
    ref_shim.reference_modules()
    pp = importlib.import_module("ml3d.torch.models.point_pillars")
    sl = importlib.import_module("ml3d.torch.modules.losses.semseg_loss")
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    loss_cfg = {"focal": {"gamma": 2.0, "alpha": 0.25, "loss_weight": 1.0}, "smooth_l1": {"beta": 0.11, "loss_weight": 2.0},
                "cross_entropy": {"loss_weight": 0.2}}
    model = pp.PointPillars(device="cpu", augment={}, loss=loss_cfg, **cfg)
    out = {}
    for tag, seed, n_gt in (("a", 5, (5, 0, 3)), ("b", 6, (0, 0)), ("c", 7, (12,))):
        maps, boxes, labels = loss_inputs(cfg, seed, n_gt)

        class _In:
            bboxes, pass_ = boxes, None
        _In.labels = labels
        tb, ti, pi, ni = model.bbox_head.assign_bboxes(maps[1], boxes)
        l = model.get_loss(maps, _In)
        out.update({tag + "_deltas": tb.numpy(), tag + "_gt_idx": ti.numpy(), tag + "_pos": pi.numpy(), tag + "_neg": ni.numpy(),
                    tag + "_loss": np.array([float(l["loss_cls"]), float(l["loss_bbox"]), float(l["loss_dir"])], np.float64),
                    tag + "_seed": seed, tag + "_n_gt": np.asarray(n_gt)})
        print(tag, [float(v) for v in l.values()], len(pi), len(ni))
    g = torch.Generator().manual_seed(9)
    scores = torch.randn((2, 500, 8), generator=g)
    labels = torch.randint(0, 9, (2, 500), generator=g)
    w = torch.rand(8, generator=g) + 0.5
    for tag, ign, nc, lab in (("ign0", [0], 8, labels), ("none", [], 8, labels.clamp(max=7)), ("ign03", [0, 3], 8, (labels + (labels >= 3)).clamp(max=9))):
        vs, vl = sl.filter_valid_label(scores, lab, nc, ign, "cpu")
        out["sem_" + tag + "_labels"] = vl.numpy()
        out["sem_" + tag + "_scores_sum"] = np.float64(vs.double().sum())
        out["sem_" + tag + "_loss"] = np.float64(torch.nn.CrossEntropyLoss(weight=w)(vs, vl))
    # the training-split augmentations of the RandLA-Net YAMLs (rotate / scale / noise) through the reference's own augmenter,
    # drawing from a seeded generator the way RandLANet.transform hands its model generator over (randlanet.py:198-203)
    aug = importlib.import_module("ml3d.datasets.augment.augmentation")
    pc = (np.random.default_rng(3).random((700, 3)).astype(np.float32) - 0.5) * np.float32([20, 20, 4])
    pc[:, :2] -= pc[:, :2].mean(0)
    train_cfg = {"rotate": {"method": "vertical"}, "scale": {"min_s": 0.9, "max_s": 1.1}, "noise": {"noise_std": 0.001}}
    a = aug.SemsegAugmentation(train_cfg)
    apc, _, _ = a.augment(pc.copy(), None, np.zeros(700, np.int32), train_cfg, seed=np.random.default_rng(41))
    out.update(aug_in=pc, aug_out=apc, aug_seed=41)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "losses.npz"), **out)
    print("written")


if __name__ == "__main__":
    main()
