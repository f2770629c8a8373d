"""CPU: the training-side host functions of the three models against goldens from the REAL reference
(tests/golden/losses.npz, oracle/gen_golden_loss.py): PointPillars' anchor-target assignment and its three loss terms
(point_pillars.py:140-205, 842-943), the valid-label filter + class-weighted cross entropy of the segmentation models
(semseg_loss.py:7-38, randlanet.py:359-380, kpconv.py:315-351), and the optimizers' parameter groups."""
import os
import types

import numpy as np
import pytest
import torch

import synth_weights
from oracle.gen_golden_loss import loss_inputs

LOSS_CFG = {"focal": {"gamma": 2.0, "alpha": 0.25, "loss_weight": 1.0}, "smooth_l1": {"beta": 0.11, "loss_weight": 2.0},
            "cross_entropy": {"loss_weight": 0.2}}


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "losses.npz"))


@pytest.fixture(scope="module")
def pillars():
    from ml3d.torch.models import PointPillars
    return PointPillars(device="cpu", loss=LOSS_CFG, **synth_weights.POINTPILLARS_SMALL_CFG)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_pointpillars_assignment_and_losses_match_the_reference(golden, pillars, tag):
    from ml3d.torch.modules import assign_anchor_targets
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    maps, boxes, labels = loss_inputs(cfg, int(golden[tag + "_seed"]), tuple(int(v) for v in golden[tag + "_n_gt"]))
    head = pillars.bbox_head
    anchors = head.grid_anchors(maps[1].shape[-2:], "cpu")
    deltas, gt_idx, pos, neg = assign_anchor_targets(anchors, head.num_classes, len(head.rotations), head.iou_thr, boxes)
    assert np.array_equal(pos.numpy(), golden[tag + "_pos"]) and np.array_equal(neg.numpy(), golden[tag + "_neg"])
    assert np.array_equal(gt_idx.numpy(), golden[tag + "_gt_idx"])
    assert deltas.shape == golden[tag + "_deltas"].shape
    if deltas.numel():
        assert np.abs(deltas.numpy() - golden[tag + "_deltas"]).max() <= 1e-6
    inputs = types.SimpleNamespace(bboxes=boxes, labels=labels)
    l = pillars.get_loss(maps, inputs)
    got = np.array([float(l["loss_cls"]), float(l["loss_bbox"]), float(l["loss_dir"])])
    assert np.allclose(got, golden[tag + "_loss"], rtol=1e-5, atol=1e-6), (got, golden[tag + "_loss"])


def test_pointpillars_loss_is_differentiable_in_the_head_maps(pillars):
    maps, boxes, labels = loss_inputs(synth_weights.POINTPILLARS_SMALL_CFG, 5, (5, 0, 3))
    maps = tuple(m.clone().requires_grad_(True) for m in maps)
    l = pillars.get_loss(maps, types.SimpleNamespace(bboxes=boxes, labels=labels))
    sum(l.values()).backward()
    assert all(m.grad is not None and torch.isfinite(m.grad).all() and m.grad.abs().sum() > 0 for m in maps)


@pytest.mark.parametrize("tag,ign,fix", [("ign0", [0], lambda l: l), ("none", [], lambda l: l.clamp(max=7)),
                                         ("ign03", [0, 3], lambda l: (l + (l >= 3)).clamp(max=9))])
def test_valid_label_filter_and_weighted_cross_entropy_match_the_reference(golden, tag, ign, fix):
    from ml3d.torch.modules import valid_scores_and_labels
    g = torch.Generator().manual_seed(9)
    scores = torch.randn((2, 500, 8), generator=g)
    labels = fix(torch.randint(0, 9, (2, 500), generator=g))
    w = torch.rand(8, generator=g) + 0.5
    vs, vl = valid_scores_and_labels(scores, labels, 8, ign, "cpu")
    assert np.array_equal(vl.numpy(), golden["sem_" + tag + "_labels"])
    assert abs(float(vs.double().sum()) - float(golden["sem_" + tag + "_scores_sum"])) < 1e-9
    assert abs(float(torch.nn.CrossEntropyLoss(weight=w)(vs, vl)) - float(golden["sem_" + tag + "_loss"])) < 1e-6


def test_segmentation_models_get_loss_and_optimizers():
    from ml3d.torch.models import KPFCNN, RandLANet
    loss = types.SimpleNamespace(weighted_CrossEntropyLoss=torch.nn.CrossEntropyLoss())
    m = RandLANet(**dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG, num_points=1024, device="cpu"))
    scores = torch.randn((2, 64, 19))
    labels = torch.randint(0, 20, (2, 64))
    l, vl, vs = m.get_loss(loss, scores, {"data": {"labels": labels}}, "cpu")
    assert vs.shape[0] == vl.shape[0] == int((labels != 0).sum()) and vl.max() <= 18 and torch.isfinite(l)
    cfgp = types.SimpleNamespace(optimizer={"lr": 0.001}, scheduler_gamma=0.99)
    opt, sch = m.get_optimizer(cfgp)
    assert isinstance(opt, torch.optim.Adam) and sch.gamma == 0.99
    k = KPFCNN(**dict(synth_weights.TORONTO3D_CFG, device="cpu"))
    l, vl, vs = k.get_loss(loss, torch.randn((100, 8)), {"data": types.SimpleNamespace(labels=torch.randint(0, 9, (100,)))}, "cpu")
    assert torch.isfinite(l) and float(k.reg_loss) == 0.0
    opt, _ = k.get_optimizer(types.SimpleNamespace(learning_rate=0.01, deform_lr_factor=0.1, momentum=0.9, weight_decay=1e-3,
                                                   scheduler_gamma=0.99))
    assert len(opt.param_groups) == 2 and opt.param_groups[1]["lr"] == pytest.approx(0.001)
    # deformable blocks: the offset regulariser reads what the TRAINING forward left on them -- refused before there was one
    d = KPFCNN(**dict(synth_weights.KPCONV_DEFORM_SMALL_CFG, device="cpu"))
    with pytest.raises(RuntimeError):
        d.get_loss(loss, torch.randn((10, 5)), {"data": types.SimpleNamespace(labels=torch.randint(0, 6, (10,)))}, "cpu")


def test_randlanet_training_split_augmentation_is_the_reference_augmenters(golden):
    """RandLANet.transform for the TRAINING split adds the YAMLs' rotate / scale / noise (randlanet.py:198-203): same order, same
    draws from the model's generator, same float32 arithmetic as the reference's SemsegAugmentation (golden from the real class,
    oracle/gen_golden_loss.py) -- bit for bit; augmentations the in-scope YAMLs do not use are refused."""
    from ml3d.torch.models import RandLANet
    aug = {"recenter": {"dim": [0, 1]}, "normalize": {"feat": {"method": "linear", "bias": 0, "scale": 255}},
           "rotate": {"method": "vertical"}, "scale": {"min_s": 0.9, "max_s": 1.1}, "noise": {"noise_std": 0.001}}
    m = RandLANet(**dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG, num_points=1024, device="cpu", augment=aug))
    m.rng = np.random.default_rng(int(golden["aug_seed"]))
    out = m._training_augment(golden["aug_in"].copy())
    assert out.dtype == np.float32 and np.array_equal(out, golden["aug_out"])
    m2 = RandLANet(**dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG, num_points=1024, device="cpu",
                          augment={"RandomDropout": {"dropout_ratio": 0.2}}))
    with pytest.raises(NotImplementedError):
        m2._training_augment(golden["aug_in"].copy())
