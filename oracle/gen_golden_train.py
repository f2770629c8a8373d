"""tests/golden/train_kpconv.npz: ONE training step's forward + backward of the REAL reference KPFCNN (rigid, small architecture)
on PyTorch-CPU -- model.train() (BatchNorm on batch statistics), cross entropy on seeded labels, loss.backward()
(semantic_segmentation.py:412-437) -- for the native training forward + hand-written KPConv backward to be held against
(tests/test_gpu_training.py).  Stored: logits, loss, the gradient of every KPConv weight tensor and of a few Linear / norm
parameters, one updated running mean.  Run from the repo root:  python -m oracle.gen_golden_train"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
import synth_data  # noqa: E402
import synth_weights  # noqa: E402

TRAIN_CFG = dict(synth_weights.TORONTO3D_CFG, first_features_dim=64, num_layers=3, in_features_dim=4,
                 architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb_strided", "resnetb",
                               "nearest_upsample", "unary", "nearest_upsample", "unary"])


def train_inputs(np_seed=31):
    spheres = [synth_data.toronto3d_sphere(61, 1800), synth_data.toronto3d_sphere(62, 1400)]
    rng = np.random.default_rng(5)
    cols = [np.concatenate([s, rng.random((len(s), 3), dtype=np.float32)], 1) for s in spheres]
    labels = [rng.integers(0, 9, len(s)).astype(np.int32) for s in spheres]
    return spheres, cols, labels


def main():
    os.chdir(tempfile.mkdtemp())
    ref_shim.reference_modules()
    from oracle import kpconv_ref as K
    kp = importlib.import_module("ml3d.torch.models.kpconv")
    cb = importlib.import_module("ml3d.torch.dataloaders.concat_batcher")
    sl = importlib.import_module("ml3d.torch.modules.losses.semseg_loss")
    cfg = dict(TRAIN_CFG)
    model = kp.KPFCNN(**cfg)
    sd = K.make_state_dict(cfg, 77)
    model.load_state_dict(sd)
    model.train()
    spheres, cols, labels = train_inputs()
    data = dict(p_list=spheres, f_list=cols, l_list=labels, p0_list=[np.zeros(3) for _ in spheres],
                s_list=[np.ones(3, np.float32) for _ in spheres], R_list=[np.eye(3, dtype=np.float32) for _ in spheres],
                r_inds_list=[np.zeros(0) for _ in spheres], r_mask_list=[np.zeros(0) for _ in spheres],
                val_labels_list=[np.zeros(0) for _ in spheres], cfg=model.cfg)
    model.cfg.batch_limit = 10 ** 9
    np.random.seed(31)
    batch = cb.KPConvBatch([{"data": data}])
    logits = model(batch)
    scores, lab = sl.filter_valid_label(logits, batch.labels, cfg["num_classes"], cfg["ignored_label_inds"], "cpu")
    loss = torch.nn.CrossEntropyLoss()(scores, lab)
    loss.backward()
    out = dict(logits=logits.detach().numpy(), loss=np.float64(loss.item()), n_valid=np.int64(len(lab)))
    named = dict(model.named_parameters())
    keep = [k for k in named if k.endswith("KPConv.weights")] + ["encoder_blocks.1.unary1.mlp.weight", "encoder_blocks.2.unary2.mlp.weight",
                                                                 "encoder_blocks.0.batch_norm.batch_norm.weight", "decoder_blocks.1.mlp.weight",
                                                                 "head_softmax.mlp.weight", "head_mlp.batch_norm.batch_norm.bias"]
    for k in keep:
        out["grad:" + k] = named[k].grad.numpy()
    out["running_mean:encoder_blocks.0"] = dict(model.named_buffers())["encoder_blocks.0.batch_norm.batch_norm.running_mean"].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_kpconv.npz"), **out)
    print("loss", loss.item(), "logits", logits.shape, "grads", len(keep), {k: float(np.abs(out["grad:" + k]).max()) for k in keep[:4]})


DEFORM_TRAIN_CFG = dict(synth_weights.KPCONV_DEFORM_SMALL_CFG, first_features_dim=32, modulated=True, in_features_dim=4,
                        deform_fitting_mode="point2point", deform_fitting_power=1.0, repulse_extent=1.2)


def deform_train_inputs():
    spheres = [synth_data.toronto3d_sphere(71, 700, radius=2.0), synth_data.toronto3d_sphere(72, 600, radius=2.0)]
    rng = np.random.default_rng(6)
    cols = [np.concatenate([s, rng.random((len(s), 3), dtype=np.float32)], 1) for s in spheres]
    labels = [rng.integers(0, 6, len(s)).astype(np.int32) for s in spheres]
    return spheres, cols, labels


def deform_main():
    """tests/golden/train_kpconv_deform.npz: one training forward + backward of the REAL reference KPFCNN with three DEFORMABLE,
    modulated blocks: logits, cross entropy AND the point-to-point offset regulariser of get_loss (kpconv.py:315-351, 2167-2206),
    gradients of the offset convolutions' weights / biases and of a spread of other parameters."""
    from oracle import kpconv_ref as K
    kp = importlib.import_module("ml3d.torch.models.kpconv")
    cb = importlib.import_module("ml3d.torch.dataloaders.concat_batcher")
    cfg = dict(DEFORM_TRAIN_CFG)
    np.random.seed(5)
    model = kp.KPFCNN(**cfg)
    model.load_state_dict(K.make_state_dict(cfg, 78))
    model.train()
    spheres, cols, labels = deform_train_inputs()
    data = dict(p_list=spheres, f_list=cols, l_list=labels, p0_list=[np.zeros(3) for _ in spheres],
                s_list=[np.ones(3, np.float32) for _ in spheres], R_list=[np.eye(3, dtype=np.float32) for _ in spheres],
                r_inds_list=[np.zeros(0) for _ in spheres], r_mask_list=[np.zeros(0) for _ in spheres],
                val_labels_list=[np.zeros(0) for _ in spheres], cfg=model.cfg)
    model.cfg.batch_limit = 10 ** 9
    np.random.seed(32)
    batch = cb.KPConvBatch([{"data": data}])
    logits = model(batch)
    Loss = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    loss, lab, scores = model.get_loss(Loss, logits, {"data": batch}, "cpu")
    loss.backward()
    named = dict(model.named_parameters())
    keep = [k for k in named if "offset" in k and named[k].grad is not None] + \
           [k for k in named if k.endswith("KPConv.weights") and "offset" not in k][:4] + ["head_softmax.mlp.weight"]
    out = dict(logits=logits.detach().numpy(), loss=np.float64(loss.item()), output_loss=np.float64(float(model.output_loss)),
               reg_loss=np.float64(float(model.reg_loss)), n_valid=np.int64(len(lab)))
    for k in keep:
        out["grad:" + k] = named[k].grad.numpy()
    # the reference's validation loop: an EVAL-mode forward (running statistics as the training forward above left them) whose
    # get_loss regularises the deformed kernel points of THAT forward (kpconv.py:1058,1074 run in eval mode too)
    model.eval()
    with torch.no_grad():
        logits_e = model(batch)
        model.get_loss(Loss, logits_e, {"data": batch}, "cpu")
    out.update(eval_logits=logits_e.numpy(), eval_output_loss=np.float64(float(model.output_loss)),
               eval_reg_loss=np.float64(float(model.reg_loss)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_kpconv_deform.npz"), **out)
    print("eval ce", float(model.output_loss), "eval reg", float(model.reg_loss))
    print("deform loss", loss.item(), "ce", float(model.output_loss), "reg", float(model.reg_loss), "grads", len(keep),
          {k: float(np.abs(out["grad:" + k]).max()) for k in keep[:3]})


RANDLA_TRAIN_CFG = dict(num_neighbors=16, num_layers=3, num_points=1024, num_classes=8, sub_sampling_ratio=[4, 4, 2], in_channels=6,
                        dim_features=8, dim_output=[16, 32, 64], ignored_label_inds=[0], grid_size=0.06)


def randla_train_inputs():
    rng = np.random.default_rng(8)
    pts = np.stack([synth_data.semantickitti_patch(70 + b, 1024) for b in range(2)])
    feats = np.concatenate([pts, rng.random((2, 1024, 3), dtype=np.float32)], 2)
    labels = rng.integers(0, 9, (2, 1024)).astype(np.int64)
    return pts, feats, labels


def randla_main():
    """tests/golden/train_randlanet.npz: the same for the REAL reference RandLANet (train mode; fc1's Dropout switched to eval on
    both sides: its mask is a device-specific random stream)."""
    from oracle import ops as oops
    from oracle import randlanet_ref as R
    rl = importlib.import_module("ml3d.torch.models.randlanet")
    sl = importlib.import_module("ml3d.torch.modules.losses.semseg_loss")
    cfg = dict(RANDLA_TRAIN_CFG)
    model = rl.RandLANet(**cfg)
    model.load_state_dict(R.make_state_dict(cfg, 55))
    model.device = torch.device("cpu")
    model.train()
    model.fc1[2].eval()
    pts, feats, labels = randla_train_inputs()
    inp = R.build_inputs(pts, feats, cfg, oops.knn_search)
    logits = model(inp)
    scores, lab = sl.filter_valid_label(logits, torch.from_numpy(labels), cfg["num_classes"], cfg["ignored_label_inds"], "cpu")
    loss = torch.nn.CrossEntropyLoss()(scores, lab)
    loss.backward()
    named = dict(model.named_parameters())
    keep = ["fc0.weight", "bn0.weight", "encoder.0.mlp1.conv.weight", "encoder.0.lse1.mlp.conv.weight", "encoder.0.pool1.score_fn.0.weight",
            "encoder.1.pool2.mlp.conv.weight", "encoder.1.lse2.mlp.batch_norm.bias", "encoder.2.shortcut.conv.weight", "mlp.conv.weight",
            "decoder.0.conv.weight", "decoder.2.conv.bias", "fc1.3.conv.weight"]
    out = dict(logits=logits.detach().numpy(), loss=np.float64(loss.item()), n_valid=np.int64(len(lab)))
    for k in keep:
        out["grad:" + k] = named[k].grad.numpy()
    out["running_mean:bn0"] = model.bn0.running_mean.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_randlanet.npz"), **out)
    print("randla loss", loss.item(), "logits", logits.shape, {k: float(np.abs(out["grad:" + k]).max()) for k in keep[:4]})


# the widths of randlanet_semantickitti.yml (the four stage widths 16 / 64 / 128 / 256 of the fused training kernels' LDS classes), at a size
# PyTorch-CPU turns around in seconds
RANDLA_WIDE_TRAIN_CFG = dict(num_neighbors=16, num_layers=4, num_points=1024, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
                             dim_features=8, dim_output=[16, 64, 128, 256], ignored_label_inds=[0], grid_size=0.06)


def randla_wide_train_inputs():
    rng = np.random.default_rng(9)
    pts = np.stack([synth_data.semantickitti_patch(90 + b, 1024) for b in range(3)])
    labels = rng.integers(0, 20, (3, 1024)).astype(np.int64)
    return pts, pts.copy(), labels


def randla_wide_main():
    """tests/golden/train_randlanet_wide.npz: forward + backward of the REAL reference RandLANet with the SemanticKITTI widths
    (dim_output 16 / 64 / 128 / 256), 3 x 1024 points: logits, loss, the gradient of EVERY parameter's largest entry position-wise
    for a spread of tensors across all four levels (the score Linears of all eight attentive poolings included)."""
    from oracle import ops as oops
    from oracle import randlanet_ref as R
    rl = importlib.import_module("ml3d.torch.models.randlanet")
    sl = importlib.import_module("ml3d.torch.modules.losses.semseg_loss")
    cfg = dict(RANDLA_WIDE_TRAIN_CFG)
    model = rl.RandLANet(**cfg)
    model.load_state_dict(R.make_state_dict(cfg, 56))
    model.device = torch.device("cpu")
    model.train()
    model.fc1[2].eval()
    pts, feats, labels = randla_wide_train_inputs()
    inp = R.build_inputs(pts, feats, cfg, oops.knn_search)
    logits = model(inp)
    scores, lab = sl.filter_valid_label(logits, torch.from_numpy(labels), cfg["num_classes"], cfg["ignored_label_inds"], "cpu")
    loss = torch.nn.CrossEntropyLoss()(scores, lab)
    loss.backward()
    named = dict(model.named_parameters())
    keep = ["fc0.weight"]
    for l in range(4):
        keep += ["encoder.%d.pool1.score_fn.0.weight" % l, "encoder.%d.pool2.score_fn.0.weight" % l, "encoder.%d.pool1.score_fn.0.bias" % l,
                 "encoder.%d.lse1.mlp.conv.weight" % l, "encoder.%d.lse2.mlp.conv.weight" % l, "encoder.%d.mlp1.conv.weight" % l,
                 "encoder.%d.pool2.mlp.batch_norm.weight" % l, "encoder.%d.shortcut.conv.bias" % l]
    keep += ["mlp.conv.bias", "mlp.batch_norm.weight", "decoder.0.conv.bias", "decoder.0.batch_norm.bias", "decoder.3.conv.weight", "fc1.3.conv.weight"]   # (the 512 x 512 / 768 x 256 matrices would be megabytes)
    out = dict(logits=logits.detach().numpy(), loss=np.float64(loss.item()), n_valid=np.int64(len(lab)))
    for k in keep:
        out["grad:" + k] = named[k].grad.numpy()
    out["running_var:encoder.3.pool2.mlp"] = model.encoder[3].pool2.mlp.batch_norm.running_var.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_randlanet_wide.npz"), **out)
    print("randla wide loss", loss.item(), "logits", logits.shape, len(keep), "gradients",
          {k: float(np.abs(out["grad:" + k]).max()) for k in keep[1:5]})


PP_LOSS_CFG = {"focal": {"gamma": 2.0, "alpha": 0.25, "loss_weight": 1.0}, "smooth_l1": {"beta": 0.11, "loss_weight": 2.0},
               "cross_entropy": {"loss_weight": 0.2}}


def pp_train_inputs():
    """Two synthetic sweeps cropped to the small config's range (xyz only: in_channels 3) + ground-truth boxes."""
    from oracle import pointpillars_ref as P
    from oracle.gen_golden_loss import loss_inputs
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(80 + i), cfg)[:, :3].copy() for i in range(2)]
    _, boxes, labels = loss_inputs(cfg, 12, (4, 6))
    return clouds, boxes, labels


def pointpillars_main():
    """tests/golden/train_pointpillars.npz: ONE training forward + backward of the REAL reference PointPillars (small two-PFN-layer
    config, train mode: BatchNorm on batch statistics, the training-side max_voxels), the sum of the three ``get_loss`` terms
    (object_detection.py:273-283), loss.backward(): head maps (strided), loss terms, gradients of a spread of parameters."""
    from oracle import pointpillars_ref as P
    pp = importlib.import_module("ml3d.torch.models.point_pillars")
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    model = pp.PointPillars(device="cpu", augment={}, loss=PP_LOSS_CFG, **cfg)
    model.load_state_dict(P.make_state_dict(cfg, 21))
    model.train()
    clouds, boxes, labels = pp_train_inputs()

    class _In:
        point = [torch.from_numpy(c) for c in clouds]
        bboxes = boxes
    _In.labels = labels
    maps = model(_In)
    terms = model.get_loss(maps, _In)
    loss = sum(terms.values())
    loss.backward()
    named = dict(model.named_parameters())
    keep = ["voxel_encoder.pfn_layers.0.linear.weight", "voxel_encoder.pfn_layers.1.norm.weight", "backbone.blocks.0.0.weight",
            "backbone.blocks.1.3.weight", "backbone.blocks.2.1.bias", "neck.deblocks.0.0.weight", "neck.deblocks.2.0.weight",
            "neck.deblocks.1.1.weight", "bbox_head.conv_cls.weight", "bbox_head.conv_reg.bias", "bbox_head.conv_dir_cls.weight"]
    out = dict(cls=maps[0].detach().numpy()[:, :, ::2, ::2], reg=maps[1].detach().numpy()[:, :, ::2, ::2],
               dir=maps[2].detach().numpy()[:, :, ::2, ::2],
               loss=np.array([float(terms["loss_cls"]), float(terms["loss_bbox"]), float(terms["loss_dir"])], np.float64),
               n_points=np.asarray([len(c) for c in clouds]))
    for k in keep:
        out["grad:" + k] = named[k].grad.numpy()
    out["running_mean:backbone.blocks.0.1"] = dict(model.named_buffers())["backbone.blocks.0.1.running_mean"].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_pointpillars.npz"), **out)
    print("pointpillars loss", out["loss"], "maps", maps[0].shape, {k: float(np.abs(out["grad:" + k]).max()) for k in keep[:4]})


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "kpconv"):
        main()
    else:
        os.chdir(tempfile.mkdtemp())
        ref_shim.reference_modules()
    if which in ("all", "randlanet"):
        randla_main()
    if which in ("all", "randlanet_wide"):
        randla_wide_main()
    if which in ("all", "pointpillars"):
        pointpillars_main()
    if which in ("all", "deform"):
        deform_main()
