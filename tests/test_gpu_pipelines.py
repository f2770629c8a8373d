"""GPU: whole-cloud inference through the model-class API the reference's pipelines drive — ``preprocess -> transform ->
batcher -> forward -> update_probs`` (segmentation) and ``forward -> inference_end`` (detection) — at the sizes of the
unchanged YAML configs (45 056-point patches; in_radius 4.0 m / 10 000-point spheres; the KITTI range), against
tests/golden/pipeline_*.npz: the labels / votes / boxes the REAL reference pipeline classes produced with the reference's
PyTorch-CPU models (oracle/gen_golden_pipeline.py).  The loop itself is tests/pipeline_loop.py, pinned to be identical to
the checkout's ``run_inference`` by that generator.  No checkout is needed here (the GPU box has none)."""
import json
import os

import numpy as np
import pytest
import torch

import pipeline_loop as PL
import synth_data
import synth_weights

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _seed(s):
    import random
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def _golden(name):
    g = np.load(os.path.join(GOLD, "pipeline_%s.npz" % name))
    return g, json.loads(str(g["model_cfg_json"]))


def native_segmentation(name, cfg, data, batch_size, device):
    """The native class through the pipeline loop; ``cfg`` = the model section of the YAML."""
    from ml3d.torch.dataloaders import ConcatBatcher, DefaultBatcher
    from ml3d.torch.models import KPFCNN, RandLANet
    _seed(7)
    if name == "randlanet":
        model = RandLANet(**cfg, device=device)
        model.load_state_dict(synth_weights.randlanet_state_dict(cfg, 31))
        collate = DefaultBatcher().collate_fn
    else:
        # sampler_index='sklearn': the radius sampler's sphere order is the traversal order of sklearn's tree (see
        # ml3d/torch/models/_datapath.py:GpuSearchTree) -- bit-identical spheres need the reference's own index for that query
        model = KPFCNN(**cfg, device=device, sampler_index="sklearn")
        model.load_state_dict(synth_weights.kpconv_state_dict(cfg, 32))

        collate = ConcatBatcher(device, "KPFCNN").collate_fn           # GPU build of the neighbour / pooling matrices
    model.eval()
    _seed(11)
    return PL.run_segmentation(model, data, batch_size, collate)


def check_segmentation(res, g, min_agree):
    labels = res["predict_labels"]
    assert labels.shape == g["predict_labels"].shape
    agree = float((labels == g["predict_labels"]).mean())
    votes = res["predict_scores"][::int(g["stride"])].astype(np.float32)
    dv = float(np.abs(votes - g["predict_scores_strided"].astype(np.float32)).max())
    assert int(res["steps"]) == int(g["steps"]), "the sampler took a different number of steps: the patches differ"
    assert agree >= min_agree, "label agreement %.5f" % agree
    assert dv <= 2.0 ** -8, "max |d vote| %.4g" % dv                 # float16 accumulator, 0.95-smoothed over ~10 visits
    return agree, dv


def test_randlanet_semantickitti_cloud_through_the_pipeline_api_matches_the_reference_pipeline():
    g, cfg = _golden("randlanet")
    assert cfg["num_points"] == 45056
    data = dict(point=synth_data.lidar_sweep(4100), feat=None, label=None)
    data["label"] = (1 + (np.arange(data["point"].shape[0]) % 19)).astype(np.int32)
    res = native_segmentation("randlanet", cfg, data, int(g["batch_size"]), "cuda:0")
    check_segmentation(res, g, 0.9995)


def test_kpfcnn_toronto3d_cloud_through_transform_make_batch_forward_update_probs_matches_the_reference_pipeline():
    g, cfg = _golden("kpconv")
    assert cfg["in_radius"] == 4.0 and cfg["first_subsampling_dl"] == 0.08
    data = synth_data.toronto3d_tile(7, half=6.0, density=0.25)
    res = native_segmentation("kpconv", cfg, data, int(g["batch_size"]), "cuda:0")
    check_segmentation(res, g, 0.9995)


def test_kpfcnn_parislille3d_deformable_cloud_matches_the_reference_pipeline():
    """kpconv_parislille3d.yml unchanged (five ``resnetb_deformable*`` blocks, deform_radius 6.0): the native class through the
    pipeline loop against what the REAL reference pipeline + the reference's PyTorch-CPU KPFCNN produced."""
    g, cfg = _golden("kpconv_deform")
    assert sum("deformable" in b for b in cfg["architecture"]) == 5 and cfg["deform_radius"] == 6.0
    data = synth_data.toronto3d_tile(7, half=6.0, density=0.25)
    res = native_segmentation("kpconv_deform", cfg, data, int(g["batch_size"]), "cuda:0")
    check_segmentation(res, g, 0.9995)


def test_pointpillars_kitti_sweep_through_forward_and_inference_end_matches_the_reference_pipeline():
    from ml3d.torch.models import PointPillars
    g, cfg = _golden("pointpillars")
    model = PointPillars(**cfg, device="cuda:0")
    model.load_state_dict(synth_weights.pointpillars_state_dict(cfg, 33))
    data = dict(point=np.ascontiguousarray(synth_data.kitti_sweep(11), np.float32), calib=None, bounding_boxes=[])
    # preprocess / transform of the detection data path (point_pillars.py:206-267): range crop, dict for the batcher
    pre = model.preprocess(dict(data), {'split': 'test'})
    mn, mx = np.array(cfg["point_cloud_range"][:3]), np.array(cfg["point_cloud_range"][3:])
    inside = np.all((data["point"][:, :3] >= mn) & (data["point"][:, :3] < mx), axis=1)
    assert np.array_equal(pre["point"], data["point"][inside][:, :4])
    t = model.transform(pre, {'split': 'test'})
    assert set(t) == {"point", "calib"} and t["point"] is pre["point"]
    # ObjectDetection.run_inference hands the RAW dict to the batcher (object_detection.py:58-66): voxelize crops
    from ml3d.torch.dataloaders import ConcatBatcher
    batcher = ConcatBatcher("cuda:0", "PointPillars")
    boxes = PL.run_detection(model, data, "cuda:0", batcher)[0]
    got = np.array([b.to_xyzwhlr() for b in boxes], np.float32).reshape(-1, 7)
    lab = np.array([model.name2lbl[b.label_class] for b in boxes])
    assert got.shape == g["boxes"].shape and np.array_equal(lab, g["labels"])          # class-major lists: same counts per class
    assert np.abs(np.array([b.confidence for b in boxes], np.float32) - g["scores"]).max() <= 1e-4
    assert PL.box_lists_agree(got, lab, g["boxes"], g["labels"]) <= 1e-4
    # the cropped cloud gives the same detections (preprocess -> transform -> batcher path of run_test)
    boxes2 = PL.run_detection(model, t, "cuda:0", batcher)[0]
    assert len(boxes2) == len(boxes)
    got2 = np.array([b.to_xyzwhlr() for b in boxes2], np.float32).reshape(-1, 7)
    assert PL.box_lists_agree(got2, np.array([model.name2lbl[b.label_class] for b in boxes2]), got, lab) <= 1e-5
