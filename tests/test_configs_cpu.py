"""CPU: the 16 in-scope config files of the reference (tests/golden/configs/*.npz, oracle/gen_golden_configs.py) -- every
native model class is constructible from the YAML's unchanged ``model`` section and loads a state dict with the reference's
keys; the oracle's restatement reproduces the REAL reference's stored results on one config per family (the generator holds it
to all 16 when the goldens are made).  The MI355X side of the same files: tests/test_gpu_configs.py."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import synth_data

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "configs", "*.npz")))


def _load(name):
    g = np.load(os.path.join(HERE, "golden", "configs", name + ".npz"))
    return g, json.loads(str(g["model_json"]))


def test_sixteen_configs():
    assert len(NAMES) == 16


@pytest.mark.parametrize("name", NAMES)
def test_native_class_takes_the_yaml_model_section_and_the_reference_state_dict(name):
    from ml3d.torch.models import KPFCNN, PointPillars, RandLANet
    from oracle import kpconv_ref as K, pointpillars_ref as P, randlanet_ref as R
    g, mcfg = _load(name)
    fam = str(g["family"])
    cls, make = {"randlanet": (RandLANet, R.make_state_dict), "kpconv": (KPFCNN, K.make_state_dict),
                 "pointpillars": (PointPillars, P.make_state_dict)}[fam]
    m = cls(**dict(mcfg, device="cpu"))
    sd = make(mcfg, int(g["weights_seed"]))
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert m.cfg["name"] == mcfg["name"]


def test_oracle_reproduces_the_reference_randlanet_s3dis():
    from oracle import ops as oops, randlanet_ref as R
    g, mcfg = _load("randlanet_s3dis")          # 5 layers, 512 wide, in_channels 6, 40 960 points
    n = int(mcfg["num_points"])
    pts = synth_data.semantickitti_patch(int(g["frame_id"]), n)[None]
    feats = np.concatenate([pts, np.random.default_rng(int(g["feat_seed"])).random((1, n, 3), dtype=np.float32)], 2)
    inp = R.build_inputs(pts, feats, mcfg, oops.knn_search)
    for l in range(mcfg["num_layers"]):
        nb = inp["neighbor_indices"][l].numpy()[0]
        assert int((nb * (np.arange(16) + 1)).sum()) == int(g["nbr_checksum%d" % l])
    out = R.forward(R.make_state_dict(mcfg, int(g["weights_seed"])), mcfg, inp).numpy()
    assert np.abs(out[:, ::64] - g["logits_every64"]).max() <= 1e-5 * max(1.0, float(g["logit_scale"]) / 4)
    assert np.array_equal(out.argmax(-1).astype(np.int8), g["argmax"])


def test_oracle_reproduces_the_reference_kpconv_s3dis():
    from oracle import kpconv_ref as K
    from ml3d.torch.dataloaders import kpconv_input_features
    from test_gpu_configs import kpconv_inputs
    g, mcfg = _load("kpconv_s3dis")             # 13 blocks, KP_extent 1.2, in_features_dim 5, dl 0.04
    spheres, cols = kpconv_inputs(mcfg, g)
    pts = np.concatenate(spheres)
    np.random.seed(int(g["np_seed"]))
    seg = K.segmentation_inputs(pts, [len(s) for s in spheres], mcfg, rotations="random")
    for l in range(mcfg["num_layers"]):
        m = seg["neighbors"][l].astype(np.int64)
        assert int((m * (np.arange(m.shape[1]) + 1)).sum()) == int(g["neighbors_checksum%d" % l])
    feats = torch.from_numpy(kpconv_input_features(pts, np.concatenate(cols), 5).astype(np.float32))
    out = K.forward(K.make_state_dict(mcfg, int(g["weights_seed"])), mcfg, K.to_torch_batch(seg), feats).numpy()
    assert np.abs(out[::8] - g["logits_every8"]).max() <= 1e-5 * max(1.0, float(g["logit_scale"]) / 4)


def test_oracle_reproduces_the_reference_pointpillars_argoverse():
    from oracle import pointpillars_ref as P
    g, mcfg = _load("pointpillars_argoverse")   # xyz-only points, two PFN layers, 400 x 400 canvas, nms_pre 1000
    cloud = P.crop_for_cfg(synth_data.kitti_sweep(int(g["frame_id"])), mcfg)
    (mc, mr, md), aux = P.forward(P.make_state_dict(mcfg, int(g["weights_seed"])), mcfg, [torch.from_numpy(cloud)])
    assert len(aux["coors"]) == int(g["n_pillars"])
    s = int(g["stride"])
    for nm, t in zip(("cls", "reg", "dir"), (mc, mr, md)):
        assert np.abs(t.numpy()[:, :, ::s, ::s] - g[nm]).max() <= 1e-5
    b, sc, lb = P.get_bboxes_single(mcfg, mc[0], mr[0], md[0])
    assert np.array_equal(lb.numpy(), g["labels"]) and np.abs(b.numpy() - g["boxes"]).max() <= 1e-5
