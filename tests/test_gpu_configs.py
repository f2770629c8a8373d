"""GPU parity on EVERY in-scope config file of the reference: the 16
ml3d/configs/{randlanet,kpconv,pointpillars}_*.yml, each loaded by the reference's own Config.load_from_file and run through
the REAL reference model on PyTorch-CPU by oracle/gen_golden_configs.py (tests/golden/configs/<yaml>.npz holds the YAML's
``model`` section, the seeds and the strided results).  Here the NATIVE class is built from that same dict and run on the
MI355X at the YAML's own sizes (num_points 40 960 .. 81 920, batch_limit 10 000 .. 50 000, 400x400 .. 640x640 canvases,
nms_pre up to 4096, max_voxels up to 60 000, two PFN layers, 5-layer / 512-wide RandLA-Net, KP_extent 1.2, 13-block KPFCNN,
deformable blocks).  Indices / voxel ids exact, floats <= 1e-4 (relative 4.4e-6 where the reference's |logit| exceeds 23: the
pseudo-trained 13-block KPFCNNs without reduce_fc reach |logit| 67 .. 187).  Every run appends its MEASURED deviations to
gpurun_out/parity_per_yaml.jsonl; the committed table is profiles/r05_parity_per_yaml.md."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import synth_data

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "configs", "*.npz")))


def tol_for(g, base=1e-4):
    """1e-4 absolute (north_star), relaxed only where the reference's own logits are large: 4.4e-6 of the largest |logit| of the
    golden -- 2x the worst MEASURED deviation (profiles/r05_parity_per_yaml.md: kpconv_semantickitti 3.81e-4 at |logit| 187 =
    2.04e-6 relative, kpconv_s3dis 1.45e-4 at 67 = 2.16e-6; every other YAML is inside the plain 1e-4 by 3x or more).  f32
    accumulation-order noise grows with the magnitude (1 ulp at 187 is 1.5e-5); until round 4 this was 6.25e-6 relative and
    another factor 2 for deformable blocks, neither of which the measurements need."""
    return max(base, 4.4e-6 * float(g["logit_scale"])) if "logit_scale" in g else base


def record(name, **kv):
    """The MEASURED figures of every YAML (max |delta| of the logits / head maps, label agreement, the margin of every flipped
    label, unmatched boxes) go to gpurun_out/parity_per_yaml.jsonl on the GPU box; tools/parity_table.py turns the file into
    profiles/rNN_parity_per_yaml.md (the tolerances below are justified by that table, not by argument)."""
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE))
    d = os.path.join(root, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_per_yaml.jsonl"), "a") as f:
        f.write(json.dumps(dict(name=name, **kv)) + "\n")


def flips(out2d, ref_argmax):
    """(agreement, [top1 - top2 margin of OUR logits at every point whose label differs from the reference's])"""
    am = out2d.argmax(-1)
    bad = np.nonzero(am.astype(np.int8).reshape(-1) != ref_argmax.reshape(-1))[0]
    flat = out2d.reshape(-1, out2d.shape[-1])
    srt = np.sort(flat[bad], axis=1)
    return float(1.0 - len(bad) / max(1, flat.shape[0])), [float(x) for x in (srt[:, -1] - srt[:, -2])[:50]]


def test_all_sixteen_in_scope_configs_have_a_golden():
    fam = [n.split("_")[0] for n in NAMES]
    assert len(NAMES) == 16 and fam.count("randlanet") == 6 and fam.count("kpconv") == 5 and fam.count("pointpillars") == 5


@pytest.mark.parametrize("name", [n for n in NAMES if n.startswith("randlanet")])
def test_randlanet_yaml(golden_dir, name):
    from oracle import randlanet_ref as R
    from ml3d import ops
    from ml3d.torch.models import RandLANet
    g = np.load(os.path.join(golden_dir, "configs", name + ".npz"))
    mcfg = json.loads(str(g["model_json"]))
    n, c = int(mcfg["num_points"]), int(mcfg["in_channels"])
    pts = synth_data.semantickitti_patch(int(g["frame_id"]), n)[None]
    assert abs(pts.astype(np.float64).sum() - float(g["points_sum"])) < 1e-6
    feats = pts.copy() if c == 3 else np.concatenate(
        [pts, np.random.default_rng(int(g["feat_seed"])).random((1, n, c - 3), dtype=np.float32)], 2)
    m = RandLANet(**mcfg, device="cuda:0")
    m.load_state_dict(R.make_state_dict(mcfg, int(g["weights_seed"])))
    m.eval()
    t = torch.from_numpy(pts).cuda()
    nbr, itp = ops.randla_knn_pyramid(t, mcfg["sub_sampling_ratio"], mcfg["num_neighbors"])
    for l in range(mcfg["num_layers"]):
        nb = nbr[l].cpu().numpy().astype(np.int64)[0]
        assert int((nb * (np.arange(nb.shape[1]) + 1)).sum()) == int(g["nbr_checksum%d" % l]), l
        up = itp[l].cpu().numpy().astype(np.int64)[0, :, 0]
        assert int((up * (np.arange(nb.shape[0]) % 7 + 1)).sum()) == int(g["interp_checksum%d" % l]), l
    out = m({"coords": [t], "features": torch.from_numpy(feats).cuda()})
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    err = float(np.abs(out[:, ::64] - g["logits_every64"]).max())
    agree, margins = flips(out, g["argmax"])
    record(name, family="randlanet", max_abs_delta=err, tol=tol_for(g), logit_scale=float(g["logit_scale"]) if "logit_scale" in g else None,
           ref_abs_max=float(np.abs(g["logits_every64"]).max()), label_agreement=agree, flipped_margins=margins, points=int(out.shape[1]))
    assert err <= tol_for(g)
    assert agree >= 0.9999


def kpconv_inputs(mcfg, g):
    """= oracle/gen_golden_configs.kpconv_spheres, from the seeds in the golden."""
    cap = int(mcfg.get("max_in_points", mcfg["batch_limit"]))
    spheres = [synth_data.toronto3d_sphere(int(g["first_frame"]) + i, cap, radius=float(mcfg["in_radius"]),
                                           grid=float(mcfg["first_subsampling_dl"])) for i in range(int(g["n_spheres"]))]
    rng = np.random.default_rng(int(g["colour_seed"]))
    cols = [np.concatenate([s, rng.random((len(s), 3), dtype=np.float32)], 1) for s in spheres]
    return spheres, cols


@pytest.mark.parametrize("name", [n for n in NAMES if n.startswith("kpconv")])
def test_kpconv_yaml(golden_dir, name):
    from oracle import kpconv_ref as K
    from ml3d.torch.dataloaders import kpconv_input_features
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    g = np.load(os.path.join(golden_dir, "configs", name + ".npz"))
    mcfg = json.loads(str(g["model_json"]))
    spheres, cols = kpconv_inputs(mcfg, g)
    pts, columns = np.concatenate(spheres), np.concatenate(cols)
    m = KPFCNN(**mcfg, device="cuda:0")
    m.load_state_dict(K.make_state_dict(mcfg, int(g["weights_seed"])))
    m.eval()
    feats = kpconv_input_features(pts, columns, mcfg["in_features_dim"]).astype(np.float32)
    assert np.allclose(feats.astype(np.float64).sum(0), g["features_sum"], rtol=1e-12)
    np.random.seed(int(g["np_seed"]))
    batch = KPConvBatch(pts, [len(s) for s in spheres], mcfg, features=feats, device="cuda:0")
    for l in range(mcfg["num_layers"]):
        assert np.array_equal(batch.lengths[l].numpy(), g["lengths"][l]), l
        assert np.array_equal(batch.points[l].cpu().numpy().astype(np.float64).sum(0), g["points_sum%d" % l]), l
        for key in ("neighbors", "pools", "upsamples"):
            mtx = getattr(batch, key)[l].cpu().numpy().astype(np.int64)
            if mtx.size == 0 and int(np.prod(g["%s_shape%d" % (key, l)])) == 0:
                continue
            assert list(mtx.shape) == list(g["%s_shape%d" % (key, l)]), (key, l)
            assert int((mtx * (np.arange(mtx.shape[1]) + 1)).sum()) == int(g["%s_checksum%d" % (key, l)]), (key, l)
    out = m(batch)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    deform = any("deformable" in b for b in mcfg["architecture"])
    err = float(np.abs(out[::8] - g["logits_every8"]).max())
    agree, margins = flips(out, g["argmax"])
    record(name, family="kpconv", max_abs_delta=err, tol=tol_for(g),
           logit_scale=float(g["logit_scale"]) if "logit_scale" in g else None, ref_abs_max=float(np.abs(g["logits_every8"]).max()),
           label_agreement=agree, flipped_margins=margins, points=int(out.shape[0]), deformable=deform)
    assert err <= tol_for(g)
    assert agree >= 0.9999            # SURVEY.md §8(c); measured 1.0 on all five YAMLs (profiles/r05_parity_per_yaml.md)


@pytest.mark.parametrize("name", [n for n in NAMES if n.startswith("pointpillars")])
def test_pointpillars_yaml(golden_dir, name):
    from oracle import pointpillars_ref as P
    from ml3d.torch.models import PointPillars
    g = np.load(os.path.join(golden_dir, "configs", name + ".npz"))
    mcfg = json.loads(str(g["model_json"]))
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(int(g["frame_id"])), mcfg)]
    assert [len(c) for c in clouds] == list(g["n_points"])
    m = PointPillars(device="cuda:0", **mcfg)
    m.load_state_dict(P.make_state_dict(mcfg, int(g["weights_seed"])))
    m.eval()
    pts = [torch.from_numpy(c).cuda() for c in clouds]
    voxels, num_points, coors = m.voxelize(pts)
    assert len(coors) == int(g["n_pillars"]) and int(num_points.sum()) == int(g["num_points_sum"])
    assert int((coors.cpu().long() * torch.tensor([1000003, 10007, 101, 1])).sum()) == int(g["coors_checksum"])
    assert np.array_equal(coors[:256].cpu().numpy().astype(np.int32), g["coors_head"])

    class In:
        point = pts
    outs = m(In())
    torch.cuda.synchronize()
    s = int(g["stride"])
    errs = {}
    for nm, t in zip(("cls", "reg", "dir"), outs):
        a = t.cpu().numpy()
        assert list(a.shape) == list(g[nm + "_shape"])
        errs[nm] = float(np.abs(a[:, :, ::s, ::s] - g[nm]).max())
        assert errs[nm] <= 1e-4, nm
        assert abs(a.astype(np.float64).sum() - float(g[nm + "_sum"])) <= 1e-5 * float(g[nm + "_abssum"]) + 1e-3
    # decode + rotated NMS at the YAML's nms_pre / score_thr / per-class thresholds.  The pseudo-trained heads saturate
    # (dozens of anchors at sigmoid = 0.9999999), so the nms_pre cut runs through float32 TIES and head maps that agree to
    # 1e-4 may keep different tied candidates: the box sets are compared as sets with a 0.4 % budget of unmatched boxes
    # (decode + NMS on IDENTICAL inputs is exact: test_gpu_pointpillars.py, test_emulated_api.py)
    boxes, scores, labels = m.bbox_head.get_bboxes(*outs)
    b, sc, lb = boxes[0].cpu().numpy(), scores[0].cpu().numpy(), labels[0].cpu().numpy()
    rb, rs, rl = g["boxes"], g["scores"], g["labels"]
    assert abs(len(b) - len(rb)) <= max(3, len(rb) // 250), (len(b), len(rb))

    def unmatched(xa, sa, la, xb, sb, lbb):
        miss = 0
        for i in range(len(xa)):
            cand = np.nonzero(lbb == la[i])[0]
            if not cand.size:
                miss += 1
                continue
            d = np.abs(xb[cand] - xa[i]).max(1) / max(1.0, np.abs(xa[i]).max())
            j = d.argmin()
            if d[j] > 1e-3 or abs(sb[cand[j]] - sa[i]) > 1e-4:
                miss += 1
        return miss
    # measured (profiles/r05_parity_per_yaml.md): 0 .. 11 unmatched of 236 .. 6948 boxes (<= 0.16 %); the budget is ~2x that
    budget = max(6, len(rb) // 250)
    u1, u2 = unmatched(rb, rs, rl, b, sc, lb), unmatched(b, sc, lb, rb, rs, rl)
    record(name, family="pointpillars", max_abs_delta=max(errs.values()), head_map_delta=errs, tol=1e-4, boxes_ref=int(len(rb)),
           boxes_gpu=int(len(b)), unmatched_ref_in_gpu=int(u1), unmatched_gpu_in_ref=int(u2), budget=int(budget))
    assert u1 <= budget and u2 <= budget
