"""Generate tests/golden/configs/<yaml>.npz: EVERY in-scope config file of the reference
(ml3d/configs/{randlanet,kpconv,pointpillars}_*.yml, 16 files) loaded with the reference's own
``Config.load_from_file`` (ml3d/utils/config.py:209-241), the REAL reference model built from ``cfg.model`` and run
on PyTorch-CPU on one seeded synthetic input at the YAML's own sizes.  Run from the repo root:

    python -m oracle.gen_golden_configs [name ...]

Stored per file: the YAML's ``model`` section as JSON (the GPU box has no checkout: the -m gpu test builds the native class
from exactly this dict), the seeds, strided logits / head maps, arg-max labels, index checksums.  The inputs themselves are
regenerated from the seeds by synth_data.py on the test side.  The oracle's restatement is held to the reference run here as
well (<= 1e-5), so that the oracle stays pinned on every config, not only on the three the round-1..3 goldens covered.
"""
import glob
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ops as oops  # noqa: E402
from oracle import randlanet_ref as R  # noqa: E402
from oracle import ref_shim  # noqa: E402
import synth_data  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "configs")
CFG_DIR = os.path.join(ref_shim.REF_ROOT, "ml3d", "configs")


def plain(x):
    """addict / ConfigDict tree -> plain python (what json and the native constructors take)."""
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    return x


def load_model_cfg(name):
    from ml3d.utils import Config         # the reference's, through the shim
    cfg = Config.load_from_file(os.path.join(CFG_DIR, name + ".yml"))
    return cfg, plain(cfg.model)


# ------------------------------------------------------------------------------------------------ RandLA-Net
def randla_inputs(mcfg, frame_id, feat_seed):
    """The test side regenerates exactly this (tests/test_gpu_configs.py)."""
    n = int(mcfg["num_points"])
    pts = synth_data.semantickitti_patch(frame_id, n)[None]
    c = int(mcfg["in_channels"])
    if c == 3:
        feats = pts.copy()
    else:
        rng = np.random.default_rng(feat_seed)
        feats = np.concatenate([pts, rng.random((1, n, c - 3), dtype=np.float32)], 2)
    return pts, feats


def randla_case(name, frame_id=3, weights_seed=2024, feat_seed=9):
    import importlib
    from ml3d.datasets.utils import DataProcessing
    rl = importlib.import_module("ml3d.torch.models.randlanet")
    cfg, mcfg = load_model_cfg(name)
    model = rl.RandLANet(**cfg.model)
    sd = R.make_state_dict(mcfg, weights_seed)
    ref_sd = model.state_dict()
    assert set(ref_sd) == set(sd), sorted(set(ref_sd) ^ set(sd))[:8]
    model.load_state_dict(sd)
    model.eval()
    model.device = torch.device("cpu")
    pts, feats = randla_inputs(mcfg, frame_id, feat_seed)
    # the neighbour loop of RandLANet.transform (randlanet.py:213-236) with the reference's own DataProcessing.knn_search
    coords, nbrs, pools, ups = [], [], [], []
    pc = pts[0]
    for i in range(mcfg["num_layers"]):
        nb = DataProcessing.knn_search(pc, pc, mcfg["num_neighbors"])
        n_sub = pc.shape[0] // mcfg["sub_sampling_ratio"][i]
        sub = pc[:n_sub]
        up = DataProcessing.knn_search(sub, pc, 1)
        coords.append(torch.from_numpy(pc[None]))
        nbrs.append(torch.from_numpy(nb[None].astype(np.int64)))
        pools.append(torch.from_numpy(nb[None, :n_sub].astype(np.int64)))
        ups.append(torch.from_numpy(up[None].astype(np.int64)))
        pc = sub
    inputs = {"coords": coords, "neighbor_indices": nbrs, "sub_idx": pools, "interp_idx": ups,
              "features": torch.from_numpy(feats)}
    with torch.no_grad():
        logits = model(inputs).numpy()
    mine = R.forward(sd, mcfg, inputs).numpy()
    scale = float(np.abs(logits).max())
    assert np.abs(mine - logits).max() <= 1e-5 * max(1.0, scale / 4), (np.abs(mine - logits).max(), scale)
    g = dict(model_json=json.dumps(mcfg), family="randlanet", logit_scale=scale, frame_id=frame_id, weights_seed=weights_seed, feat_seed=feat_seed,
             points_sum=pts.astype(np.float64).sum(), logits_every64=logits[:, ::64], argmax=logits.argmax(-1).astype(np.int8))
    for l in range(mcfg["num_layers"]):
        nb = nbrs[l].numpy()[0]
        g["nbr_checksum%d" % l] = np.int64((nb * (np.arange(nb.shape[1]) + 1)).sum())
        g["interp_checksum%d" % l] = np.int64((ups[l].numpy()[0, :, 0] * (np.arange(nb.shape[0]) % 7 + 1)).sum())
    return g, "logits %s" % (logits.shape,)


# ------------------------------------------------------------------------------------------------ KPConv
def kpconv_spheres(mcfg, first_frame, colour_seed):
    """Spheres of the YAML's in_radius on the YAML's first_subsampling_dl grid, capped at max_in_points each, as many whole
    spheres as fit the YAML's batch_limit (concat_batcher.py:41-60); the per-point columns [xyz | 3 colour channels] the
    dataset's transform would hand to the batcher (``f_list``)."""
    spheres, total, f = [], 0, first_frame
    cap = int(mcfg.get("max_in_points", mcfg["batch_limit"]))
    while True:
        s = synth_data.toronto3d_sphere(f, cap, radius=float(mcfg["in_radius"]), grid=float(mcfg["first_subsampling_dl"]))
        if total + len(s) > int(mcfg["batch_limit"]) and spheres:
            break
        spheres.append(s)
        total += len(s)
        f += 1
        if total >= int(mcfg["batch_limit"]) or len(spheres) >= 6:
            break
    rng = np.random.default_rng(colour_seed)
    cols = [np.concatenate([s, rng.random((len(s), 3), dtype=np.float32)], 1) for s in spheres]
    return spheres, cols


def kpconv_case(name, first_frame=21, weights_seed=303, np_seed=4321, colour_seed=17):
    import importlib
    from oracle import kpconv_ref as K
    kp = importlib.import_module("ml3d.torch.models.kpconv")
    cb = importlib.import_module("ml3d.torch.dataloaders.concat_batcher")
    cfg, mcfg = load_model_cfg(name)
    model = kp.KPFCNN(**cfg.model)
    sd = K.make_state_dict(mcfg, weights_seed)
    ref_sd = model.state_dict()
    assert set(ref_sd) == set(sd), sorted(set(ref_sd) ^ set(sd))[:8]
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    spheres, cols = kpconv_spheres(mcfg, first_frame, colour_seed)
    data = dict(p_list=spheres, f_list=cols,
                l_list=[np.zeros(len(s), np.int32) for s in spheres], p0_list=[np.zeros(3) for _ in spheres],
                s_list=[np.ones(3, np.float32) for _ in spheres], R_list=[np.eye(3, dtype=np.float32) for _ in spheres],
                r_inds_list=[np.zeros(0) for _ in spheres], r_mask_list=[np.zeros(0) for _ in spheres],
                val_labels_list=[np.zeros(0) for _ in spheres], cfg=model.cfg)
    np.random.seed(np_seed)
    batch = cb.KPConvBatch([{"data": data}])
    assert batch.points[0].shape[0] == sum(len(s) for s in spheres)        # the YAML's batch_limit kept every sphere
    with torch.no_grad():
        logits = model(batch).numpy()
    np.random.seed(np_seed)
    seg = K.segmentation_inputs(np.concatenate(spheres), [len(s) for s in spheres], mcfg, rotations="random")
    for l in range(mcfg["num_layers"]):
        assert np.array_equal(seg["points"][l], batch.points[l].numpy()), l
        assert np.array_equal(seg["neighbors"][l], batch.neighbors[l].numpy()), l
        assert np.array_equal(seg["pools"][l], batch.pools[l].numpy()), l
        assert np.array_equal(seg["upsamples"][l], batch.upsamples[l].numpy()), l
    mine = K.forward(sd, mcfg, K.to_torch_batch(seg), batch.features).numpy()
    # 13-block architectures without reduce_fc give logits of magnitude ~60 with the pseudo-trained weights: two f32 summation
    # orders differ by a few ulp OF THAT MAGNITUDE, so the gate scales with it (1e-5 at |logits| <= 4)
    scale = float(np.abs(logits).max())
    tol = (5e-5 if any("deformable" in b for b in mcfg["architecture"]) else 1e-5) * max(1.0, scale / 4)
    assert np.abs(mine - logits).max() <= tol, (np.abs(mine - logits).max(), scale)
    g = dict(model_json=json.dumps(mcfg), family="kpconv", first_frame=first_frame, n_spheres=len(spheres), weights_seed=weights_seed,
             np_seed=np_seed, colour_seed=colour_seed, logits_every8=logits[::8], logit_scale=scale, argmax=logits.argmax(1).astype(np.int8),
             features_sum=batch.features.numpy().astype(np.float64).sum(0),
             lengths=np.stack([np.asarray(x, np.int64) for x in seg["lengths"]]))
    for l in range(mcfg["num_layers"]):
        for key in ("neighbors", "pools", "upsamples"):
            m = seg[key][l].astype(np.int64)
            g["%s_shape%d" % (key, l)] = np.asarray(m.shape)
            g["%s_checksum%d" % (key, l)] = np.int64((m * (np.arange(m.shape[1]) + 1)).sum()) if m.size else np.int64(0)
        g["points_sum%d" % l] = seg["points"][l].astype(np.float64).sum(0)
    return g, "logits %s, lengths %s" % (logits.shape, [int(x.sum()) for x in g["lengths"]])


# ------------------------------------------------------------------------------------------------ PointPillars
def pointpillars_case(name, frame_id=2, weights_seed=404):
    import importlib
    from oracle import pointpillars_ref as P
    pp = importlib.import_module("ml3d.torch.models.point_pillars")
    cfg, mcfg = load_model_cfg(name)
    model = pp.PointPillars(device="cpu", **cfg.model)
    sd = P.make_state_dict(mcfg, weights_seed)
    ref_sd = model.state_dict()
    assert set(ref_sd) == set(sd), sorted(set(ref_sd) ^ set(sd))[:8]
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(frame_id), mcfg)]
    pts = [torch.from_numpy(c) for c in clouds]

    class _In:
        point = pts
    with torch.no_grad():
        voxels, num_points, coors = model.voxelize(pts)
        outs = model(_In())
        rb, rs, rl = model.bbox_head.get_bboxes(*outs)
    (mc, mr, md), aux = P.forward(sd, mcfg, pts)
    assert torch.equal(aux["coors"], coors) and torch.equal(aux["num_points"], num_points) and torch.equal(aux["voxels"], voxels)
    for a, b in zip(outs, (mc, mr, md)):
        assert (a - b).abs().max() <= 1e-5, (a - b).abs().max()
    b, s_, l = P.get_bboxes_single(mcfg, outs[0][0], outs[1][0], outs[2][0])
    assert torch.equal(l, rl[0]) and torch.equal(s_, rs[0]) and torch.allclose(b, rb[0], atol=1e-6)
    stride = 8
    g = dict(model_json=json.dumps(mcfg), family="pointpillars", frame_id=frame_id, weights_seed=weights_seed, stride=stride,
             n_points=np.asarray([len(c) for c in clouds]), n_pillars=np.int64(len(coors)),
             coors_checksum=np.int64((coors.long() * torch.tensor([1000003, 10007, 101, 1])).sum()),
             num_points_sum=np.int64(num_points.sum()), coors_head=coors[:256].numpy().astype(np.int32),
             boxes=rb[0].numpy(), scores=rs[0].numpy(), labels=rl[0].numpy())
    for nm, t in zip(("cls", "reg", "dir"), outs):
        a = t.numpy()
        g[nm] = a[:, :, ::stride, ::stride].copy()
        g[nm + "_sum"] = a.astype(np.float64).sum()
        g[nm + "_abssum"] = np.abs(a.astype(np.float64)).sum()
        g[nm + "_shape"] = np.asarray(a.shape)
    return g, "maps %s, %d pillars of %d points, %d boxes" % ([tuple(t.shape) for t in outs], len(coors), len(clouds[0]), len(rb[0]))


CASES = {"randlanet": randla_case, "kpconv": kpconv_case, "pointpillars": pointpillars_case}


def in_scope_configs():
    names = []
    for fam in CASES:
        names += sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(CFG_DIR, fam + "_*.yml")))
    return names


def main(argv):
    os.makedirs(OUT, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())     # the reference may write caches into the CWD
    ref_shim.reference_modules()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    names = argv or in_scope_configs()
    for name in names:
        t0 = time.time()
        g, note = CASES[name.split("_")[0]](name)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
        print("%-28s %5.1f s  %s" % (name, time.time() - t0, note), flush=True)
    os.chdir(cwd)


if __name__ == "__main__":
    main(sys.argv[1:])
