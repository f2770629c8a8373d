"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/ref_shim.py) in this container.  Run from the repo root:  python -m oracle.gen_golden

The reference's RandLANet module (ml3d/torch/models/randlanet.py) executes its own PyTorch-CPU
forward on seeded inputs; neighbour indices come from the oracle's knn_search wired into the
reference's DataProcessing.knn_search call path (open3d is not installable).
The GPU box has no /root/reference: tests only read the .npz files written here.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ops as oops  # noqa: E402
from oracle import randlanet_ref as R  # noqa: E402
from oracle import ref_shim  # noqa: E402
import synth_data  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

KITTI_CFG = dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4],
                 in_channels=3, dim_features=8, dim_output=[16, 64, 128, 256])
SMALL_CFG = dict(num_neighbors=16, num_layers=3, num_classes=8, sub_sampling_ratio=[4, 4, 2],
                 in_channels=6, dim_features=8, dim_output=[16, 32, 64])


def reference_logits(rl, cfg, sd, pts, feats):
    """Run the reference's own transform-style neighbour loop + forward."""
    from ml3d.datasets.utils import DataProcessing  # the reference's, resolved through the shim
    B = pts.shape[0]
    model = rl.RandLANet(**cfg, num_points=pts.shape[1], ignored_label_inds=[0], grid_size=0.06)
    model.load_state_dict(sd)
    model.eval()
    model.device = torch.device("cpu")
    coords, nbrs, pools, ups = [], [], [], []
    pcs = [pts[b] for b in range(B)]
    for i in range(cfg["num_layers"]):      # mirrors randlanet.py:218-229 with the reference's own helper
        nb = [DataProcessing.knn_search(pc, pc, cfg["num_neighbors"]) for pc in pcs]
        n_sub = pcs[0].shape[0] // cfg["sub_sampling_ratio"][i]
        subs = [pc[:n_sub] for pc in pcs]
        up = [DataProcessing.knn_search(s, pc, 1) for s, pc in zip(subs, pcs)]
        coords.append(torch.from_numpy(np.stack(pcs)))
        nbrs.append(torch.from_numpy(np.stack(nb).astype(np.int64)))
        pools.append(torch.from_numpy(np.stack([x[:n_sub] for x in nb]).astype(np.int64)))
        ups.append(torch.from_numpy(np.stack(up).astype(np.int64)))
        pcs = subs
    inputs = {"coords": coords, "neighbor_indices": nbrs, "sub_idx": pools, "interp_idx": ups,
              "features": torch.from_numpy(feats)}
    with torch.no_grad():
        out = model(inputs)
    return out.numpy(), inputs


def kpconv_golden(out_path, frame_ids, max_points, weights_seed, np_seed, cfg=None):
    """KPFCNN (Toronto3D config, or ``cfg``: the small deformable architecture) through the REAL reference: KPConvBatch builds
    the batch with the oracle's radius / subsample ops wired in by the shim, KPFCNN.forward runs on PyTorch-CPU."""
    import importlib
    from oracle import kpconv_ref as K
    kp = importlib.import_module("ml3d.torch.models.kpconv")
    cb = importlib.import_module("ml3d.torch.dataloaders.concat_batcher")
    from ml3d.utils import Config
    assert os.path.abspath(kp.__file__).startswith(os.path.abspath(ref_shim.REF_ROOT))
    cfg = dict(K.TORONTO3D_CFG if cfg is None else cfg)
    model = kp.KPFCNN(**cfg)
    sd = K.make_state_dict(cfg, weights_seed)
    ref_sd = model.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), (set(ref_sd) ^ set(sd))
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    spheres = [synth_data.toronto3d_sphere(f, max_points) for f in frame_ids]
    data = dict(p_list=spheres, f_list=[s.copy() for s in spheres],
                l_list=[np.zeros(len(s), np.int32) for s in spheres], p0_list=[np.zeros(3) for _ in spheres],
                s_list=[np.ones(3, np.float32) for _ in spheres], R_list=[np.eye(3, dtype=np.float32) for _ in spheres],
                r_inds_list=[np.zeros(0) for _ in spheres], r_mask_list=[np.zeros(0) for _ in spheres],
                val_labels_list=[np.zeros(0) for _ in spheres], cfg=model.cfg)
    model.cfg.batch_limit = 10 ** 9          # keep every sphere of the list in the batch
    np.random.seed(np_seed)
    batch = cb.KPConvBatch([{"data": data}])
    with torch.no_grad():
        logits = model(batch).numpy()
    # the restatement, same np.random stream
    np.random.seed(np_seed)
    pts = np.concatenate(spheres)
    lens = [len(s) for s in spheres]
    seg = K.segmentation_inputs(pts, lens, cfg, rotations="random")
    for l in range(cfg["num_layers"]):
        assert np.array_equal(seg["points"][l], batch.points[l].numpy()), l
        assert np.array_equal(seg["neighbors"][l], batch.neighbors[l].numpy()), l
        assert np.array_equal(seg["pools"][l], batch.pools[l].numpy()), l
        assert np.array_equal(seg["upsamples"][l], batch.upsamples[l].numpy()), l
    mine = K.forward(sd, cfg, K.to_torch_batch(seg), batch.features).numpy()
    # (deformable blocks: the offsets move the kernel points, which amplifies the last-bit differences of the inner convolution)
    tol = 5e-5 if any('deformable' in b for b in cfg['architecture']) else 1e-5
    assert np.abs(mine - logits).max() <= tol, np.abs(mine - logits).max()
    g = dict(frame_ids=np.asarray(frame_ids), max_points=max_points, weights_seed=weights_seed, np_seed=np_seed,
             logits=logits, lengths=np.stack([np.asarray(x, np.int64) for x in seg["lengths"]]),
             rot0=seg["rotations"][0])
    for l in range(cfg["num_layers"]):
        nb = seg["neighbors"][l]
        g["nbr_shape%d" % l] = np.asarray(nb.shape)
        g["nbr_checksum%d" % l] = np.int64((nb * (np.arange(nb.shape[1]) + 1)).sum())
        g["points_sum%d" % l] = seg["points"][l].astype(np.float64).sum(0)
    g["nbr0_head"] = seg["neighbors"][0][:64].astype(np.int32)
    g["pool0_head"] = seg["pools"][0][:64].astype(np.int32)
    g["up0_head"] = seg["upsamples"][0][:64].astype(np.int32)
    np.savez_compressed(out_path, **g)
    print("kpconv golden:", out_path, logits.shape, [tuple(x) for x in g["lengths"]])


def pointpillars_golden(out_path, cfg_name, frame_ids, weights_seed, stride):
    """PointPillars through the REAL reference module (voxelize / ragged_to_dense come from the oracle via the
    shim); stores the three head maps subsampled by `stride` plus per-map checksums."""
    import importlib
    from oracle import pointpillars_ref as P
    pp = importlib.import_module("ml3d.torch.models.point_pillars")
    assert os.path.abspath(pp.__file__).startswith(os.path.abspath(ref_shim.REF_ROOT))
    cfg = getattr(P, cfg_name)
    model = pp.PointPillars(device="cpu", augment={}, **cfg)
    sd = P.make_state_dict(cfg, weights_seed)
    ref_sd = model.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), sorted(set(ref_sd) ^ set(sd))[:10]
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(f), cfg) for f in frame_ids]
    pts = [torch.from_numpy(c) for c in clouds]

    class _In:
        point = pts
    with torch.no_grad():
        voxels, num_points, coors = model.voxelize(pts)
        outs = model(_In())
    (mc, mr, md), aux = P.forward(sd, cfg, pts)
    assert torch.equal(aux["coors"], coors) and torch.equal(aux["num_points"], num_points)
    assert torch.equal(aux["voxels"], voxels)
    for a, b in zip(outs, (mc, mr, md)):
        assert (a - b).abs().max() <= 1e-5, (a - b).abs().max()
    # box decoding + NMS: the reference head's own get_bboxes vs the restatement (exact: same torch ops, same nms)
    rb, rs, rl = model.bbox_head.get_bboxes(*outs)
    dec = {}
    for i in range(len(clouds)):
        b, s_, l = P.get_bboxes_single(cfg, outs[0][i], outs[1][i], outs[2][i])
        assert torch.equal(l, rl[i]) and torch.equal(s_, rs[i]) and torch.allclose(b, rb[i], atol=1e-6), i
        dec["boxes%d" % i], dec["scores%d" % i], dec["labels%d" % i] = rb[i].numpy(), rs[i].numpy(), rl[i].numpy()
    g = dict(frame_ids=np.asarray(frame_ids), weights_seed=weights_seed, stride=stride, **dec,
             n_points=np.asarray([len(c) for c in clouds]), n_pillars=np.int64(len(coors)),
             coors_checksum=np.int64((coors.long() * torch.tensor([1000003, 10007, 101, 1])).sum()),
             num_points_sum=np.int64(num_points.sum()), coors_head=coors[:256].numpy().astype(np.int32))
    for name, t in zip(("cls", "reg", "dir"), outs):
        a = t.numpy()
        g[name] = a[:, :, ::stride, ::stride].copy()
        g[name + "_sum"] = a.astype(np.float64).sum()
        g[name + "_abssum"] = np.abs(a.astype(np.float64)).sum()
        g[name + "_shape"] = np.asarray(a.shape)
    np.savez_compressed(out_path, **g)
    print("pointpillars golden:", out_path, [tuple(t.shape) for t in outs], "pillars", len(coors))


def main():
    os.makedirs(OUT, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())     # the reference may write caches into the CWD
    rl = ref_shim.reference_modules()
    torch.set_num_threads(max(1, os.cpu_count() or 1))

    # 1. small heterogeneous config, batch 2, extra feature channels
    rng = np.random.default_rng(42)
    pts = synth_data.uniform_cloud(42, 2 * 1024).reshape(2, 1024, 3)
    feats = np.concatenate([pts, rng.random((2, 1024, 3), dtype=np.float32)], 2)
    sd = R.make_state_dict(SMALL_CFG, 101)
    logits, inp = reference_logits(rl, SMALL_CFG, sd, pts, feats)
    mine = R.forward(sd, SMALL_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6, "oracle restatement deviates from the reference module"
    np.savez_compressed(os.path.join(OUT, "randlanet_small.npz"), points=pts, features=feats, logits=logits,
                        weights_seed=101, nbr0=inp["neighbor_indices"][0].numpy().astype(np.int32),
                        interp0=inp["interp_idx"][0].numpy().astype(np.int32))

    # 2. SemanticKITTI widths on a 4096-point lidar-shaped patch
    pts = synth_data.semantickitti_patch(7, 4096)[None]
    sd = R.make_state_dict(KITTI_CFG, 2024)
    logits, inp = reference_logits(rl, KITTI_CFG, sd, pts, pts.copy())
    mine = R.forward(sd, KITTI_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6
    np.savez_compressed(os.path.join(OUT, "randlanet_kitti4096.npz"), points=pts, logits=logits, weights_seed=2024,
                        nbr0=inp["neighbor_indices"][0].numpy().astype(np.int32),
                        nbr3=inp["neighbor_indices"][3].numpy().astype(np.int32),
                        interp0=inp["interp_idx"][0].numpy().astype(np.int32))

    # 3. full-size frame (45056 points): logits of every 64th point + checksums
    pts = synth_data.semantickitti_patch(0, 45056)[None]
    logits, inp = reference_logits(rl, KITTI_CFG, sd, pts, pts.copy())
    mine = R.forward(sd, KITTI_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6
    nb0 = inp["neighbor_indices"][0].numpy()
    np.savez_compressed(os.path.join(OUT, "randlanet_kitti45056.npz"), frame_id=0, weights_seed=2024,
                        logits_every64=logits[:, ::64], argmax=logits.argmax(-1).astype(np.int8),
                        points_sum=pts.astype(np.float64).sum(), nbr0_rowsum=nb0.sum(-1).astype(np.int64)[0, ::16],
                        nbr0_checksum=np.int64((nb0.astype(np.int64) * (np.arange(16) + 1)).sum()))
    # 4. KPConv (Toronto3D config): one small 2-sphere batch (full logits) and one 10k-point sphere
    kpconv_golden(os.path.join(OUT, "kpconv_small.npz"), [11, 12], 1500, 303, 1234)
    kpconv_golden(os.path.join(OUT, "kpconv_toronto3d.npz"), [1], 10000, 303, 99)
    kpconv_golden(os.path.join(OUT, "kpconv_deform_small.npz"), [3, 4], 1500, 505, 77, cfg=K.KPCONV_DEFORM_SMALL_CFG)
    # 5. PointPillars: small 2-PFN-layer config, 2 samples, full maps; KITTI config, 1 sample, every 6th pixel
    pointpillars_golden(os.path.join(OUT, "pointpillars_small.npz"), "SMALL_CFG", [5, 6], 404, 1)
    pointpillars_golden(os.path.join(OUT, "pointpillars_kitti.npz"), "KITTI_CFG", [0], 404, 6)
    os.chdir(cwd)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
