"""GPU: the reference's data-path API on the MI355X-native model classes and the ``open3d`` shim's call shapes.

Row a2 / b2 of the coverage table: one synthetic sweep -> ``preprocess`` -> sampler loop -> ``transform`` -> ``forward``
-> vote accumulation -> labels, against the same path restated on the CPU oracle (oracle subsample / kd-tree k-NN /
PyTorch-CPU forward, numpy vote update).  Indices / sub-clouds exact, logits <= 1e-4, final labels identical."""
import numpy as np
import pytest
import torch

import synth_data
from oracle import ops as oops
from oracle import randlanet_ref as R

pytestmark = pytest.mark.gpu

# the model section of randlanet_semantickitti.yml: 45 056-point patches, 0.06 m grid
CFG = dict(num_neighbors=16, num_layers=4, num_points=45056, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
           dim_features=8, dim_output=[16, 64, 128, 256], grid_size=0.06, augment={'recenter': {'dim': [0, 1]}})


def _OracleTree(pts):
    """The reference's own search structure (randlanet.py:142): scikit-learn's KDTree.  The patch query must come back in
    ITS order (float64 distances) -- the order feeds the shuffle and the prefix subsampling."""
    from sklearn.neighbors import KDTree
    return KDTree(pts)


def _make_sampler(possibility, seed):
    """Spatially regular sampler (semseg_spatially_regular.py:62-108) with a private seeded generator, so the GPU path and
    the oracle path draw the same centres and the same shuffles."""
    rng = np.random.default_rng(seed)

    def sampler(pc, feat, label, search_tree, num_points, radius=None):
        center_id = int(np.argmin(possibility))
        center_point = pc[center_id, :].reshape(1, -1)
        idxs = search_tree.query(center_point, k=num_points)[1][0]
        idxs = rng.permutation(idxs)
        sel = pc[idxs]
        dists = np.sum(np.square((sel - center_point).astype(np.float32)), axis=1)
        possibility[idxs] += np.square(1 - dists / np.max(dists))
        return sel, idxs, center_point
    return sampler


def _oracle_preprocess(data, grid):
    pts = np.array(data['point'][:, :3], np.float32)
    lab = np.array(data['label'], np.int32).reshape(-1)
    sp, sl = oops.subsample(pts, classes=lab, sampleDl=grid)
    proj = oops.knn_search(sp, pts, 1)[:, 0].astype(np.int32)
    return {'point': sp, 'feat': None, 'label': sl.astype(np.int32), 'search_tree': _OracleTree(sp), 'proj_inds': proj}


def test_randlanet_cloud_to_labels_matches_the_oracle_path():
    from ml3d.torch.models import RandLANet
    sweep = synth_data.lidar_sweep(5)
    rng = np.random.default_rng(0)
    data = {'point': sweep, 'feat': None, 'label': rng.integers(0, 19, len(sweep)).astype(np.int32)}
    sd = R.make_state_dict(CFG, 3)
    model = RandLANet(**CFG, device="cuda:0")
    model.load_state_dict(sd)
    model.eval()
    attr = {'split': 'test'}

    # ---- preprocess: sub-cloud, labels and the raw -> sub projection are exact ------------------------------------------
    got = model.preprocess(data, attr)
    ref = _oracle_preprocess(data, CFG['grid_size'])
    assert np.array_equal(got['point'], ref['point']) and np.array_equal(got['label'], ref['label'])
    assert np.array_equal(got['proj_inds'], ref['proj_inds'])
    n = got['point'].shape[0]
    assert n > 1.5 * CFG['num_points']

    # ---- sampler loop: transform -> forward -> update_probs, 6 patches, both paths from identical sampler state -------------
    poss_g = np.random.default_rng(9).random(n) * 1e-3
    poss_o = poss_g.copy()
    model.trans_point_sampler = _make_sampler(poss_g, 123)
    samp_o = _make_sampler(poss_o, 123)
    probs_g = np.zeros((n, CFG['num_classes']), np.float16)
    probs_o = np.zeros((n, CFG['num_classes']), np.float16)
    for step in range(3):
        inp = model.transform(got, attr)
        # oracle path: same crop, recentre, pyramid on the CPU oracle
        pc, idxs, _ = samp_o(ref['point'].copy(), None, ref['label'], ref['search_tree'], CFG['num_points'])
        pc[:, [0, 1]] = pc[:, [0, 1]] - pc.mean(0)[[0, 1]]
        assert np.array_equal(inp['point_inds'], idxs)
        assert np.array_equal(inp['coords'][0], pc) and np.array_equal(inp['features'], pc)
        assert np.array_equal(inp['labels'], ref['label'][idxs].astype(np.int64))
        oin = R.build_inputs(pc[None], pc[None].copy(), CFG, oops.knn_search)
        for l in range(CFG['num_layers']):
            assert torch.equal(inp['neighbor_indices'][l].cpu().long(), oin['neighbor_indices'][l][0])
            assert torch.equal(inp['interp_idx'][l].cpu().long(), oin['interp_idx'][l][0])
            assert torch.equal(inp['sub_idx'][l].cpu().long(), oin['sub_idx'][l][0])
        batch = {k: ([t[None] for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else
                     ([torch.as_tensor(t)[None] for t in v] if isinstance(v, list) else torch.as_tensor(v)[None]))
                 for k, v in inp.items()}
        scores = model(batch)
        want = R.forward(sd, CFG, oin)
        assert (scores.cpu() - want).abs().max().item() <= 1e-4
        probs_g = model.update_probs({'data': batch}, scores, probs_g)
        p = torch.softmax(want[0], -1).numpy()
        probs_o[idxs] = 0.95 * probs_o[idxs] + (1 - 0.95) * p
    assert np.array_equal(poss_g, poss_o)
    d = np.abs(probs_g.astype(np.float32) - probs_o.astype(np.float32))
    assert d.max() <= 2 ** -9            # float16 accumulator: at most an ulp or two apart after three updates
    seen = probs_o.sum(1) > 0
    lab_g = np.argmax(probs_g, 1)[got['proj_inds']]
    lab_o = np.argmax(probs_o, 1)[ref['proj_inds']]
    agree = (lab_g == lab_o)[seen[ref['proj_inds']]].mean()
    assert agree >= 0.9999


def test_randlanet_legacy_inference_api_runs_to_completion():
    from ml3d.torch.models import RandLANet
    cfg = dict(CFG, num_points=2048, grid_size=0.5)
    sweep = synth_data.lidar_sweep(6)[::3]
    data = {'point': sweep, 'feat': None, 'label': np.zeros(len(sweep), np.int32)}
    model = RandLANet(**cfg, device="cuda:0", seed=4)
    model.load_state_dict(R.make_state_dict(cfg, 8))
    model.inference_begin(data)
    done, steps = False, 0
    while not done and steps < 400:
        inputs = model.inference_preprocess()
        done = model.inference_end(inputs, model(inputs['data']))
        steps += 1
    assert done and model.inference_result['predict_labels'].shape == (len(sweep),)
    assert model.inference_result['predict_scores'].shape == (len(sweep), cfg['num_classes'])


def test_open3d_shim_call_shapes_of_the_reference():
    """numpy / CPU-tensor calls exactly as ``dataprocessing.py:32-49,99-103`` and ``kpconv.py:2016-2032`` make them."""
    import open3d.core as o3c
    from open3d.ml.contrib import subsample, subsample_batch
    from open3d.ml.torch.layers import FixedRadiusSearch
    from open3d.ml.torch.ops import ragged_to_dense, voxelize, nms
    rng = np.random.default_rng(2)
    pts = synth_data.toronto3d_sphere(3)
    # DataProcessing.knn_search
    nns = o3c.nns.NearestNeighborSearch(o3c.Tensor.from_numpy(pts))
    nns.knn_index()
    idx, dist = nns.knn_search(o3c.Tensor.from_numpy(pts), 16)
    assert idx.numpy().dtype == np.int64 and np.array_equal(idx.numpy().astype(np.int32), oops.knn_search(pts, pts, 16))
    q = pts[:100] + 0.01
    idx, _ = nns.knn_search(o3c.Tensor.from_numpy(q), 1)
    assert np.array_equal(idx.numpy().astype(np.int32), oops.knn_search(pts, q, 1))
    # DataProcessing.grid_subsampling
    lab = rng.integers(0, 8, len(pts)).astype(np.int32)
    feat = rng.random((len(pts), 2), dtype=np.float32)
    sp, sf, sl = subsample(pts, features=feat, classes=lab, sampleDl=0.2)
    rp, rf, rl = oops.subsample(pts, features=feat, classes=lab, sampleDl=0.2)
    assert isinstance(sp, np.ndarray) and np.array_equal(sp, rp) and np.array_equal(sf, rf) and np.array_equal(sl, rl)
    assert np.array_equal(subsample(pts, sampleDl=0.2), oops.subsample(pts, sampleDl=0.2))
    lens = np.array([4000, len(pts) - 4000], np.int32)
    bp, bl = subsample_batch(pts, lens, sampleDl=0.2)
    op, ol = oops.subsample_batch(pts, lens, sampleDl=0.2)[:2]
    assert np.array_equal(bp, op) and np.array_equal(bl, ol) and bl.dtype == np.int32
    # batch_neighbors (kpconv.py:2002-2034), CPU tensors in, CPU tensors out
    q_splits = torch.LongTensor([0, 4000, len(pts)])
    res = FixedRadiusSearch()(torch.from_numpy(pts), torch.from_numpy(pts), 0.2, q_splits, q_splits)
    assert res.neighbors_index.device.type == "cpu" and res.neighbors_index.dtype == torch.int32
    idx2 = res.neighbors_index.reshape(-1, 1)
    splits = res.neighbors_row_splits
    max_nbrs = torch.max(splits[1:] - splits[:-1]).item()
    dense = ragged_to_dense(idx2, splits, max_nbrs, torch.Tensor([pts.shape[0]]).to(torch.int32)).squeeze(2).numpy()
    ref = oops.fixed_radius_search(pts, pts, 0.2, q_splits.numpy(), q_splits.numpy())
    ref_dense = oops.ragged_to_dense(ref.neighbors_index.reshape(-1, 1), ref.neighbors_row_splits, max_nbrs,
                                     np.array([pts.shape[0]], np.int32)).squeeze(2)
    assert np.array_equal(dense, ref_dense)
    # voxelize with CPU parameter tensors and a GPU cloud (point_pillars.py:317-320, 354-357)
    cloud = torch.from_numpy(synth_data.kitti_sweep(1)[:20000]).cuda()
    vs, mn, mx = torch.Tensor([0.16, 0.16, 4]), torch.Tensor([0, -39.68, -3]), torch.Tensor([69.12, 39.68, 1])
    r = voxelize(cloud[:, :3], torch.LongTensor([0, cloud.shape[0]]).cuda(), vs, mn, mx, 32, 16000)
    ro = oops.voxelize(cloud[:, :3].cpu().numpy(), np.array([0, cloud.shape[0]]), vs.numpy(), mn.numpy(), mx.numpy(), 32, 16000)
    assert r.voxel_coords.is_cuda and np.array_equal(r.voxel_coords.cpu().numpy(), ro.voxel_coords)
    assert np.array_equal(r.voxel_point_indices.cpu().numpy(), ro.voxel_point_indices)
    # nms on CPU tensors
    b = rng.random((200, 5)).astype(np.float32) * 10
    b[:, 2:4] = b[:, :2] + 1 + rng.random((200, 2)).astype(np.float32) * 3
    sc = rng.random(200).astype(np.float32)
    keep = nms(torch.from_numpy(b), torch.from_numpy(sc), 0.3)
    assert keep.device.type == "cpu" and np.array_equal(keep.numpy(), oops.nms(b, sc, 0.3))


def test_pairwise_box_iou_matches_the_oracle():
    """``iou_bev`` / ``iou_3d`` on the mAP call shapes (ml3d/metrics/mAP.py:85-88) against the oracle twin: float
    results, tolerance 1e-5 (the device's sinf / cosf differ from libm's in the last bit; north_star allows 1e-4)."""
    from open3d.ml.contrib import iou_bev_cuda, iou_3d_cpu
    rng = np.random.default_rng(4)

    def boxes(n):
        b = np.zeros((n, 7), np.float32)
        b[:, [0, 2]] = rng.uniform(-10, 10, (n, 2))
        b[:, 1] = rng.uniform(0.5, 2.0, n)
        b[:, 3:6] = rng.uniform(0.5, 4.0, (n, 3))
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
        return b
    pred, tgt = boxes(40), boxes(25)
    tgt[:5] = pred[:5]                                         # identical boxes: IoU 1
    bev = iou_bev_cuda(pred[:, [0, 2, 3, 5, 6]], tgt[:, [0, 2, 3, 5, 6]])
    assert bev.shape == (40, 25) and bev.dtype == np.float32
    assert np.abs(bev - oops.iou_bev(pred[:, [0, 2, 3, 5, 6]], tgt[:, [0, 2, 3, 5, 6]])).max() <= 1e-5
    assert np.allclose(np.diag(bev[:5, :5]), 1.0, atol=1e-5)
    i3 = iou_3d_cpu(pred, tgt)
    assert np.abs(i3 - oops.iou_3d(pred, tgt)).max() <= 1e-5
    assert np.allclose(np.diag(i3[:5, :5]), 1.0, atol=1e-5) and (i3 <= bev + 1e-6).all()
    assert iou_bev_cuda(np.zeros((0, 5), np.float32), tgt[:, [0, 2, 3, 5, 6]]).shape == (0, 25)


def test_device_resident_patch_loop_equals_the_host_loop_at_the_yaml_size():
    """randlanet_semantickitti.yml sizes (45 056-point patches of a 0.06 m sub-cloud): the loop that keeps the cloud, the
    possibilities and every patch on the device (``RandLANet._transform_device``: ml3d_nearest_to_center_dev,
    ml3d_patch_crop, ml3d_patch_recenter) against the host loop that crops and recentres in numpy -- the same patches index for
    index, bit-identical recentred coordinates (numpy's sequential float32 column mean reproduced), neighbour lists, scores,
    possibilities and votes; and nothing but the data-free shuffle crosses the bus per patch."""
    from ml3d.torch.models import RandLANet
    cfg = dict(num_neighbors=16, num_layers=4, num_points=45056, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
               dim_features=8, dim_output=[16, 64, 128, 256], grid_size=0.06, augment={"recenter": {"dim": [0, 1]}})
    sweep = synth_data.lidar_sweep(4200)
    data = dict(point=sweep, feat=None, label=np.zeros(sweep.shape[0], np.int32))
    sd = R.make_state_dict(cfg, 12)
    runs = []
    for device_loop in (False, True):
        m = RandLANet(**cfg, device="cuda:0", seed=21)
        m.load_state_dict(sd)
        m.inference_begin(dict(data))
        assert m._dev_loop is not None
        if not device_loop:
            m._dev_loop = None
        got = []
        for _ in range(4):
            inp = m.inference_preprocess()["data"]
            res = m(inp)
            m.inference_end({"data": inp}, res)
            assert isinstance(inp["point_inds"], torch.Tensor) and inp["point_inds"].is_cuda == device_loop
            got.append(dict(sel=np.asarray(inp["point_inds"][0].cpu()), pts=np.asarray(inp["coords"][0][0].cpu()),
                            nbr=np.asarray(inp["neighbor_indices"][1][0].cpu()), logits=res.cpu().numpy()))
        poss = m._dev_loop["possibility"].cpu().numpy() if device_loop else m.possibility.copy()
        runs.append((got, poss, m.test_probs.cpu().numpy()))
    (a, pa, va), (b, pb, vb) = runs
    for x, y in zip(a, b):
        assert np.array_equal(x["sel"], y["sel"]) and np.array_equal(x["pts"], y["pts"]) and np.array_equal(x["nbr"], y["nbr"])
        assert np.array_equal(x["logits"], y["logits"])
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)


def test_graphed_patch_loop_is_bit_identical_to_the_eager_loop():
    """HIP graphs around the per-patch sequence of the model-class API (``RandLANet.transform`` -> batcher -> ``forward`` ->
    ``update_probs`` at ``test_batch_size: 1``, randlanet_semantickitti.yml:38-45): the device patch loop's ~60 launches replayed
    as one captured graph into a static arena + one clone, the forward as a second graph behind one arena copy.  Six patches
    with graphs on against the same six with ``use_graphs = False``: identical indices, coordinates, neighbour lists, labels,
    scores, possibilities and votes; the returned tensors of one patch are untouched by the next patch's replay; a batch of
    TWO patches (stacked copies: no arena mark) takes the eager forward and still matches."""
    from ml3d.torch.dataloaders import DefaultBatcher
    from ml3d.torch.models import RandLANet
    cfg = dict(num_neighbors=16, num_layers=4, num_points=8192, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
               dim_features=8, dim_output=[16, 64, 128, 256], grid_size=0.06, augment={"recenter": {"dim": [0, 1]}})
    sweep = synth_data.lidar_sweep(4300)
    data = dict(point=sweep, feat=None, label=(np.arange(sweep.shape[0]) % 19).astype(np.int32))
    sd = R.make_state_dict(cfg, 14)
    collate = DefaultBatcher().collate_fn
    attr = {"split": "test"}
    runs = []
    for graphs in (True, False):
        m = RandLANet(**cfg, device="cuda:0", seed=33)
        m.load_state_dict(sd)
        m.use_graphs = graphs
        m.inference_begin(dict(data))
        got, held = [], []
        for i in range(6):
            B = 2 if i == 4 else 1
            items = [{"data": m.transform(m.inference_data, attr), "attr": attr} for _ in range(B)]
            inputs = collate(items)
            scores = m(inputs["data"])
            m.update_probs(inputs, scores, m.test_probs)
            d = inputs["data"]
            held.append((d["coords"][0], d["neighbor_indices"][0], scores, d["coords"][0].clone(), d["neighbor_indices"][0].clone(),
                         scores.clone()))
            got.append(dict(sel=d["point_inds"].cpu().numpy(), pts=d["coords"][0].cpu().numpy(), lab=d["labels"].cpu().numpy(),
                            nbr=[t.cpu().numpy() for t in d["neighbor_indices"]], itp=[t.cpu().numpy() for t in d["interp_idx"]],
                            feats=d["features"].cpu().numpy(), logits=scores.cpu().numpy()))
        torch.cuda.synchronize()
        for a, b, c, a0, b0, c0 in held:          # what a patch returned is its own: later replays did not write into it
            assert torch.equal(a, a0) and torch.equal(b, b0) and torch.equal(c, c0)
        st = m._dev_loop
        assert (st.get("graph") is not None and st.get("fwd_graph") is not None) == graphs, (st.get("graph_failed"), st.get("fwd_failed"))
        runs.append((got, st["possibility"].cpu().numpy(), m.test_probs.cpu().numpy()))
    (a, pa, va), (b, pb, vb) = runs
    for x, y in zip(a, b):
        for key in ("sel", "pts", "lab", "feats", "logits"):
            assert np.array_equal(x[key], y[key]), key
        for key in ("nbr", "itp"):
            assert all(np.array_equal(p, q) for p, q in zip(x[key], y[key])), key
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
