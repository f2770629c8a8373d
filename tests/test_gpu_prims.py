"""GPU parity (through the C ABI via ml3d.ops): fixed-radius search, ragged_to_dense, voxelize, grid
subsample vs the CPU oracle — indices / voxel ids / row_splits / barycentres bit-exact."""
import numpy as np
import pytest
import torch

import synth_data
from oracle import ops as oops

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(_dev()) if dtype is None else t.to(_dev(), dtype)


def _kpconv_sphere(seed, n=10000):
    """Toronto3D-shaped input sphere: urban surfaces, ~0.08 m grid, radius 4 m (SURVEY.md §8d frame 2)."""
    return synth_data.toronto3d_sphere(seed, n)


@pytest.mark.parametrize("kind,n,r", [("sphere", 10000, 0.2), ("sphere", 10000, 0.4), ("vol", 20000, 0.5),
                                      ("dup", 4000, 0.2), ("tiny", 5, 1.0)])
def test_radius_ragged_bit_exact(kind, n, r):
    from ml3d import ops
    rng = np.random.default_rng(n)
    if kind == "sphere":
        p = _kpconv_sphere(1, n)
    elif kind == "vol":
        p = rng.random((n, 3), dtype=np.float32) * 8
    elif kind == "dup":
        p = np.repeat(rng.random((n // 4, 3), dtype=np.float32), 4, 0)
    else:
        p = rng.random((n, 3), dtype=np.float32)
    n = len(p)
    tp = _t(p)
    res = ops.fixed_radius_search(tp, tp, r, return_distances=True)
    ref = oops.fixed_radius_search(p, p, r, return_distances=True)
    assert np.array_equal(res.neighbors_row_splits.cpu().numpy(), ref.neighbors_row_splits)
    assert np.array_equal(res.neighbors_index.cpu().numpy(), ref.neighbors_index)
    assert np.array_equal(res.neighbors_distance.cpu().numpy(), ref.neighbors_distance)


def test_radius_layer_api_batched_pool_queries():
    """FixedRadiusSearch()(supports, queries, r, s_splits, q_splits) as batch_neighbors calls it (kpconv.py:2021)."""
    from ml3d.layers import FixedRadiusSearch
    a, b = _kpconv_sphere(2, 9000), _kpconv_sphere(3, 7000)
    sup = np.concatenate([a, b])
    qa, qb = oops.subsample(a, sampleDl=0.16), oops.subsample(b, sampleDl=0.16)
    qry = np.concatenate([qa, qb])
    ss = torch.tensor([0, len(a), len(sup)], dtype=torch.int64)
    qs = torch.tensor([0, len(qa), len(qry)], dtype=torch.int64)
    res = FixedRadiusSearch()(_t(sup), _t(qry), 0.2, ss.to(_dev()), qs.to(_dev()))
    ref = oops.fixed_radius_search(sup, qry, 0.2, ss.numpy(), qs.numpy())
    assert res.neighbors_index.dtype == torch.int32 and res.neighbors_row_splits.dtype == torch.int64
    assert np.array_equal(res.neighbors_row_splits.cpu().numpy(), ref.neighbors_row_splits)
    assert np.array_equal(res.neighbors_index.cpu().numpy(), ref.neighbors_index)


def test_radius_long_rows_spill_and_workspace_growth():
    from ml3d import ops
    p = (np.random.default_rng(9).random((3000, 3), dtype=np.float32) * 0.3).astype(np.float32)
    res = ops.fixed_radius_search(_t(p), _t(p[:500]), 1.0)        # every row lists all 3000 points
    ref = oops.fixed_radius_search(p, p[:500], 1.0)
    assert np.array_equal(res.neighbors_row_splits.cpu().numpy(), ref.neighbors_row_splits)
    assert np.array_equal(res.neighbors_index.cpu().numpy(), ref.neighbors_index)


def test_batch_neighbors_dense_matches_reference_recipe():
    from ml3d import ops
    a, b = _kpconv_sphere(4, 10000), _kpconv_sphere(5, 6000)
    p = np.concatenate([a, b])
    lens = [len(a), len(b)]
    dense = ops.radius_neighbors_dense(_t(p), _t(p), lens, lens, 0.2).cpu().numpy()
    ref = oops.fixed_radius_search(p, p, 0.2, [0, len(a), len(p)], [0, len(a), len(p)])
    cols = int(np.diff(ref.neighbors_row_splits).max())
    ref_dense = oops.ragged_to_dense(ref.neighbors_index.reshape(-1, 1), ref.neighbors_row_splits, cols,
                                     np.array([len(p)], np.int32))[:, :, 0]
    assert dense.dtype == np.int32 and np.array_equal(dense, ref_dense)
    # size-independent properties: self is the first neighbour; no index crosses a batch item
    assert (dense[:, 0] == np.arange(len(p))).all()
    real = dense[:len(a)][dense[:len(a)] < len(p)]
    assert real.max() < len(a)


def test_ragged_to_dense_bit_exact():
    from ml3d import ops
    rng = np.random.default_rng(0)
    rs = np.concatenate([[0], np.cumsum(rng.integers(0, 40, 5000))])
    vals = rng.random((rs[-1], 4), dtype=np.float32)
    out = ops.ragged_to_dense(_t(vals), _t(rs), 32, torch.zeros(4))
    assert np.array_equal(out.cpu().numpy(), oops.ragged_to_dense(vals, rs, 32, np.zeros(4, np.float32)))
    iv = rng.integers(0, 10 ** 6, (rs[-1],)).astype(np.int64)
    out = ops.ragged_to_dense(_t(iv), _t(rs), 20, torch.tensor(-1))
    assert np.array_equal(out.cpu().numpy(), oops.ragged_to_dense(iv, rs, 20, np.int64(-1)))


def test_voxelize_upstream_docstring_example():
    from ml3d import ops
    pts = np.array([[.1, .1, .1], [.5, .5, .5], [1.7, 1.7, 1.7], [1.8, 1.8, 1.8], [9.3, 9.4, 9.4]], np.float32)
    r = ops.voxelize(_t(pts), torch.tensor([0, 5]), torch.tensor([1., 1, 1]), torch.tensor([0., 0, 0]),
                     torch.tensor([2., 2, 2]))
    assert r.voxel_coords.cpu().tolist() == [[0, 0, 0], [1, 1, 1]]
    assert r.voxel_point_indices.cpu().tolist() == [0, 1, 2, 3]
    assert r.voxel_point_row_splits.cpu().tolist() == [0, 2, 4] and r.voxel_batch_splits.cpu().tolist() == [0, 2]
    assert r.voxel_coords.dtype == torch.int32 and r.voxel_point_indices.dtype == torch.int64


@pytest.mark.parametrize("max_points,max_voxels", [(32, 40000), (32, 2000), (5, 2 ** 62), (2 ** 62, 2 ** 62)])
def test_voxelize_kitti_sweep_bit_exact(max_points, max_voxels):
    """pointpillars_kitti.yml: voxel 0.16 x 0.16 x 4, range [0,-39.68,-3, 69.12,39.68,1], <=32 pts, <=40000 voxels."""
    from ml3d import ops
    a, b = synth_data.kitti_sweep(0), synth_data.kitti_sweep(1)        # [~120k, 4] xyz + intensity
    pts = np.concatenate([a, b])
    rs = [0, len(a), len(pts)]
    vs, mn, mx = [0.16, 0.16, 4.0], [0, -39.68, -3], [69.12, 39.68, 1]
    tp = _t(pts)
    r = ops.voxelize(tp[:, :3], torch.tensor(rs), torch.tensor(vs), torch.tensor(mn), torch.tensor(mx),
                     max_points, max_voxels)                              # strided view, as the reference passes
    ref = oops.voxelize(pts[:, :3], rs, vs, mn, mx, max_points, max_voxels)
    assert np.array_equal(r.voxel_batch_splits.cpu().numpy(), ref.voxel_batch_splits)
    assert np.array_equal(r.voxel_coords.cpu().numpy(), ref.voxel_coords)
    assert np.array_equal(r.voxel_point_row_splits.cpu().numpy(), ref.voxel_point_row_splits)
    assert np.array_equal(r.voxel_point_indices.cpu().numpy(), ref.voxel_point_indices)
    assert len(ref.voxel_coords) > 1000


@pytest.mark.parametrize("sweeps", [16, 36])
def test_voxelize_bench_batch_in_one_call_bit_exact(sweeps):
    """The PointPillars bench's shape -- 16 sweeps (1.9 M points, ~930 tiles in 29 groups of the fused hand-off, sort.h) in ONE call,
    an empty item in the middle -- and 36 sweeps (4.3 M points: past FS_MAX_TILES, the call takes the launch chain): both identical
    to the oracle."""
    from ml3d import ops
    clouds = [synth_data.kitti_sweep(40 + (i % 6))[:, :3] for i in range(sweeps)]
    clouds.insert(sweeps // 2, np.zeros((0, 3), np.float32))
    pts = np.ascontiguousarray(np.concatenate(clouds), dtype=np.float32)
    rs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])])
    vs, mn, mx = [0.16, 0.16, 4.0], [0, -39.68, -3], [69.12, 39.68, 1]
    r = ops.voxelize(_t(pts), torch.tensor(rs), torch.tensor(vs), torch.tensor(mn), torch.tensor(mx), 32, 40000)
    ref = oops.voxelize(pts, rs, vs, mn, mx, 32, 40000)
    assert np.array_equal(r.voxel_batch_splits.cpu().numpy(), ref.voxel_batch_splits)
    assert np.array_equal(r.voxel_coords.cpu().numpy(), ref.voxel_coords)
    assert np.array_equal(r.voxel_point_row_splits.cpu().numpy(), ref.voxel_point_row_splits)
    assert np.array_equal(r.voxel_point_indices.cpu().numpy(), ref.voxel_point_indices)
    assert (len(pts) + 2047) // 2048 > (2048 if sweeps == 36 else 800)


def test_subsample_raw_sweep_006_bit_exact():
    """DataProcessing.grid_subsampling(points, grid_size=0.06) on a raw ~120k-point sweep (dataprocessing.py:14-49)."""
    from ml3d import ops
    sweep = synth_data.kitti_sweep(2)
    pts = np.ascontiguousarray(sweep[:, :3])
    feats = np.ascontiguousarray(sweep[:, 3:4])
    labs = (np.abs(pts[:, 0]) * 3).astype(np.int32) % 19
    rp, rf, rl = oops.subsample(pts, feats, labs, sampleDl=0.06)
    op, of, ol = ops.subsample(_t(pts), _t(feats), _t(labs), sampleDl=0.06)
    assert np.array_equal(op.cpu().numpy(), rp) and np.array_equal(of.cpu().numpy(), rf)
    assert np.array_equal(ol.cpu().numpy(), rl)
    assert 0.3 * len(pts) < len(rp) < len(pts)


def test_subsample_batch_kpconv_pooling_grid():
    from ml3d import ops
    a, b = _kpconv_sphere(6, 10000), _kpconv_sphere(7, 8000)
    p = np.concatenate([a, b])
    lens = [len(a), len(b)]
    rp, rl = oops.subsample_batch(p, lens, sampleDl=0.16)
    op, ol = ops.subsample_batch(_t(p), lens, sampleDl=0.16)
    assert ol.dtype == torch.int32 and np.array_equal(ol.cpu().numpy(), rl)
    assert np.array_equal(op.cpu().numpy(), rp)
    # idempotence (size-independent property): barycentres of a 0.16 grid re-sampled at 0.01 are unchanged
    op2, ol2 = ops.subsample_batch(op, ol.tolist(), sampleDl=0.01)
    assert torch.equal(ol2, ol) and torch.equal(torch.sort(op2.sum(1))[0], torch.sort(op.sum(1))[0])


def _boxes(seed, n, spread):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 2), dtype=np.float32) * spread
    wh = 0.5 + rng.random((n, 2), dtype=np.float32) * 3
    r = (rng.random(n, dtype=np.float32) * 2 - 1) * np.pi
    b = np.concatenate([c - wh / 2, c + wh / 2, r[:, None]], 1).astype(np.float32)
    s = rng.random(n, dtype=np.float32)
    if n:
        s[::7] = s[0]
    return b, s


@pytest.mark.parametrize("n,spread,thr", [(100, 12.0, 0.01), (1000, 40.0, 0.01), (4096, 90.0, 0.3), (65, 3.0, 0.5), (0, 1.0, 0.1)])
def test_rotated_nms_matches_oracle(n, spread, thr):
    """open3d.ml.torch.ops.nms as multiclass_nms calls it (objdet_helper.py:346); nms_pre is 100..4096 in the configs."""
    from ml3d import ops
    b, s = _boxes(n, n, spread)
    keep = ops.nms(_t(b), _t(s), thr)
    assert keep.dtype == torch.int64
    assert np.array_equal(keep.cpu().numpy(), oops.nms(b, s, thr))


def test_sampler_patch_query_and_projection_and_votes():
    """the 45 056-NN patch query, the 1-NN projection of raw points onto the sub-cloud and the
    float16 vote update."""
    from ml3d import ops
    sweep = synth_data.lidar_sweep(77)
    sub = oops.subsample(sweep, sampleDl=0.06)
    c = sub[4321]
    idx, d2 = ops.nearest_to_center(_t(sub), c, 45056, return_distances=True)
    # the reference's own search structure (randlanet.py:142): scikit-learn's KDTree, float64 order
    from sklearn.neighbors import KDTree
    rdist, ref = KDTree(sub).query(c.reshape(1, -1), k=45056)
    assert np.array_equal(idx.cpu().numpy(), ref[0]) and np.array_equal(np.sqrt(d2.cpu().numpy()), rdist[0])
    # proj_inds = search_tree.query(points) (randlanet.py:147-150): 1-NN of every raw point in the sub-cloud
    proj = ops.knn_search(_t(sub), _t(sweep), 1).neighbors_index[:, 0].cpu().numpy()
    assert np.array_equal(proj, oops.knn_search(sub, sweep, 1)[:, 0])
    rng = np.random.default_rng(1)
    probs = rng.random((len(sub), 19)).astype(np.float16)
    logits = (rng.standard_normal((45056, 19)) * 3).astype(np.float32)
    out = ops.vote_update(_t(probs), idx, _t(logits), 0.95).cpu().numpy()
    ref_p = probs.copy()
    ii = ref[0]
    ref_p[ii] = 0.95 * ref_p[ii] + (1 - 0.95) * torch.softmax(torch.from_numpy(logits), -1).numpy()
    d = np.abs(out.astype(np.float32) - ref_p.astype(np.float32))
    assert d.max() <= 2 ** -10 and (d == 0).mean() > 0.99


def test_argmax_labels_matches_torch_argmax():
    from ml3d import ops
    rng = np.random.default_rng(3)
    s = rng.standard_normal((4, 45056, 19)).astype(np.float32)
    s[0, :500] = np.round(s[0, :500])
    s[1, 3, 5] = np.nan
    t = _t(s)
    got = ops.argmax_labels(t)
    assert got.dtype == torch.uint8 and torch.equal(got.long(), torch.argmax(t, -1))


def test_topk_rows_is_the_nms_pre_topk_bit_exact():
    """ml3d_topk_rows (replaces ``max_scores.topk(nms_pre)``, point_pillars.py:985-992): indices == the oracle's canonical order
    (descending value, ties by ascending index), values == torch.topk's -- the small cases of the CPU suite plus a KITTI-sized
    batch (8 samples x 321 408 anchors, nms_pre 100 and 4096) with a block of equal scores across the threshold."""
    from ml3d import ops
    from test_oracle_ops import _topk_cases
    rng = np.random.default_rng(5)
    kitti = (1 / (1 + np.exp(-(rng.standard_normal((8, 321408)) * 3 - 6)))).astype(np.float32)
    kitti[1, ::7] = kitti[1, 3]
    kitti[2] = 0.25
    for v, k in _topk_cases() + [(kitti, 100), (kitti, 4096)]:
        idx, val = ops.topk_rows(_t(v), k, with_values=True)
        assert idx.dtype == torch.int64 and tuple(idx.shape) == (v.shape[0], k)
        assert np.array_equal(idx.cpu().numpy(), oops.topk_rows(v, k)), (v.shape, k)
        assert np.array_equal(val.cpu().numpy(), torch.topk(torch.from_numpy(v), k, dim=1)[0].numpy(), equal_nan=True)
    one = ops.topk_rows(_t(kitti[0]), 100)                       # the per-sample call of get_bboxes_single
    assert np.array_equal(one.cpu().numpy(), oops.topk_rows(kitti[:1], 100)[0])


def test_subsample_one_workgroup_per_item_matches_the_oracle_and_the_sorted_op():
    """``ops.batch_grid_subsampling`` on its round-6 path (``ml3d_subsample_items_*``: one 1024-thread workgroup per batch item, bitmap
    ranks + grouping + ordered float sums in LDS, two launches per call) at the KPConv bench's shape (24 input spheres, a one-point
    and an empty item among them), axis-aligned and on rotated grids, against the oracle AND the sort-based op (forced through
    ``ops.voxel._FORCE_SORTED``): barycentres and lengths bit for bit.  Then the two cues for the sort-based op: a grid of more cells
    than the kernel's bitmap (read back as stats[1] == 2, handled inside the plan) and an item of more points than its LDS takes."""
    from ml3d import ops
    from ml3d.ops import voxel as V
    from ml3d.torch.models.kpconv import random_grid_rotations
    items = [_kpconv_sphere(300 + i) for i in range(22)] + [np.zeros((0, 3), np.float32), np.array([[1.5, -2.5, 0.25]], np.float32)]
    lens = [len(x) for x in items]
    p = np.concatenate(items).astype(np.float32)
    tp = _t(p)
    np.random.seed(3)
    R = random_grid_rotations(len(items))
    for dl in (0.16, 0.32, 1.28):
        for rot in (None, R):
            tr = None if rot is None else _t(rot)
            assert V._SubsamplePlan(tp, lens, dl, tr).items
            q, ql = ops.batch_grid_subsampling(tp, lens, dl, tr)
            V._FORCE_SORTED = True
            try:
                qs, qls = ops.batch_grid_subsampling(tp, lens, dl, tr)
            finally:
                V._FORCE_SORTED = False
            assert torch.equal(q, qs) and torch.equal(ql, qls)
            if rot is None:
                rp, rl = oops.subsample_batch(p, lens, sampleDl=dl)[:2]
                assert np.array_equal(q.cpu().numpy(), rp) and np.array_equal(ql.cpu().numpy(), rl)
    # a grid with more cells than the bitmap: the plan falls back by itself and still equals the oracle
    wide = synth_data.uniform_cloud(2, 4000)
    plan = V._SubsamplePlan(_t(wide), [4000], 0.02)
    assert plan.items
    q, ql = plan.fill()
    assert not plan.items
    rp, rl = oops.subsample_batch(wide, [4000], sampleDl=0.02)[:2]
    assert np.array_equal(q.cpu().numpy(), rp) and np.array_equal(ql.cpu().numpy(), rl)
    big = synth_data.uniform_cloud(1, 20000)
    assert not V._SubsamplePlan(_t(big), [20000], 0.3).items
