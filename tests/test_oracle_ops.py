"""CPU: pin the oracle's primitive ops (oracle/ml3d_oracle.c) against independent implementations
and the only published known-answer vector for them (upstream voxelize docstring)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import ops


def _cloud(seed, n, surface=False):
    rng = np.random.default_rng(seed)
    ext = np.array([30, 30, 0.05] if surface else [10, 10, 10], np.float32)
    return rng.random((n, 3), dtype=np.float32) * ext


@pytest.mark.parametrize("n,k,surface", [(1, 1, False), (17, 16, False), (3000, 16, False), (5000, 16, True), (800, 5, False)])
def test_knn_kdtree_equals_bruteforce(n, k, surface):
    p = _cloud(n, n, surface)
    a, da = ops.knn_search(p, p, k, return_distances=True)
    b, db = ops.knn_search(p, p, k, brute=True, return_distances=True)
    assert a.shape == (n, min(k, n))
    assert np.array_equal(a, b) and np.array_equal(da, db)
    assert np.array_equal(a[:, 0], np.arange(n))          # self match first (d2 = 0)
    assert (np.diff(da, axis=1) >= 0).all()


def test_knn_matches_scipy_where_distances_are_distinct():
    p = _cloud(3, 20000)
    idx, d2 = ops.knn_search(p, p, 16, return_distances=True)
    _, ref = cKDTree(p).query(p, 16)
    assert (idx == ref).mean() > 0.9999      # float64 vs canonical float32 ordering differs only on near-ties
    assert np.allclose(d2, ((p[:, None] - p[idx]) ** 2).sum(-1), rtol=1e-5, atol=1e-7)


def test_knn_tie_break_by_index():
    base = _cloud(5, 300)
    p = np.repeat(base, 4, 0)                 # every point duplicated 4x: exact distance ties
    idx = ops.knn_search(p, p, 8)
    for i in (0, 5, 1199):
        g = (i // 4) * 4
        assert list(idx[i, :4]) == [g, g + 1, g + 2, g + 3]


def test_knn_batched_global_indices_and_padding():
    p = _cloud(9, 1000)
    splits = [0, 400, 400, 410, 1000]
    idx, d2 = ops.knn_search_batched(p, splits, p, splits, 16)
    for b in range(4):
        s, e = splits[b], splits[b + 1]
        if e - s == 0:
            continue
        kk = min(16, e - s)
        ref = ops.knn_search(p[s:e], p[s:e], 16)
        assert np.array_equal(idx[s:e, :kk], ref + s)
        assert (idx[s:e, kk:] == -1).all() and np.isinf(d2[s:e, kk:]).all()


def test_fixed_radius_search_matches_scipy_sets_and_is_sorted():
    p = _cloud(11, 4000)
    q = _cloud(12, 500)
    r = ops.fixed_radius_search(p, q, 0.7, return_distances=True)
    ref = cKDTree(p).query_ball_point(q, 0.7)
    rs = r.neighbors_row_splits
    assert rs[0] == 0 and (np.diff(rs) >= 0).all()
    bad = 0
    for i in range(len(q)):
        mine = r.neighbors_index[rs[i]:rs[i + 1]]
        d = r.neighbors_distance[rs[i]:rs[i + 1]]
        assert (np.diff(d) >= 0).all() and (d <= np.float32(0.7) * np.float32(0.7)).all()
        bad += set(mine.tolist()) != set(ref[i])
    assert bad <= 2          # only points within float rounding of the sphere may differ


def test_fixed_radius_batched():
    p = _cloud(13, 900)
    ps = [0, 300, 900]
    r = ops.fixed_radius_search(p, p, 1.0, ps, ps)
    rs = r.neighbors_row_splits
    for i in (0, 299):
        assert r.neighbors_index[rs[i]:rs[i + 1]].max() < 300
    for i in (300, 899):
        assert r.neighbors_index[rs[i]:rs[i + 1]].min() >= 300
    assert r.neighbors_index[rs[5]] == 5     # self first


def test_ragged_to_dense():
    v = np.arange(10, dtype=np.int32).reshape(-1, 1)
    out = ops.ragged_to_dense(v, [0, 3, 3, 10], 4, [-1]).squeeze(-1)
    assert out.tolist() == [[0, 1, 2, -1], [-1, -1, -1, -1], [3, 4, 5, 6]]


def test_voxelize_upstream_docstring_known_answer():
    pts = np.array([[.1, .1, .1], [.5, .5, .5], [1.7, 1.7, 1.7], [1.8, 1.8, 1.8], [9.3, 9.4, 9.4]], np.float32)
    r = ops.voxelize(pts, [0, 5], [1, 1, 1], [0, 0, 0], [2, 2, 2])
    assert r.voxel_coords.tolist() == [[0, 0, 0], [1, 1, 1]]
    assert r.voxel_point_indices.tolist() == [0, 1, 2, 3]
    assert r.voxel_point_row_splits.tolist() == [0, 2, 4]
    assert r.voxel_batch_splits.tolist() == [0, 2]


def test_voxelize_against_numpy_grouping_with_limits():
    rng = np.random.default_rng(4)
    pts = (rng.random((20000, 3), dtype=np.float32) * np.array([80, 90, 5], np.float32) - np.array([5, 45, 3.5], np.float32))
    vs, mn, mx = [0.16, 0.16, 4.0], [0, -39.68, -3], [69.12, 39.68, 1]
    r = ops.voxelize(pts, [0, 12000, 20000], vs, mn, mx, max_points_per_voxel=3, max_voxels=4000)
    assert r.voxel_batch_splits[0] == 0 and r.voxel_batch_splits[-1] == len(r.voxel_coords)
    assert (np.diff(r.voxel_batch_splits) <= 4000).all()
    cnt = np.diff(r.voxel_point_row_splits)
    assert cnt.min() >= 1 and cnt.max() <= 3
    vsf, mnf = np.asarray(vs, np.float32), np.asarray(mn, np.float32)
    for v in (0, len(cnt) // 2, len(cnt) - 1):
        ids = r.voxel_point_indices[r.voxel_point_row_splits[v]:r.voxel_point_row_splits[v + 1]]
        assert (np.diff(ids) > 0).all()
        c = ((pts[ids] - mnf) / vsf).astype(np.int32)
        assert (c == r.voxel_coords[v]).all()
    # voxels ascending by linear id inside each batch item
    G = (((np.asarray(mx, np.float32) - mnf) / vsf).astype(np.int64) + 1)
    for b in range(2):
        c = r.voxel_coords[r.voxel_batch_splits[b]:r.voxel_batch_splits[b + 1]].astype(np.int64)
        lin = c[:, 0] + G[0] * (c[:, 1] + G[1] * c[:, 2])
        assert (np.diff(lin) > 0).all()


def test_subsample_barycentres_and_labels():
    rng = np.random.default_rng(8)
    pts = rng.random((5000, 3), dtype=np.float32) * 4
    feat = rng.random((5000, 2), dtype=np.float32)
    lab = rng.integers(0, 5, 5000).astype(np.int32)
    sp, sf, sl = ops.subsample(pts, features=feat, classes=lab, sampleDl=0.5)
    org = np.floor(pts.min(0) / np.float32(0.5)) * np.float32(0.5)
    key = np.floor((pts - org) / np.float32(0.5)).astype(np.int64)
    dims = key.max(0) + 1
    lin = key[:, 0] + dims[0] * (key[:, 1] + dims[1] * key[:, 2])
    uniq, inv = np.unique(lin, return_inverse=True)
    assert sp.shape[0] == uniq.size
    ref = np.zeros((uniq.size, 3)); np.add.at(ref, inv, pts)
    ref /= np.bincount(inv)[:, None]
    assert np.allclose(sp, ref, atol=1e-5)
    reff = np.zeros((uniq.size, 2)); np.add.at(reff, inv, feat)
    assert np.allclose(sf, reff / np.bincount(inv)[:, None], atol=1e-5)
    v = 7
    assert sl[v] == np.bincount(lab[inv == v]).argmax()
    only = ops.subsample(pts, sampleDl=0.5)
    assert np.array_equal(only, sp)


def test_subsample_batch_lengths():
    pts = np.random.default_rng(2).random((3000, 3), dtype=np.float32)
    sp, lens = ops.subsample_batch(pts, [1000, 2000], sampleDl=0.25)
    assert lens.sum() == sp.shape[0] and len(lens) == 2
    assert np.array_equal(sp[:lens[0]], ops.subsample(pts[:1000], sampleDl=0.25))


def test_nms_rotated():
    boxes = np.array([[0, 0, 2, 2, 0], [0.1, 0, 2.1, 2, 0.05], [5, 5, 6, 6, 0], [0, 0, 2, 2, np.pi / 2]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    assert ops.nms(boxes, scores, 0.3).tolist() == [0, 2]
    assert ops.nms(boxes, scores, 0.99).tolist() == [0, 1, 2]   # box 3 == box 0 rotated by 90 deg: IoU 1
    assert ops.nms(boxes[:0], scores[:0], 0.5).tolist() == []


# ---- rotated IoU / NMS pinned to an INDEPENDENT float64 construction ---------------------------------------
# The oracle clips polygons (Sutherland-Hodgman, float32); here the intersection of two rotated rectangles is the polytope of
# their eight half-planes: Chebyshev centre by linear programming (scipy.optimize.linprog) -> scipy.spatial.HalfspaceIntersection
# (Qhull's dual construction) -> ConvexHull.volume (= area in 2-D).  Nothing is shared with the oracle but the box convention
# (x0, y0, x1, y1, yaw about the centre: objdet_helper.py:316-350 hands ``bev[:, [0, 1, 2, 3, 4]]`` to open3d's nms).
def _halfplanes(box):
    x0, y0, x1, y1, r = [float(v) for v in box]
    c = np.array([(x0 + x1) / 2, (y0 + y1) / 2])
    hw, hh = (x1 - x0) / 2, (y1 - y0) / 2
    u, v = np.array([np.cos(r), np.sin(r)]), np.array([-np.sin(r), np.cos(r)])
    # n . p <= n . c + half extent, for the four outward normals
    return np.array([[u[0], u[1], -(u @ c) - hw], [-u[0], -u[1], (u @ c) - hw],
                     [v[0], v[1], -(v @ c) - hh], [-v[0], -v[1], (v @ c) - hh]])


def _iou_f64(a, b):
    from scipy.optimize import linprog
    from scipy.spatial import ConvexHull, HalfspaceIntersection
    hs = np.concatenate([_halfplanes(a), _halfplanes(b)])
    # Chebyshev centre: maximise t s.t. n . p + t <= -offset (unit normals)
    res = linprog([0, 0, -1], A_ub=np.c_[hs[:, :2], np.ones(8)], b_ub=-hs[:, 2], bounds=[(None, None), (None, None), (0, None)])
    inter = 0.0
    if res.status == 0 and res.x[2] > 1e-9:
        inter = ConvexHull(HalfspaceIntersection(hs, res.x[:2]).intersections).volume
    area = lambda q: (float(q[2]) - float(q[0])) * (float(q[3]) - float(q[1]))
    return inter / (area(a) + area(b) - inter), inter


def _random_boxes(rng, n, spread):
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.4, 4.5, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)


def test_rotated_iou_matches_an_independent_float64_halfspace_construction():
    rng = np.random.default_rng(31)
    a, b = _random_boxes(rng, 1000, 2.5), _random_boxes(rng, 1000, 2.5)
    worst, overlapping = 0.0, 0
    for i in range(1000):
        got = float(ops.lib().ml3d_oracle_iou_bev(ops._p(np.ascontiguousarray(a[i])), ops._p(np.ascontiguousarray(b[i]))))
        ref, inter = _iou_f64(a[i], b[i])
        overlapping += inter > 1e-3
        worst = max(worst, abs(got - ref))
    assert overlapping > 400                       # the sample really exercises partial overlaps, not just disjoint pairs
    assert worst <= 1e-5, worst


def test_pairwise_iou_bev_and_iou_3d_match_the_float64_construction():
    rng = np.random.default_rng(32)
    n, m = 24, 30
    # mAP.py:85-88 conventions: bev boxes (x, z, w, l, yaw); 3-D boxes (x, y, z, w, h, l, yaw), y = bottom face, y axis down
    A = np.c_[rng.uniform(-4, 4, (n, 3)), rng.uniform(0.5, 4, (n, 3)), rng.uniform(-3, 3, n)].astype(np.float32)
    B = np.c_[rng.uniform(-4, 4, (m, 3)), rng.uniform(0.5, 4, (m, 3)), rng.uniform(-3, 3, m)].astype(np.float32)
    bev = ops.iou_bev(A[:, [0, 2, 3, 5, 6]], B[:, [0, 2, 3, 5, 6]])
    vol = ops.iou_3d(A, B)

    def corner(q):
        return np.array([q[0] - q[3] / 2, q[2] - q[5] / 2, q[0] + q[3] / 2, q[2] + q[5] / 2, q[6]], np.float64)
    for i in range(n):
        for j in range(m):
            iou2, inter = _iou_f64(corner(A[i]), corner(B[j]))
            assert abs(float(bev[i, j]) - iou2) <= 1e-5
            # heights: the box spans [y - h, y] (y = bottom, axis pointing down)
            top = max(float(A[i, 1]) - float(A[i, 4]), float(B[j, 1]) - float(B[j, 4]))
            bot = min(float(A[i, 1]), float(B[j, 1]))
            iv = inter * max(0.0, bot - top)
            va, vb = float(A[i, 3] * A[i, 4] * A[i, 5]), float(B[j, 3] * B[j, 4] * B[j, 5])
            assert abs(float(vol[i, j]) - iv / (va + vb - iv)) <= 1e-5


def test_nms_keep_lists_match_a_greedy_loop_over_the_float64_iou():
    rng = np.random.default_rng(33)
    for trial, (n, thr) in enumerate([(90, 0.01), (90, 0.3), (110, 0.5), (60, 0.7)]):
        boxes = _random_boxes(rng, n, 6.0)
        scores = rng.random(n).astype(np.float32)
        iou = np.zeros((n, n))
        for i in range(n):
            for j in range(i + 1, n):
                iou[i, j] = iou[j, i] = _iou_f64(boxes[i], boxes[j])[0]
        # a pair within 1e-5 of the threshold can legitimately fall either way in float32: one box of each such pair leaves
        amb = np.unique(np.nonzero(np.triu((np.abs(iou - thr) < 1e-5) & (iou > 0)))[0])
        if amb.size:
            sel = np.setdiff1d(np.arange(n), amb)
            boxes, scores, iou, n = boxes[sel], scores[sel], iou[np.ix_(sel, sel)], sel.size
        assert n >= 50
        order = np.argsort(-scores, kind="stable")
        keep, dead = [], np.zeros(n, bool)
        for i in order:
            if dead[i]:
                continue
            keep.append(int(i))
            dead |= iou[i] > thr
        assert ops.nms(boxes, scores, thr).tolist() == keep, trial


def _topk_cases():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 7000)).astype(np.float32)
    x[0, 5], x[0, 6000], x[1, 3] = np.nan, np.inf, -np.inf
    x[1, 100:200] = np.nan
    sig = (1 / (1 + np.exp(-(rng.standard_normal((2, 60000)) * 3 - 6)))).astype(np.float32)       # head-like: most scores tiny
    return [(rng.standard_normal((3, 10000)).astype(np.float32), 100), (rng.random((2, 5000), dtype=np.float32), 4096),
            (np.round(rng.random((4, 9000)) * 20).astype(np.float32) / 20, 300),                   # 21 distinct values: ties everywhere
            (np.zeros((2, 4500), np.float32), 17), (x, 150), (rng.standard_normal((1, 37)).astype(np.float32), 37),
            (rng.standard_normal((5, 4097)).astype(np.float32), 1), (sig, 100), (rng.random((0, 50), dtype=np.float32), 5),
            (np.where(rng.random((2, 3000)) < 0.5, np.float32(-0.0), np.float32(0.0)) * (rng.random((2, 3000)) < 0.9), 40),   # -0.0 == +0.0

            (rng.random((3, 50), dtype=np.float32), 0)]


def test_topk_rows_is_torch_topk_with_ties_by_ascending_index():
    """The nms_pre top-k (point_pillars.py:985-992): values == torch.topk's, indices == torch.topk's wherever a row's values are
    distinct, and the canonical tie order (ascending index) wherever they are not."""
    import torch
    for v, k in _topk_cases():
        idx = ops.topk_rows(v, k)
        assert idx.shape == (v.shape[0], k)
        tv, ti = torch.topk(torch.from_numpy(v), k, dim=1)
        got = np.take_along_axis(v, idx, 1)
        assert np.array_equal(got, tv.numpy(), equal_nan=True)
        for r in range(v.shape[0]):
            if np.unique(v[r]).size == v.shape[1] and not np.isnan(v[r]).any():
                assert np.array_equal(idx[r], ti[r].numpy())
            assert np.unique(idx[r]).size == k
            same = got[r, 1:] == got[r, :-1]
            assert (np.diff(idx[r])[same] > 0).all()                   # equal values: ascending index
            if k and k < v.shape[1] and not np.isnan(got[r, -1]):      # ties AT the k-th value: the lowest indices are the ones taken
                tied = np.flatnonzero(v[r] == got[r, -1])
                taken = idx[r][got[r] == got[r, -1]]
                assert np.array_equal(taken, tied[:taken.size])
