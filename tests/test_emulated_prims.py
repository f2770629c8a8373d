"""CPU: fixed-radius search, ragged_to_dense, voxelize and grid subsample — the SAME .hip sources the
GPU runs, executed through the host emulator (tests/hipemu) and compared bit for bit with the oracle."""
import os

import numpy as np
import pytest

import emu
import synth_data
from oracle import ops as oops

pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")


def _cloud(seed, n, kind="vol"):
    rng = np.random.default_rng(seed)
    if kind == "vol":
        return (rng.random((n, 3), dtype=np.float32) * 4).astype(np.float32)
    if kind == "surf":
        return (rng.random((n, 3), dtype=np.float32) * np.array([8, 8, 0.05], np.float32)).astype(np.float32)
    return np.repeat(rng.random((n // 4, 3), dtype=np.float32), 4, 0)     # duplicates: d2 ties


@pytest.mark.parametrize("kind,n,r", [("vol", 3000, 0.35), ("surf", 4000, 0.3), ("dup", 800, 0.2), ("vol", 50, 10.0)])
def test_radius_self_search_matches_oracle(kind, n, r):
    p = _cloud(n, n, kind)
    idx, rs, d2 = emu.radius(p, [0, n], p, [0, n], r, with_d2=True)
    ref = oops.fixed_radius_search(p, p, r, return_distances=True)
    assert np.array_equal(rs, ref.neighbors_row_splits)
    assert np.array_equal(idx, ref.neighbors_index)
    assert np.array_equal(d2, ref.neighbors_distance)


def test_radius_batched_queries_differ_from_supports_and_empty_item():
    p = _cloud(1, 3000)
    q = (_cloud(2, 700) * 1.3 - 0.5).astype(np.float32)       # some queries outside the support box
    ps, qs = [0, 1200, 1200, 3000], [0, 300, 450, 700]         # item 1 has queries but no supports
    idx, rs = emu.radius(p, ps, q, qs, 0.4)
    ref = oops.fixed_radius_search(p, q, 0.4, ps, qs)
    assert np.array_equal(rs, ref.neighbors_row_splits) and np.array_equal(idx, ref.neighbors_index)
    assert (rs[301:451] == rs[300]).all()


def test_radius_rows_longer_than_the_lds_stage_spill_correctly():
    p = (_cloud(3, 900) * 0.1).astype(np.float32)               # 900 points in a 0.4 box: every row ~900 long
    idx, rs = emu.radius(p, [0, 900], p[:40], [0, 40], 1.0)
    ref = oops.fixed_radius_search(p, p[:40], 1.0)
    assert (np.diff(rs) > 256).all()
    assert np.array_equal(rs, ref.neighbors_row_splits) and np.array_equal(idx, ref.neighbors_index)


def test_radius_spill_buffer_is_separate_from_the_grid_workspace():
    """the fill phase gets the count phase's workspace back untouched (sized for ZERO neighbours) and a separate spill buffer
    at an arbitrary 8-byte phase: no relocation of the grid, whatever the result size (ml3d_hip.h, ml3d_radius_fill)"""
    p = (_cloud(4, 700) * 0.1).astype(np.float32)
    ref = oops.fixed_radius_search(p, p[:30], 1.0)
    for phase in (0, 8, None):         # None: no spill buffer, ONE workspace sized for the result (and a too-small one refused)
        idx, rs = emu.radius(p, [0, 700], p[:30], [0, 30], 1.0, spill_phase=phase)
        assert np.array_equal(rs, ref.neighbors_row_splits) and np.array_equal(idx, ref.neighbors_index)


def test_radius_overflow_flag_raises_on_the_host():
    """2^31 neighbours wrap the int32 scan: the library flags it in stats[1] (radius.hip, radius_splits) and the host wrapper
    raises instead of sizing buffers from the wrapped total (checked on the host logic; 2^31 pairs do not fit a unit test)"""
    import pytest
    from ml3d.ops import _RadiusPlan
    plan = _RadiusPlan.__new__(_RadiusPlan)
    plan.total = plan.longest = None
    with pytest.raises(RuntimeError, match="2\\^31"):
        plan.resolve(values=(12345, 1 << 63))
    plan.total = plan.longest = None
    with pytest.raises(RuntimeError, match="2\\^31"):
        plan.resolve(values=(12345, -(1 << 63)))          # the same flag read back as a signed int64
    plan.total = plan.longest = None
    assert plan.resolve(values=(7, 3)).total == 7


def test_radius_dense_is_batch_neighbors_of_the_reference():
    # kpconv.py:2002-2034: ragged_to_dense(idx, splits, max_nbrs, default = Ns)
    p = _cloud(4, 2500, "surf")
    ps = [0, 1000, 2500]
    dense = emu.radius(p, ps, p, ps, 0.25, dense=True)
    ref = oops.fixed_radius_search(p, p, 0.25, ps, ps)
    cols = int(np.diff(ref.neighbors_row_splits).max())
    ref_dense = oops.ragged_to_dense(ref.neighbors_index.reshape(-1, 1), ref.neighbors_row_splits, cols,
                                     np.array([2500], np.int32))[:, :, 0]
    assert np.array_equal(dense, ref_dense)
    assert (dense[:, 0] == np.arange(2500)).all()               # the closest neighbour of a point is itself


def test_ragged_to_dense_matches_oracle():
    rng = np.random.default_rng(0)
    rs = np.concatenate([[0], np.cumsum(rng.integers(0, 9, 200))])
    vals = rng.random((rs[-1], 4), dtype=np.float32)
    out = emu.ragged_to_dense(vals, rs, 5, np.array([-1, -2, -3, -4], np.float32))
    assert np.array_equal(out, oops.ragged_to_dense(vals, rs, 5, np.array([-1, -2, -3, -4], np.float32)))
    iv = rng.integers(0, 1000, (rs[-1], 1)).astype(np.int32)
    assert np.array_equal(emu.ragged_to_dense(iv, rs, 12, np.array([777], np.int32)),
                          oops.ragged_to_dense(iv, rs, 12, np.array([777], np.int32)))


def test_voxelize_upstream_docstring_example():
    pts = np.array([[.1, .1, .1], [.5, .5, .5], [1.7, 1.7, 1.7], [1.8, 1.8, 1.8], [9.3, 9.4, 9.4]], np.float32)
    c, pi, prs, bs = emu.voxelize(pts, [0, 5], [1, 1, 1], [0, 0, 0], [2, 2, 2])
    assert c.tolist() == [[0, 0, 0], [1, 1, 1]] and pi.tolist() == [0, 1, 2, 3]
    assert prs.tolist() == [0, 2, 4] and bs.tolist() == [0, 2]


@pytest.mark.parametrize("max_points,max_voxels", [(2**62, 2**62), (5, 2**62), (32, 150), (1, 7)])
def test_voxelize_matches_oracle_batched_with_limits(max_points, max_voxels):
    rng = np.random.default_rng(11)
    pts = np.concatenate([rng.random((4000, 3), dtype=np.float32) * [80, 90, 5] + [-5, -45, -3.5],
                          rng.random((4000, 1), dtype=np.float32)], 1).astype(np.float32)   # [N,4]: stride 4
    rs = [0, 1500, 1500, 4000]
    vs, mn, mx = [0.16 * 8, 0.16 * 8, 4], [0, -39.68, -3], [69.12, 39.68, 1]
    c, pi, prs, bs = emu.voxelize(pts, rs, vs, mn, mx, max_points, max_voxels)
    ref = oops.voxelize(pts[:, :3], rs, vs, mn, mx, max_points, max_voxels)
    assert np.array_equal(bs, ref.voxel_batch_splits)
    assert np.array_equal(c, ref.voxel_coords)
    assert np.array_equal(prs, ref.voxel_point_row_splits)
    assert np.array_equal(pi, ref.voxel_point_indices)


def test_voxelize_points_on_the_upper_range_bound_are_kept():
    pts = np.array([[2.0, 2.0, 2.0], [0.0, 0.0, 0.0], [2.0000002, 1, 1], [-1e-7, 1, 1]], np.float32)
    c, pi, prs, bs = emu.voxelize(pts, [0, 4], [0.5, 0.5, 0.5], [0, 0, 0], [2, 2, 2])
    ref = oops.voxelize(pts, [0, 4], [0.5, 0.5, 0.5], [0, 0, 0], [2, 2, 2])
    assert np.array_equal(c, ref.voxel_coords) and np.array_equal(pi, ref.voxel_point_indices)
    assert c.tolist() == [[0, 0, 0], [4, 4, 4]]


def test_voxelize_empty_input():
    c, pi, prs, bs = emu.voxelize(np.zeros((0, 3), np.float32), [0, 0], [1, 1, 1], [0, 0, 0], [2, 2, 2])
    assert c.shape == (0, 3) and pi.shape == (0,) and prs.tolist() == [0] and bs.tolist() == [0, 0]


def test_subsample_matches_oracle_points_features_labels():
    rng = np.random.default_rng(5)
    pts = synth_data.semantickitti_patch(3, 6000)
    feats = rng.random((6000, 3), dtype=np.float32)
    labs = rng.integers(0, 6, 6000).astype(np.int32)
    lens = [2500, 0, 3500]
    op, ln, of, ol = emu.subsample_batch(pts, lens, 0.4, feats, labs)
    rp, rl, rf, rlab = oops.subsample_batch(pts, lens, feats, labs, sampleDl=0.4)
    assert np.array_equal(ln, rl)
    assert np.array_equal(op, rp) and np.array_equal(of, rf) and np.array_equal(ol, rlab)
    assert 0 < len(op) < 6000


def test_subsample_negative_coordinates_and_single_points():
    pts = np.array([[-0.31, -0.29, 0.0], [-0.3, -0.3, 0.01], [5.0, 5.0, 5.0], [-7.0, 2.0, 1.0]], np.float32)
    op, ln, _, _ = emu.subsample_batch(pts, [4], 0.1)
    ref = oops.subsample(pts, sampleDl=0.1)
    assert np.array_equal(op, ref) and ln.tolist() == [len(ref)]


def _boxes(seed, n, spread):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 2), dtype=np.float32) * spread
    wh = 0.5 + rng.random((n, 2), dtype=np.float32) * 3
    r = (rng.random(n, dtype=np.float32) * 2 - 1) * np.pi
    b = np.concatenate([c - wh / 2, c + wh / 2, r[:, None]], 1).astype(np.float32)
    s = rng.random(n, dtype=np.float32)
    if n:
        s[::7] = s[0]                               # score ties: lower index first
    return b, s


@pytest.mark.parametrize("n,spread,thr", [(100, 12.0, 0.01), (300, 20.0, 0.5), (70, 4.0, 0.01), (1, 1.0, 0.1), (0, 1.0, 0.1)])
def test_rotated_nms_matches_oracle(n, spread, thr):
    b, s = _boxes(n, n, spread)
    assert np.array_equal(emu.nms(b, s, thr), oops.nms(b, s, thr))


def test_nearest_to_center_is_the_sampler_query_in_sklearn_order():
    """PINNED against the reference's actual dependency: semseg_spatially_regular.py:90-91 calls ``KDTree.query(centre, k)`` of
    scikit-learn (randlanet.py:142).  Same indices in the same order, same float64 distances -- the order feeds
    random.shuffle and the prefix subsampling of RandLANet.transform."""
    from sklearn.neighbors import KDTree
    pts = synth_data.semantickitti_patch(9, 5000)
    for c in (pts[123], pts[7] + np.float32(0.013), np.array([30.5, -2.25, 1.0], np.float32)):
        idx, d2 = emu.nearest_to_center(pts, c, 2048)
        dist, ref = KDTree(pts).query(c.reshape(1, -1), k=2048)
        assert np.array_equal(idx, ref[0])
        assert np.array_equal(np.sqrt(d2), dist[0])
    idx_all, _ = emu.nearest_to_center(pts, pts[123], 5000)
    assert idx_all[0] == 123 and np.array_equal(np.sort(idx_all), np.arange(5000))
    # exact float64 ties (duplicated points) come back in ascending index order
    dup = np.concatenate([pts[:50], pts[:50]])
    idx, d2 = emu.nearest_to_center(dup, dup[3], 100)
    assert all(idx[i] < idx[i + 1] for i in range(0, 100, 2)) and np.array_equal(d2[0::2], d2[1::2])


def test_nearest_to_center_on_a_cloud_longer_than_the_short_sort():
    """The radix sort under it has two forms: <= 128 blocks of 1024 keys every scatter block derives its offsets from the raw
    block histograms, longer arrays go through the histogram scan.  140 000 points = 137 blocks; numpy's stable argsort of the
    same float64 keys is the reference (duplicates included: ties in ascending index)."""
    rng = np.random.default_rng(12)
    pts = np.concatenate([synth_data.semantickitti_patch(3, 70000), synth_data.semantickitti_patch(3, 70000)])
    pts[70000:] += rng.integers(0, 2, (70000, 1)).astype(np.float32) * np.float32(0.25)       # half of them exact duplicates
    c = pts[17] + np.float32(0.5)
    idx, d2 = emu.nearest_to_center(pts, c, 4096)
    d = pts.astype(np.float64) - c.astype(np.float64)
    key = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    ref = np.argsort(key, kind="stable")[:4096]
    assert np.array_equal(idx, ref) and np.array_equal(d2, key[ref])


def test_vote_update_follows_numpy_float16_promotion():
    """randlanet.py:457-462: test_probs (float16) = 0.95 * test_probs + 0.05 * softmax(logits) (float32)."""
    import torch
    rng = np.random.default_rng(4)
    probs = rng.random((300, 19)).astype(np.float16)
    inds = rng.permutation(300)[:120].astype(np.int32)
    logits = (rng.standard_normal((120, 19)) * 3).astype(np.float32)
    out = emu.vote_update(probs, inds, logits, 0.95)
    ref = probs.copy()
    p = torch.softmax(torch.from_numpy(logits), -1).numpy()
    ref[inds] = 0.95 * ref[inds] + (1 - 0.95) * p
    assert ref.dtype == np.float16
    d = np.abs(out.astype(np.float32) - ref.astype(np.float32))
    assert d.max() <= 2 ** -10 and (d == 0).mean() > 0.99          # at most one float16 ulp, almost always none
    untouched = np.setdiff1d(np.arange(300), inds)
    assert np.array_equal(out[untouched], probs[untouched])


# ---- edge cases (empty / degenerate inputs) ---------------------------------------------------------------------------
def test_radius_empty_queries_zero_radius_and_coincident_points():
    p = _cloud(8, 300).astype(np.float32)
    idx, rs = emu.radius(p, [0, 300], np.zeros((0, 3), np.float32), [0, 0], 0.5)
    assert idx.shape == (0,) and rs.tolist() == [0]
    # radius 0: a point finds exactly the points at its own position (d2 == 0 <= 0), duplicates in index order
    q = np.concatenate([p[:50], p[:50]])                      # every position twice
    idx, rs = emu.radius(q, [0, 100], q, [0, 100], 0.0)
    ref = oops.fixed_radius_search(q, q, 0.0)
    assert np.array_equal(rs, ref.neighbors_row_splits) and np.array_equal(idx, ref.neighbors_index)
    assert (np.diff(rs) == 2).all() and (idx.reshape(100, 2)[:, 0] < idx.reshape(100, 2)[:, 1]).all()


def test_radius_dense_with_no_neighbours_at_all_has_zero_columns():
    p = _cloud(9, 40).astype(np.float32)
    q = (p + 100.0).astype(np.float32)
    dense = emu.radius(p, [0, 40], q, [0, 40], 0.1, dense=True)
    assert dense.shape == (40, 0)


def test_ragged_to_dense_empty_rows_and_zero_columns():
    vals = np.arange(5, dtype=np.float32)
    rs = np.array([0, 0, 2, 2, 5], np.int64)
    out = emu.ragged_to_dense(vals, rs, 3, np.float32(-1))
    assert out.tolist() == [[-1, -1, -1], [0, 1, -1], [-1, -1, -1], [2, 3, 4]]
    assert emu.ragged_to_dense(vals, rs, 0, np.float32(-1)).shape == (4, 0)


def test_subsample_empty_input_and_all_points_in_one_voxel():
    out = emu.subsample_batch(np.zeros((0, 3), np.float32), [0], 0.1)
    assert out[0].shape == (0, 3) and np.asarray(out[1]).tolist() == [0]
    p = (np.random.default_rng(3).random((77, 3)) * 0.01 + 5.0).astype(np.float32)
    out = emu.subsample_batch(p, [77], 1.0)
    ref = oops.subsample_batch(p, [77], sampleDl=1.0)
    assert out[0].shape == (1, 3) and np.array_equal(out[0], ref[0]) and np.asarray(out[1]).tolist() == [1]


def test_nms_and_iou_survive_inf_nan_and_duplicate_boxes():
    """inf / NaN corners can alternate the Sutherland-Hodgman sign test so that a clipped list outgrows its 16 LDS
    slots (4 -> 6 -> 9 -> 13 -> 19).  Such pairs now count as area 0 in the kernel AND in the oracle: the keep list still
    equals the oracle's, the IoU matrix of the finite boxes is untouched, nothing is written outside a lane's own slots."""
    rng = np.random.default_rng(12)
    n = 96
    c = rng.random((n, 2), dtype=np.float32) * 6
    wh = 0.5 + rng.random((n, 2), dtype=np.float32) * 3
    b = np.concatenate([c - wh / 2, c + wh / 2, (rng.random((n, 1), dtype=np.float32) * 2 - 1) * np.pi], 1).astype(np.float32)
    b[5] = b[4]; b[17] = b[4]                                   # duplicates
    b[9, 0] = np.inf; b[21, 2] = -np.inf; b[33, 4] = np.nan     # degenerate corners / angle
    b[40] = [np.nan, 0, np.inf, 1, 0.3]; b[41] = [0, np.nan, 1, np.inf, np.inf]
    b[50, :4] = [1e30, -1e30, 3e38, 3e38]
    s = rng.random(n, dtype=np.float32)
    for thr in (0.01, 0.3, 0.7):
        assert np.array_equal(emu.nms(b, s, thr), oops.nms(b, s, thr))
    # pairwise IoU (centre / size form): rows and columns of finite boxes equal a run without the degenerate ones
    ctr = np.concatenate([c, wh, b[:, 4:5]], 1).astype(np.float32)
    bad = np.array([9, 21, 33, 40, 41, 50])
    ctr[bad] = b[bad]
    full = emu.box_iou(ctr, ctr, False)
    good = np.setdiff1d(np.arange(n), bad)
    assert np.array_equal(full[np.ix_(good, good)], emu.box_iou(ctr[good], ctr[good], False))


def test_nms_no_boxes_one_box_and_identical_boxes():
    assert emu.nms(np.zeros((0, 5), np.float32), np.zeros(0, np.float32), 0.5).tolist() == []
    b = np.array([[0, 0, 2, 1, 0.3]], np.float32)
    assert emu.nms(b, np.array([0.7], np.float32), 0.5).tolist() == [0]
    b3 = np.repeat(b, 3, 0)
    keep = emu.nms(b3, np.array([0.1, 0.9, 0.5], np.float32), 0.5)
    assert keep.tolist() == oops.nms(b3, np.array([0.1, 0.9, 0.5], np.float32), 0.5).tolist() == [1]
    # equal scores, among them -0.0 and +0.0 (ONE value): disjoint boxes all survive, in index order
    far = np.array([[10 * i, 0, 10 * i + 2, 1, 0.1 * i] for i in range(6)], np.float32)
    sc = np.array([0.0, -0.0, 0.0, -0.0, 0.5, 0.5], np.float32)
    assert emu.nms(far, sc, 0.5).tolist() == oops.nms(far, sc, 0.5).tolist() == [4, 5, 0, 1, 2, 3]


def test_argmax_labels_is_torch_argmax_first_maximum_and_nan():
    import torch
    rng = np.random.default_rng(3)
    s = rng.standard_normal((7, 301, 19)).astype(np.float32)
    s[0, :50] = np.round(s[0, :50])                     # ties: the first maximum wins
    s[1, 3, 5] = np.nan
    s[1, 4, 0] = np.nan
    s[2, 9, :] = -np.inf
    assert np.array_equal(emu.argmax_labels(s), torch.argmax(torch.from_numpy(s), -1).numpy().astype(np.uint8))
    assert np.array_equal(emu.argmax_labels(s[:, :, :1]), np.zeros((7, 301), np.uint8))


def test_pyramid_search_is_exact_on_degenerate_clouds():
    """The pyramid's one-launch search (16-NN + the prefix 1-NN riding along) against the oracle on clouds built to break
    a grid search: 400 coincident points (one cell holds them all), a far cluster plus isolated outliers that need many shells,
    a thin column (one cell wide in x / y, hundreds of rows in z), down to a level of 32 points."""
    rng = np.random.default_rng(5)
    N = 2048
    a = rng.random((N, 3), dtype=np.float32) * np.float32([8, 8, 1])
    b = a.copy(); b[:400] = b[0]
    c = a.copy(); c[:64] += np.float32([40, 40, 9])
    c[64:80] = rng.random((16, 3), dtype=np.float32) * 200
    d = (rng.random((N, 3), dtype=np.float32) * np.float32([0.5, 0.5, 20])).astype(np.float32)
    pts = np.stack([a, b, c, d])
    nbr, itp = emu.pyramid(pts, [4, 4, 4, 2], 16)
    lv = pts
    for l, ratio in enumerate([4, 4, 4, 2]):
        n_sub = lv.shape[1] // ratio
        for i in range(pts.shape[0]):
            assert np.array_equal(nbr[l][i], oops.knn_search(lv[i], lv[i], 16)), (l, i)
            assert np.array_equal(itp[l][i], oops.knn_search(lv[i][:n_sub], lv[i], 1)), (l, i)
        lv = lv[:, :n_sub]


@pytest.mark.parametrize("k", [1, 63, 64, 100, 2560, 2561, 5200, 7777])
def test_patch_recenter_means_are_numpys_sequential_float32_sums(k):
    """ml3d_patch_recenter subtracts the column means numpy computes for a float32 [k, 3] array -- a row-by-row sequential sum
    (randlanet.py:185-191 through Augmentation.recenter) -- bit for bit: stage boundaries of the two-stage LDS pipeline
    (2 560 rows), the 64-row unrolled body and its tails."""
    L = emu.lib()
    rng = np.random.default_rng(k)
    pts = ((rng.random((k, 3)) - 0.3) * np.float32([60, 40, 5])).astype(np.float32)
    extra = rng.random((k, 2)).astype(np.float32) * 255
    out = pts.copy()
    feats = np.zeros((k, 5), np.float32)
    scratch = np.zeros(64, np.uint8)
    rc = L.ml3d_patch_recenter(out.ctypes.data, k, 3, extra.ctypes.data, 2, 0.0, 255.0, feats.ctypes.data, scratch.ctypes.data, 64, None)
    assert rc == 0
    ref = pts.copy()
    ref[:, [0, 1]] = ref[:, [0, 1]] - ref.mean(0)[[0, 1]]
    assert np.array_equal(out, ref)
    assert np.array_equal(feats[:, :3], ref) and np.array_equal(feats[:, 3:], (extra - np.float32(0.0)) / np.float32(255.0))


def test_topk_rows_matches_the_oracle_and_torch_topk():
    """ml3d_topk_rows (the nms_pre top-k, point_pillars.py:985-992): indices identical to the oracle's canonical order, values
    identical to torch.topk's -- random rows, heavy ties, all-equal rows, NaN / inf, k = n, k = 1, a KITTI-sized row."""
    import torch
    from test_oracle_ops import _topk_cases
    rng = np.random.default_rng(5)
    kitti = (1 / (1 + np.exp(-(rng.standard_normal((2, 321408)) * 3 - 6)))).astype(np.float32)
    kitti[1, ::7] = kitti[1, 3]                          # 45 916 copies of one value straddling the threshold region
    for v, k in _topk_cases() + [(kitti, 100), (kitti, 4096)]:
        idx, val = emu.topk_rows(v, k)
        assert np.array_equal(idx, oops.topk_rows(v, k)), (v.shape, k)
        assert np.array_equal(val, torch.topk(torch.from_numpy(v), k, dim=1)[0].numpy(), equal_nan=True)


def test_topk_rows_argument_checks():
    L = emu.lib()
    v = np.zeros((2, 10), np.float32)
    idx = np.zeros((2, 10), np.int64)
    ws = np.zeros(1 << 20, np.uint8)
    assert L.ml3d_topk_rows(v.ctypes.data, 2, 10, 11, idx.ctypes.data, None, ws.ctypes.data, ws.size, None) == -1       # k > n
    assert L.ml3d_topk_rows(v.ctypes.data, 2, 10, 5, idx.ctypes.data, None, ws.ctypes.data, 8, None) == -2            # workspace
    assert L.ml3d_topk_rows(None, 2, 10, 5, idx.ctypes.data, None, ws.ctypes.data, ws.size, None) == -1
    assert L.ml3d_topk_rows(v.ctypes.data, 0, 10, 5, idx.ctypes.data, None, ws.ctypes.data, ws.size, None) == 0
    big = np.zeros((1, 5000), np.float32)
    assert L.ml3d_topk_rows(big.ctypes.data, 1, 5000, 4097, idx.ctypes.data, None, ws.ctypes.data, ws.size, None) == -4   # k > 4096
    assert L.ml3d_topk_rows_workspace_bytes(1, 5000, 4097) == 0


def test_pairwise_box_iou_matches_the_oracle_twin():
    """ml3d_iou_bev / ml3d_iou_3d (ml3d/metrics/mAP.py:85-88) through the emulator against the oracle twin on the mAP call shapes
    (40 x 25 boxes, identical pairs, disjoint pairs): the same float32 operation order, the same libm -> equal to the last bit
    (the MI355X test allows 1e-5 for the device's sinf / cosf); 1000 pairs cross more than one 64-pair workgroup."""
    rng = np.random.default_rng(4)

    def boxes(n):
        b = np.zeros((n, 7), np.float32)
        b[:, [0, 2]] = rng.uniform(-10, 10, (n, 2))
        b[:, 1] = rng.uniform(0.5, 2.0, n)
        b[:, 3:6] = rng.uniform(0.5, 4.0, (n, 3))
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
        return b
    pred, tgt = boxes(40), boxes(25)
    tgt[:5] = pred[:5]
    cols = [0, 2, 3, 5, 6]
    bev = emu.box_iou(pred[:, cols], tgt[:, cols], False)
    assert np.array_equal(bev, oops.iou_bev(pred[:, cols], tgt[:, cols]))
    assert np.allclose(np.diag(bev[:5, :5]), 1.0, atol=1e-5) and (bev == 0).any()
    i3 = emu.box_iou(pred, tgt, True)
    assert np.array_equal(i3, oops.iou_3d(pred, tgt)) and (i3 <= bev + 1e-6).all()
    assert emu.box_iou(np.zeros((0, 5), np.float32), tgt[:, cols], False).shape == (0, 25)


def test_topk_rows_property_random_shapes_and_tie_patterns():
    """ml3d_topk_rows against the oracle on random (rows, n, k) and value patterns -- continuous, few distinct values, constant,
    with NaN / inf sprinkled in; n around the 4096-element workgroup chunk and the 16-element thread slice, k up to 4096."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None, derandomize=True)
    @given(st.integers(1, 3), st.sampled_from([1, 2, 15, 16, 17, 255, 4095, 4096, 4097, 8191, 9000, 20000]),
           st.integers(0, 2 ** 31 - 1), st.sampled_from(["normal", "few", "const", "special"]), st.floats(0.0, 1.0))
    def check(rows, n, seed, kind, kfrac):
        rng = np.random.default_rng(seed)
        if kind == "normal":
            v = rng.standard_normal((rows, n)).astype(np.float32)
        elif kind == "few":
            v = rng.integers(0, 4, (rows, n)).astype(np.float32) * 0.25
        elif kind == "const":
            v = np.full((rows, n), rng.standard_normal(), np.float32)
        else:
            v = rng.standard_normal((rows, n)).astype(np.float32)
            m = rng.random((rows, n))
            v[m < 0.02] = np.nan
            v[(m >= 0.02) & (m < 0.04)] = np.inf
            v[(m >= 0.04) & (m < 0.06)] = -np.inf
            v[(m >= 0.06) & (m < 0.10)] = -0.0
            v[(m >= 0.10) & (m < 0.14)] = 0.0
        k = min(n, 4096, max(0, int(round(kfrac * min(n, 4096)))))
        idx, val = emu.topk_rows(v, k)
        assert np.array_equal(idx, oops.topk_rows(v, k)), (rows, n, k, kind, seed)
        assert np.array_equal(val, np.take_along_axis(v, idx, 1), equal_nan=True)
    check()


def test_subsample_one_workgroup_per_item_is_bit_identical_to_the_sorted_op_and_the_oracle():
    """ml3d_subsample_items_* (round 6: bounding box, occupancy bitmap, popcount ranks, grouping and the ordered float sums of an
    item in the LDS of ONE workgroup; what the KPConv batch build calls): barycentres and lengths identical to the sort-based op
    and to the oracle -- spheres of a KPConv batch (incl. an empty item and a one-point item), runs longer than the in-thread sort
    takes (duplicated points: > 48 per voxel, ordered by the whole workgroup), negative coordinates."""
    rng = np.random.default_rng(8)
    spheres = [synth_data.toronto3d_sphere(60 + i, 1500 + 700 * i) for i in range(3)]
    dup = np.repeat(spheres[0][:40], 70, 0)[rng.permutation(2800)] + rng.normal(0, 1e-3, (2800, 3)).astype(np.float32)   # 70 points per voxel
    one = np.array([[-3.25, 7.5, -0.125]], np.float32)
    items = [spheres[0], np.zeros((0, 3), np.float32), spheres[1], dup.astype(np.float32), one, spheres[2] - 20.0]
    pts = np.concatenate(items).astype(np.float32)
    lens = [len(x) for x in items]
    for dl in (0.16, 0.4, 2.0):
        op, ln, st = emu.subsample_items(pts, lens, dl)
        assert st == 0
        sp, sl, _, _ = emu.subsample_batch(pts, lens, dl)
        rp, rl = oops.subsample_batch(pts, lens, sampleDl=dl)[:2]
        assert np.array_equal(ln, sl) and np.array_equal(ln, rl)
        assert np.array_equal(op, sp) and np.array_equal(op, rp)
    # the two smaller size classes (items of <= 4096 / <= 1024 points: 512 / 256 threads), long runs included
    for cap_n in (3000, 900):
        its = [synth_data.toronto3d_sphere(80 + i, cap_n - 200 * i) for i in range(3)] + [dup[:cap_n - 100].astype(np.float32)]
        pp = np.concatenate(its).astype(np.float32)
        ll = [len(x) for x in its]
        assert max(ll) <= cap_n
        for dl in (0.3, 0.7):
            op, ln, st = emu.subsample_items(pp, ll, dl)
            rp, rl = oops.subsample_batch(pp, ll, sampleDl=dl)[:2]
            assert st == 0 and np.array_equal(ln, rl) and np.array_equal(op, rp), (cap_n, dl)
    # everything in ONE voxel (a run of the item's size) and every point its own voxel
    p = (rng.random((3000, 3)) * 0.01 + 5.0).astype(np.float32)
    op, ln, st = emu.subsample_items(p, [3000], 1.0)
    ref = oops.subsample_batch(p, [3000], sampleDl=1.0)
    assert st == 0 and op.shape == (1, 3) and np.array_equal(op, ref[0])
    q = synth_data.uniform_cloud(4, 2000, extent=(1.0, 1.0, 1.0))
    op, ln, st = emu.subsample_items(q, [2000], 0.03)           # (34^3 cells: inside the 4096-point class's 65 536)
    ref = oops.subsample_batch(q, [2000], sampleDl=0.03)
    assert st == 0 and len(op) > 1850 and np.array_equal(op, ref[0])
    assert emu.subsample_items(q, [2000], 0.02)[2] == 2          # 50^3 cells: more than that class takes -> the caller's cue


def test_subsample_per_item_kernel_reports_what_it_cannot_take():
    """Limits of the per-item kernel: more points than fit its LDS -> ML3D_E_UNSUPPORTED at the call (host value); a grid of more
    cells than its bitmap -> stats[1] == 2 after the count (device check).  Both are the caller's cue for the sort-based op."""
    L = emu.lib()
    nmax = int(L.ml3d_subsample_items_max_points())
    assert nmax >= 10000                      # a KPConv input sphere (batch_limit / max_in_points of the Toronto3D config)
    big = synth_data.uniform_cloud(1, nmax + 1)
    assert emu.subsample_items(big, [nmax + 1], 0.5)[2] == -4
    wide = synth_data.uniform_cloud(2, 4000)   # 10 x 10 x 2 m at 0.02 m: 500 * 500 * 100 cells
    assert emu.subsample_items(wide, [4000], 0.02)[2] == 2
    assert emu.subsample_items(np.concatenate([wide, wide]), [4000, 4000], 0.5)[2] == 0


@pytest.mark.parametrize("max_points,max_voxels", [(2**62, 2**62), (32, 2**62), (32, 40)])
def test_voxelize_orders_long_runs_like_the_oracle(max_points, max_voxels):
    """Pillars with 1 point, with a few, with tens, with hundreds and with thousands of points (the pillars next to the sensor), points
    outside the range, an empty item -- coordinates, ragged point lists (original point order inside a pillar, the first max_points
    kept) and batch splits identical to the oracle, for every truncation.  (Written for round 6's occupancy-bitmap grouping, which was
    exact and SLOWER than the stable sort on exactly these pillars -- profiles/r06_voxelize_bitmap_ab.log -- and was removed.)"""
    rng = np.random.default_rng(21)
    spread = rng.random((5000, 3), dtype=np.float32) * [69, 79, 3.9] + [0, -39.6, -3]
    dense = rng.random((600, 3), dtype=np.float32) * [0.15, 0.15, 3.5] + [10.0, 0.0, -3]            # one pillar, 600 points
    huge = rng.random((5000, 3), dtype=np.float32) * [0.15, 0.15, 3.5] + [20.005, 5.125, -3]           # one pillar, 5000 points
    mid = rng.random((2000, 3), dtype=np.float32) * [1.2, 1.2, 3.5] + [30.0, -10.0, -3]              # ~60 pillars of ~35 points
    out = rng.random((300, 3), dtype=np.float32) * 5 + [100, 100, 10]                                # outside the range
    a = np.concatenate([spread, dense, huge, mid, out]).astype(np.float32)
    a = a[rng.permutation(len(a))]
    b = (rng.random((3000, 3), dtype=np.float32) * [69, 79, 3.9] + [0, -39.6, -3]).astype(np.float32)
    pts = np.concatenate([a, b])
    rs = [0, len(a), len(a), len(pts)]
    vs, mn, mx = [0.16, 0.16, 4], [0, -39.68, -3], [69.12, 39.68, 1]
    c, pi, prs, bs = emu.voxelize(pts, rs, vs, mn, mx, max_points, max_voxels)
    ref = oops.voxelize(pts, rs, vs, mn, mx, max_points, max_voxels)
    assert np.array_equal(bs, ref.voxel_batch_splits)
    assert np.array_equal(c, ref.voxel_coords)
    assert np.array_equal(prs, ref.voxel_point_row_splits)
    assert np.array_equal(pi, ref.voxel_point_indices)
    if max_points > 5000:
        assert np.diff(prs).max() >= 5000


def _vox_cloud(seed, sizes):
    """KITTI-shaped items: a spread of returns over the range, pillars of tens / hundreds / thousands of points next to the sensor,
    returns outside the range, in shuffled order"""
    rng = np.random.default_rng(seed)
    items = []
    for n in sizes:
        if n == 0:
            items.append(np.zeros((0, 3), np.float32))
            continue
        parts = [rng.random((n * 6 // 10, 3), dtype=np.float32) * [69, 79, 3.9] + [0, -39.6, -3],
                 rng.random((n // 10, 3), dtype=np.float32) * [0.15, 0.15, 3.5] + [10.0, 0.0, -3],
                 rng.random((n // 10, 3), dtype=np.float32) * [1.2, 1.2, 3.5] + [30.0, -10.0, -3],
                 rng.random((n // 10, 3), dtype=np.float32) * 40 + [70, 40, 1]]
        a = np.concatenate(parts)
        a = np.concatenate([a, rng.random((n - len(a), 3), dtype=np.float32) * [3, 3, 3] + [5, 5, -3]]).astype(np.float32)
        items.append(a[rng.permutation(n)])
    return np.concatenate(items), np.concatenate([[0], np.cumsum(sizes)])


@pytest.mark.parametrize("sizes,max_points,max_voxels", [
    ((30000, 0, 41000, 70000, 9), 32, 2**62),        # 69 tiles: three groups of the hand-off, the last one partial
    ((66000, 1, 0), 32, 900),                        # max_voxels cuts an item; 33 tiles: a group of one
    ((2048 * 32,), 2**62, 2**62),                    # exactly one full group, the last tile full
    ((2049,), 5, 2**62), ((1,), 32, 2**62)])
def test_voxelize_fused_hand_off_over_many_tiles_matches_the_oracle(sizes, max_points, max_voxels):
    """The fused voxelize (one launch per radix pass + one grouping launch, tiles handing their counts on through tagged words in
    global memory: sort.h) at sizes where the flat two-level prefix has several groups, partial groups and single tiles: everything
    identical to the oracle.  The emulator runs the workgroups of a launch on several OS threads, so the waits are real waits."""
    pts, rs = _vox_cloud(len(sizes) * 7 + sizes[0] % 5, sizes)
    vs, mn, mx = [0.16, 0.16, 4], [0, -39.68, -3], [69.12, 39.68, 1]
    c, pi, prs, bs = emu.voxelize(pts, rs, vs, mn, mx, max_points, max_voxels)
    ref = oops.voxelize(pts, rs, vs, mn, mx, max_points, max_voxels)
    assert np.array_equal(bs, ref.voxel_batch_splits)
    assert np.array_equal(c, ref.voxel_coords)
    assert np.array_equal(prs, ref.voxel_point_row_splits)
    assert np.array_equal(pi, ref.voxel_point_indices)


def test_voxelize_fused_with_no_point_in_range_and_with_every_point_in_one_pillar():
    far = (np.random.default_rng(2).random((5000, 3), dtype=np.float32) * 5 + [100, 100, 10]).astype(np.float32)
    c, pi, prs, bs = emu.voxelize(far, [0, 2000, 5000], [0.16, 0.16, 4], [0, -39.68, -3], [69.12, 39.68, 1], 32, 100)
    assert c.shape == (0, 3) and pi.shape == (0,) and prs.tolist() == [0] and bs.tolist() == [0, 0, 0]
    one = (np.random.default_rng(3).random((9000, 3), dtype=np.float32) * [0.1, 0.1, 3] + [3.05, 3.05, -3]).astype(np.float32)
    c, pi, prs, bs = emu.voxelize(one, [0, 0, 9000], [0.16, 0.16, 4], [0, -39.68, -3], [69.12, 39.68, 1], 2**62, 2**62)
    ref = oops.voxelize(one, [0, 0, 9000], [0.16, 0.16, 4], [0, -39.68, -3], [69.12, 39.68, 1])
    assert np.array_equal(c, ref.voxel_coords) and np.array_equal(pi, ref.voxel_point_indices) and bs.tolist() == [0, 0, 1]
    assert prs.tolist() == [0, 9000]


@pytest.mark.parametrize("batch", [64, 65, 96, 128, 130])
def test_radius_segment_lookup_by_ballot_over_many_items_with_empty_ones(batch):
    """Wave-per-query kernels find a query's batch item with ONE lane-parallel load of the row splits and a ballot count
    (grid.h seg_locate_wave: <= 64 items one load, <= 128 two, beyond that the binary search): 64 / 65 / 96 (a KPConv batch) / 128 /
    130 items, several of them empty on either side, queries and supports with different splits."""
    rng = np.random.default_rng(batch)
    plen = rng.integers(0, 60, batch); plen[[1, batch // 2, batch - 1]] = 0
    qlen = rng.integers(0, 25, batch); qlen[[0, 2, batch // 2]] = 0
    ps, qs = np.concatenate([[0], np.cumsum(plen)]), np.concatenate([[0], np.cumsum(qlen)])
    p = rng.random((ps[-1], 3), dtype=np.float32)
    q = rng.random((qs[-1], 3), dtype=np.float32)
    idx, rs = emu.radius(p, ps, q, qs, 0.3)
    ref = oops.fixed_radius_search(p, q, 0.3, ps, qs)
    assert np.array_equal(rs, ref.neighbors_row_splits) and np.array_equal(idx, ref.neighbors_index)
    dense = emu.radius(p, ps, q, qs, 0.3, dense=True)
    cols = int(np.diff(ref.neighbors_row_splits).max())
    assert np.array_equal(dense, oops.ragged_to_dense(ref.neighbors_index.reshape(-1, 1), ref.neighbors_row_splits, cols,
                                                      np.array([len(p)], np.int32))[:, :, 0])
