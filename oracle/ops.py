"""ctypes front end of the CPU ORACLE (test infrastructure, not product code).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  Every function restates an op the reference imports
from the un-vendored ``open3d`` wheel; the reference call site each one follows
is cited in ``ml3d_oracle.c`` next to the C body.
"""
import ctypes as C
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libml3d_oracle.so")
    src = os.path.join(_HERE, "ml3d_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libml3d_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.ml3d_oracle_iou_bev.restype = C.c_float
        _LIB.ml3d_oracle_nms.restype = C.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def num_threads():
    return int(lib().ml3d_oracle_num_threads())


def knn_search(points, queries, k, brute=False, return_distances=False):
    """[Nq, min(k, Ns)] int32 indices in canonical (d2, idx) order.

    Restates DataProcessing.knn_search (ml3d/datasets/utils/dataprocessing.py:87-103).
    """
    points, queries = _f32(points), _f32(queries)
    ns, nq = points.shape[0], queries.shape[0]
    kk = min(int(k), ns)
    idx = np.empty((nq, kk), np.int32)
    d2 = np.empty((nq, kk), np.float32)
    fn = lib().ml3d_oracle_knn_brute if brute else lib().ml3d_oracle_knn
    rc = fn(_p(points), C.c_int64(ns), _p(queries), C.c_int64(nq), C.c_int(int(k)), _p(idx), _p(d2))
    if rc != 0:
        raise RuntimeError("oracle knn failed: %d" % rc)
    return (idx, d2) if return_distances else idx


def knn_search_batched(points, points_row_splits, queries, queries_row_splits, k):
    points, queries = _f32(points), _f32(queries)
    ps, qs = _i64(points_row_splits), _i64(queries_row_splits)
    nq = queries.shape[0]
    idx = np.empty((nq, k), np.int32)
    d2 = np.empty((nq, k), np.float32)
    rc = lib().ml3d_oracle_knn_batched(_p(points), _p(ps), _p(queries), _p(qs),
                                       C.c_int64(len(ps) - 1), C.c_int(int(k)), _p(idx), _p(d2))
    if rc != 0:
        raise RuntimeError("oracle knn_batched failed: %d" % rc)
    return idx, d2


RadiusResult = namedtuple("RadiusResult",
                          ["neighbors_index", "neighbors_row_splits", "neighbors_distance"])


def fixed_radius_search(points, queries, radius, points_row_splits=None, queries_row_splits=None,
                        return_distances=False):
    """Ragged neighbours with d2 <= r^2, canonical (d2, idx) order, GLOBAL indices.

    Restates FixedRadiusSearch()(supports, queries, r, s_splits, q_splits) as used at
    ml3d/torch/models/kpconv.py:2021-2026.
    """
    points, queries = _f32(points), _f32(queries)
    if points_row_splits is None:
        points_row_splits = [0, points.shape[0]]
    if queries_row_splits is None:
        queries_row_splits = [0, queries.shape[0]]
    ps, qs = _i64(points_row_splits), _i64(queries_row_splits)
    nq = queries.shape[0]
    splits = np.zeros(nq + 1, np.int64)
    B = C.c_int64(len(ps) - 1)
    r = C.c_float(float(radius))
    rc = lib().ml3d_oracle_radius(_p(points), _p(ps), _p(queries), _p(qs), B, r, C.c_int(0),
                                  _p(splits), None, None)
    if rc != 0:
        raise RuntimeError("oracle radius(count) failed: %d" % rc)
    total = int(splits[-1])
    idx = np.empty(total, np.int32)
    d2 = np.empty(total, np.float32) if return_distances else None
    rc = lib().ml3d_oracle_radius(_p(points), _p(ps), _p(queries), _p(qs), B, r, C.c_int(1),
                                  _p(splits), _p(idx), _p(d2))
    if rc != 0:
        raise RuntimeError("oracle radius(fill) failed: %d" % rc)
    return RadiusResult(idx, splits, d2 if return_distances else np.empty(0, np.float32))


def ragged_to_dense(values, row_splits, out_col_size, default_value):
    """Restates open3d.ml.torch.ops.ragged_to_dense (kpconv.py:2030, point_pillars.py:364)."""
    values = np.ascontiguousarray(values)
    rs = _i64(row_splits)
    rows = len(rs) - 1
    inner = values.shape[1:]
    dv = np.ascontiguousarray(np.broadcast_to(np.asarray(default_value, values.dtype), inner))
    elem = int(values.dtype.itemsize * int(np.prod(inner, dtype=np.int64)))
    out = np.empty((rows, int(out_col_size)) + tuple(inner), values.dtype)
    lib().ml3d_oracle_ragged_to_dense(_p(values), _p(rs), C.c_int64(rows), C.c_int64(int(out_col_size)),
                                      _p(dv), C.c_int64(elem), _p(out))
    return out


VoxelizeResult = namedtuple("VoxelizeResult", ["voxel_coords", "voxel_point_indices",
                                               "voxel_point_row_splits", "voxel_batch_splits"])


def voxelize(points, row_splits, voxel_size, points_range_min, points_range_max,
             max_points_per_voxel=np.iinfo(np.int64).max, max_voxels=np.iinfo(np.int64).max):
    """Restates open3d.ml.torch.ops.voxelize (ml3d/torch/models/point_pillars.py:354-357)."""
    points = _f32(points)
    rs = _i64(row_splits)
    vs, mn, mx = _f32(voxel_size), _f32(points_range_min), _f32(points_range_max)
    B = len(rs) - 1
    nv, ni = C.c_int64(0), C.c_int64(0)
    args = (_p(points), _p(rs), C.c_int64(B), _p(vs), _p(mn), _p(mx),
            C.c_int64(int(max_points_per_voxel)), C.c_int64(int(max_voxels)))
    lib().ml3d_oracle_voxelize(*args, C.c_int(0), C.byref(nv), C.byref(ni), None, None, None, None)
    coords = np.empty((nv.value, 3), np.int32)
    pidx = np.empty(ni.value, np.int64)
    prs = np.zeros(nv.value + 1, np.int64)
    bs = np.zeros(B + 1, np.int64)
    lib().ml3d_oracle_voxelize(*args, C.c_int(1), C.byref(nv), C.byref(ni), _p(coords), _p(pidx),
                               _p(prs), _p(bs))
    return VoxelizeResult(coords, pidx, prs, bs)


def subsample(points, features=None, classes=None, sampleDl=0.1, verbose=0):
    """Restates open3d.ml.contrib.subsample (ml3d/datasets/utils/dataprocessing.py:32-49)."""
    points = _f32(points)
    n = points.shape[0]
    feats = None if features is None else _f32(features)
    labels = None if classes is None else np.ascontiguousarray(classes, np.int32).reshape(-1)
    fdim = 0 if feats is None else feats.shape[1]
    m = C.c_int64(0)
    lib().ml3d_oracle_subsample(_p(points), C.c_int64(n), _p(feats), C.c_int64(fdim), _p(labels),
                                C.c_float(float(sampleDl)), C.c_int(0), C.byref(m), None, None, None)
    op = np.empty((m.value, 3), np.float32)
    of = None if feats is None else np.empty((m.value, fdim), np.float32)
    ol = None if labels is None else np.empty(m.value, np.int32)
    lib().ml3d_oracle_subsample(_p(points), C.c_int64(n), _p(feats), C.c_int64(fdim), _p(labels),
                                C.c_float(float(sampleDl)), C.c_int(1), C.byref(m), _p(op), _p(of), _p(ol))
    out = [op]
    if of is not None:
        out.append(of)
    if ol is not None:
        out.append(ol)
    return out[0] if len(out) == 1 else tuple(out)


def subsample_batch(points, batches_len, features=None, classes=None, sampleDl=0.1, max_p=0,
                    verbose=0):
    """Restates open3d.ml.contrib.subsample_batch (ml3d/torch/models/kpconv.py:2098-2155)."""
    points = _f32(points)
    outs, lens = [], []
    i0 = 0
    for ln in np.asarray(batches_len).reshape(-1):
        ln = int(ln)
        sl = slice(i0, i0 + ln)
        r = subsample(points[sl], None if features is None else features[sl],
                      None if classes is None else np.asarray(classes)[sl], sampleDl)
        r = r if isinstance(r, tuple) else (r,)
        if max_p and max_p > 0:
            r = tuple(x[:max_p] for x in r)
        outs.append(r)
        lens.append(r[0].shape[0])
        i0 += ln
    cat = [np.concatenate([o[j] for o in outs], 0) for j in range(len(outs[0]))]
    return (cat[0], np.asarray(lens, np.int32)) + tuple(cat[1:])


def nms(boxes, scores, nms_overlap_thresh):
    """Restates open3d.ml.torch.ops.nms (ml3d/torch/utils/objdet_helper.py:346)."""
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), np.int64)
    m = lib().ml3d_oracle_nms(_p(boxes), _p(scores), C.c_int64(n), C.c_float(float(nms_overlap_thresh)),
                              _p(keep))
    return keep[:m].copy()


def topk_rows(values, k):
    """Restates ``max_scores.topk(self.nms_pre)`` of Anchor3DHead.get_bboxes_single (ml3d/torch/models/point_pillars.py:985-992)
    for every row of ``values`` [rows, n]: indices [rows, k] of the k largest in descending value, NaN above +inf (torch.topk's
    convention).  torch leaves the order of EQUAL values unspecified; the canonical order fixed here is ascending index (also for
    the ties at the k-th value).  Pinned to torch.topk itself in tests/test_oracle_ops.py: values identical, indices identical
    wherever a row's values are distinct."""
    v = _f32(values)
    v = v.reshape(1, -1) if v.ndim == 1 else v
    nan = np.isnan(v)
    key = np.where(nan, np.float32(np.inf), v)
    out = np.empty((v.shape[0], int(k)), np.int64)
    for r in range(v.shape[0]):
        # lexsort: last key is the primary one -- NaN first, then descending value, then ascending index
        out[r] = np.lexsort((np.arange(v.shape[1]), -key[r].astype(np.float64), ~nan[r]))[:int(k)]
    return out


def _corner_form(cx, cy, dx, dy, r):
    hx, hy = np.float32(0.5) * dx, np.float32(0.5) * dy
    return np.stack([cx - hx, cy - hy, cx + hx, cy + hy, r], -1).astype(np.float32)


def iou_bev(boxes_a, boxes_b):
    """Pairwise rotated BEV IoU, boxes (x, z, w, l, yaw) -- restates the call ml3d/metrics/mAP.py:85 makes into
    open3d.ml.contrib.iou_bev_*; the rotated intersection is the NMS oracle's (ml3d_oracle_iou_bev)."""
    a, b = _f32(boxes_a).reshape(-1, 5), _f32(boxes_b).reshape(-1, 5)
    ca = _corner_form(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4])
    cb = _corner_form(b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 4])
    out = np.zeros((len(a), len(b)), np.float32)
    L = lib()
    for i in range(len(a)):
        for j in range(len(b)):
            out[i, j] = L.ml3d_oracle_iou_bev(_p(np.ascontiguousarray(ca[i])), _p(np.ascontiguousarray(cb[j])))
    return out


def iou_3d(boxes_a, boxes_b):
    """Pairwise 3-D IoU, boxes (x, y, z, w, h, l, yaw), y = bottom face, y axis down (ml3d/metrics/mAP.py:87)."""
    a, b = _f32(boxes_a).reshape(-1, 7), _f32(boxes_b).reshape(-1, 7)
    ca = _corner_form(a[:, 0], a[:, 2], a[:, 3], a[:, 5], a[:, 6])
    cb = _corner_form(b[:, 0], b[:, 2], b[:, 3], b[:, 5], b[:, 6])
    L = lib()
    L.ml3d_oracle_inter_bev.restype = C.c_float
    out = np.zeros((len(a), len(b)), np.float32)
    f = np.float32
    for i in range(len(a)):
        for j in range(len(b)):
            inter = f(L.ml3d_oracle_inter_bev(_p(np.ascontiguousarray(ca[i])), _p(np.ascontiguousarray(cb[j]))))
            aa = f(f(ca[i, 2] - ca[i, 0]) * f(ca[i, 3] - ca[i, 1]))
            ab = f(f(cb[j, 2] - cb[j, 0]) * f(cb[j, 3] - cb[j, 1]))
            top = max(f(a[i, 1] - a[i, 4]), f(b[j, 1] - b[j, 4]))
            bot = min(a[i, 1], b[j, 1])
            oh = max(f(bot - top), f(0))
            num = f(inter * oh)
            den = f(f(f(aa * a[i, 4]) + f(ab * b[j, 4])) - num)
            out[i, j] = f(num / den) if den > 1e-8 else 0.0
    return out
