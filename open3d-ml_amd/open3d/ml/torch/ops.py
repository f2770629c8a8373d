"""``open3d.ml.torch.ops`` — the five ops the inference hot path takes from the wheel.

Each function accepts the tensors the reference passes — CPU tensors from collate code (``kpconv.py:2021-2032``), GPU
tensors from the models, small CPU tensors for voxel sizes (``point_pillars.py:317-320``) — runs on the MI355X and
returns the reference's named tuples with every tensor on the device of the first input."""
from collections import namedtuple

import torch

from ... import _product as _P

VoxelizeResult = namedtuple("VoxelizeResult", ["voxel_coords", "voxel_point_indices", "voxel_point_row_splits",
                                               "voxel_batch_splits"])
KnnResult = namedtuple("KnnResult", ["neighbors_index", "neighbors_row_splits", "neighbors_distance"])
RadiusResult = namedtuple("RadiusResult", ["neighbors_index", "neighbors_row_splits", "neighbors_distance"])


def voxelize(points, row_splits, voxel_size, points_range_min, points_range_max, max_points_per_voxel=2 ** 62,
             max_voxels=2 ** 62):
    p, src = _P.to_dev(points, torch.float32)
    r = _P.ops().voxelize(p, row_splits, voxel_size, points_range_min, points_range_max, max_points_per_voxel, max_voxels)
    return VoxelizeResult(*(_P.back(t, src) for t in r))


def ragged_to_dense(values, row_splits, out_col_size, default_value):
    v, src = _P.to_dev(values)
    return _P.back(_P.ops().ragged_to_dense(v, _P.to_dev(row_splits, torch.int64)[0], int(out_col_size), default_value), src)


def fixed_radius_search(points, queries, radius, points_row_splits, queries_row_splits, hash_table_splits=None,
                        hash_table_index=None, hash_table_cell_splits=None, index_dtype=3, metric="L2",
                        ignore_query_point=False, return_distances=False):
    if metric != "L2" or ignore_query_point:
        raise NotImplementedError("fixed_radius_search: metric='L2', ignore_query_point=False only")
    p, src = _P.to_dev(points, torch.float32)
    q = p if queries is points else _P.to_dev(queries, torch.float32)[0]
    r = _P.ops().fixed_radius_search(p, q, float(radius), points_row_splits, queries_row_splits, return_distances)
    return RadiusResult(_P.back(r.neighbors_index, src), _P.back(r.neighbors_row_splits, src),
                        _P.back(r.neighbors_distance, src))


def knn_search(points, queries, k, points_row_splits, queries_row_splits, index_dtype=3, metric="L2",
               ignore_query_point=False, return_distances=False):
    """(API surface only in this repository's scope: the reference's in-scope models call ``open3d.core.nns``.)"""
    if metric != "L2" or ignore_query_point:
        raise NotImplementedError("knn_search: metric='L2', ignore_query_point=False only")
    p, src = _P.to_dev(points, torch.float32)
    q = p if queries is points else _P.to_dev(queries, torch.float32)[0]
    r = _P.ops().knn_search(p, q, int(k), points_row_splits, queries_row_splits, return_distances=return_distances)
    nq = q.shape[0]
    splits = torch.arange(0, (nq + 1) * int(k), int(k), dtype=torch.int64, device=p.device)
    d = r.neighbors_distance if return_distances else torch.empty(0, dtype=torch.float32, device=p.device)
    return KnnResult(_P.back(r.neighbors_index.reshape(-1), src), _P.back(splits, src), _P.back(d.reshape(-1), src))


def nms(boxes, scores, nms_overlap_thresh):
    """keep indices (int64) of rotated-BEV NMS: boxes [N, 5] = (x0, y0, x1, y1, r), ``objdet_helper.py:346-348``"""
    b, src = _P.to_dev(boxes, torch.float32)
    s, _ = _P.to_dev(scores, torch.float32)
    return _P.back(_P.ops().nms(b, s, float(nms_overlap_thresh)), src)


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("open3d.ml.torch.ops.%s belongs to a model outside this repository's scope (SURVEY.md §2: "
                                  "PointRCNN / SparseConvNet / PVCNN / PointTransformer); only its import target exists" % name)
    fn.__name__ = name
    return fn


# import targets of the reference's out-of-scope models (ml3d/torch/models/{sparseconvnet,pvcnn,point_rcnn}.py,
# ml3d/torch/utils/{pointnet,roipool3d}): resolvable so that `import ml3d.torch.models` works, inert otherwise
for _n in ("reduce_subarrays_sum", "roi_pool", "furthest_point_sampling", "three_nn", "three_interpolate",
           "three_interpolate_grad", "ball_query", "trilinear_devoxelize_forward", "trilinear_devoxelize_backward",
           "continuous_conv", "sparse_conv", "sparse_conv_transpose", "invert_neighbors_list", "build_spatial_hash_table"):
    globals()[_n] = _out_of_scope(_n)
del _n
