"""``ml3d.layers`` — module-style mirrors of ``open3d.ml.torch.layers`` for the hot path."""
import torch

from .. import ops


class FixedRadiusSearch(torch.nn.Module):
    """Drop-in for ``open3d.ml.torch.layers.FixedRadiusSearch`` as used by ``batch_neighbors``
    (ml3d/torch/models/kpconv.py:2021-2026): ``forward(points, queries, radius, points_row_splits,
    queries_row_splits)`` -> namedtuple(neighbors_index i32, neighbors_row_splits i64, neighbors_distance).
    Only the configuration the reference uses is supported (L2, query point not ignored, int32 indices)."""

    def __init__(self, metric="L2", ignore_query_point=False, return_distances=False,
                 max_hash_table_size=32 * 2 ** 20, index_dtype=torch.int32, **kwargs):
        super().__init__()
        if metric != "L2" or ignore_query_point or index_dtype != torch.int32:
            raise RuntimeError("FixedRadiusSearch: only metric='L2', ignore_query_point=False, int32 indices")
        self.return_distances = return_distances

    def forward(self, points, queries, radius, points_row_splits=None, queries_row_splits=None, *args, **kwargs):
        return ops.fixed_radius_search(points, queries, float(radius), points_row_splits, queries_row_splits,
                                       return_distances=self.return_distances)
