"""``open3d.ml.torch.layers`` — ``FixedRadiusSearch`` as ``batch_neighbors`` builds it (``kpconv.py:2021-2026``)."""
import torch

from . import ops


class FixedRadiusSearch(torch.nn.Module):

    def __init__(self, metric="L2", ignore_query_point=False, return_distances=False, max_hash_table_size=32 * 2 ** 20,
                 index_dtype=torch.int32, **kwargs):
        super().__init__()
        if metric != "L2" or ignore_query_point:
            raise NotImplementedError("FixedRadiusSearch: metric='L2', ignore_query_point=False only")
        self.return_distances = return_distances
        self.index_dtype = index_dtype

    def forward(self, points, queries, radius, points_row_splits=None, queries_row_splits=None, hash_table_size_factor=1 / 64,
                hash_table=None):
        if points_row_splits is None:
            points_row_splits = torch.LongTensor([0, points.shape[0]])
        if queries_row_splits is None:
            queries_row_splits = torch.LongTensor([0, queries.shape[0]])
        r = ops.fixed_radius_search(points, queries, radius, points_row_splits, queries_row_splits,
                                    return_distances=self.return_distances)
        if self.index_dtype == torch.int64:
            r = r._replace(neighbors_index=r.neighbors_index.long())
        return r


class KNNSearch(torch.nn.Module):

    def __init__(self, metric="L2", ignore_query_point=False, return_distances=False, index_dtype=torch.int32, **kwargs):
        super().__init__()
        self.return_distances = return_distances
        self.index_dtype = index_dtype

    def forward(self, points, queries, k, points_row_splits=None, queries_row_splits=None):
        if points_row_splits is None:
            points_row_splits = torch.LongTensor([0, points.shape[0]])
        if queries_row_splits is None:
            queries_row_splits = torch.LongTensor([0, queries.shape[0]])
        r = ops.knn_search(points, queries, k, points_row_splits, queries_row_splits, return_distances=self.return_distances)
        if self.index_dtype == torch.int64:
            r = r._replace(neighbors_index=r.neighbors_index.long())
        return r


class _OutOfScopeLayer(torch.nn.Module):
    """Import target of the reference's out-of-scope SparseConvNet (``sparseconvnet.py:9``): constructing it raises."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("open3d.ml.torch.layers.%s: sparse / continuous convolutions are outside this "
                                  "repository's scope (SURVEY.md §2)" % type(self).__name__)


class SparseConv(_OutOfScopeLayer):
    pass


class SparseConvTranspose(_OutOfScopeLayer):
    pass


class ContinuousConv(_OutOfScopeLayer):
    pass
