"""``open3d.core.nns.NearestNeighborSearch`` — the k-NN of ``DataProcessing.knn_search``
(ml3d/datasets/utils/dataprocessing.py:87-103; 8 calls per frame from ``RandLANet.transform``, randlanet.py:218-229).

numpy in, numpy out, computed by the grid + tile k-NN kernels of ``libml3d_hip.so`` (``ml3d_knn_search``): indices in the
canonical ascending (d2, index) order, int64 like upstream."""
import numpy as np
import torch

from .. import _product as P


class NearestNeighborSearch:

    def __init__(self, dataset_points, index_dtype=None):
        from . import Tensor
        a = dataset_points.numpy() if isinstance(dataset_points, Tensor) else np.asarray(dataset_points)
        self._host = np.ascontiguousarray(a, dtype=np.float32)
        self._dev = None

    def _points(self):
        if self._dev is None:
            self._dev = torch.from_numpy(self._host).to(P.device())
        return self._dev

    # index builders: the uniform grid is rebuilt inside every search call (it costs less than the search); these only
    # upload the points once
    def knn_index(self):
        self._points()
        return True

    def fixed_radius_index(self, radius=None):
        self._points()
        return True

    def hybrid_index(self, radius=None):
        self._points()
        return True

    def knn_search(self, query_points, knn):
        """-> (indices [Nq, knn] int64, squared distances [Nq, knn] float32) as ``open3d.core.Tensor``."""
        from . import Tensor
        q = query_points.numpy() if isinstance(query_points, Tensor) else np.asarray(query_points)
        q = np.ascontiguousarray(q, dtype=np.float32)
        pts = self._points()
        same = q.shape == self._host.shape and (q is self._host or np.shares_memory(q, self._host) or
                                                np.array_equal(q, self._host))
        qd = pts if same else torch.from_numpy(q).to(pts.device)
        k = min(int(knn), self._host.shape[0])
        res = P.ops().knn_search(pts, qd, k, return_distances=True)
        return Tensor(res.neighbors_index.cpu().numpy().astype(np.int64)), Tensor(res.neighbors_distance.cpu().numpy())

    def fixed_radius_search(self, query_points, radius, sort=True):
        """-> (indices [T] int64, squared distances [T], row_splits [Nq + 1] int64), rows ascending (d2, index)."""
        from . import Tensor
        q = query_points.numpy() if isinstance(query_points, Tensor) else np.asarray(query_points)
        qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(P.device())
        r = P.ops().fixed_radius_search(self._points(), qd, float(radius), return_distances=True)
        return (Tensor(r.neighbors_index.cpu().numpy().astype(np.int64)), Tensor(r.neighbors_distance.cpu().numpy()),
                Tensor(r.neighbors_row_splits.cpu().numpy()))
