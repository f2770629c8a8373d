"""Host-side data path shared by the segmentation models (``preprocess`` of randlanet.py:115-154 / kpconv.py:353-396):
grid subsampling, the search structure the pipeline's samplers query, and the raw -> sub-cloud projection, all on the
MI355X ops."""
import numpy as np
import torch

from ... import _abi
from ... import ops


class GpuSearchTree:
    """What ``preprocess`` stores as ``data['search_tree']`` (the reference builds an sklearn ``KDTree`` there,
    randlanet.py:142).  The pipeline's samplers only use ``.data``, ``.query(X, k)`` and ``.query_radius(X, r)``
    (ml3d/datasets/samplers/semseg_spatially_regular.py:86-91; randlanet.py:147-150, 389); all are served by the GPU ops:

    * the num_points-nearest patch query (ONE centre, k up to the whole cloud) by a keyed radix sort of the cloud
      (``ops.nearest_to_center``) in sklearn's own order -- ascending float64 reduced distance -- because the patch order is
      observable downstream (``random.shuffle``, then the prefix subsampling of ``RandLANet.transform``);
    * small k / many queries by the grid k-NN (float32 squared distances, ascending (d2, index));
    * ``query_radius`` by the fixed-radius search.  sklearn documents that result as UNSORTED: its order is the traversal
      order of the tree's internal index array, which only sklearn's own build reproduces.  The radius sampler shuffles the
      list and then cuts it by position (kpconv.py:480-489), so bit-identical spheres need that very order:
      ``host_index=True`` (model kwarg ``sampler_index='sklearn'``) builds the reference's ``KDTree`` for this ONE
      single-centre query per sphere -- the sampler's index only, never a neighbour search of the hot path.  The default
      answers from the GPU in ascending (d2, index) order: the same SET, a different (equally arbitrary) order."""

    def __init__(self, points, device, host_index=False):
        self.data = np.ascontiguousarray(points, dtype=np.float32)
        self.device = torch.device(device)
        self._dev = None
        self._host = None
        if host_index:
            from sklearn.neighbors import KDTree      # the reference's own dependency (randlanet.py:8, kpconv.py:9)
            self._host = KDTree(self.data)

    def __getstate__(self):
        """The reference's dataloader caches ``preprocess`` results with ``np.save`` (pickle; ml3d/utils/dataset_helper.py:
        65-69) and re-reads the file for every patch: only host state is stored, the device copy is rebuilt on first use."""
        return dict(data=self.data, device=str(self.device), host=self._host)

    def __setstate__(self, st):
        self.data, self.device, self._host, self._dev = st['data'], torch.device(st['device']), st['host'], None

    def _pts(self):
        if self._dev is None:
            self._dev = torch.from_numpy(self.data).to(self.device)
        return self._dev

    def query(self, X, k=1, return_distance=True, **unused):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float32).reshape(-1, 3))
        k = int(k)
        if X.shape[0] == 1 and k > 16:
            idx, d2 = ops.nearest_to_center(self._pts(), X[0], k, return_distances=True)
            idx, dist = idx.reshape(1, -1), np.sqrt(d2.cpu().numpy()).reshape(1, -1)
        else:
            r = ops.knn_search(self._pts(), torch.from_numpy(X).to(self.device), k, return_distances=True)
            idx, dist = r.neighbors_index, np.sqrt(r.neighbors_distance.cpu().numpy().astype(np.float64))
        idx = idx.cpu().numpy().astype(np.int64)
        return (dist, idx) if return_distance else idx

    def query_radius(self, X, r, **unused):
        """-> object array with one int64 index array per query row (sklearn's ``KDTree.query_radius``; used by the
        radius-based point sampler of KPConv, semseg_spatially_regular.py:86-87)."""
        if self._host is not None:
            return self._host.query_radius(X, r=r)
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float32).reshape(-1, 3))
        res = ops.fixed_radius_search(self._pts(), torch.from_numpy(X).to(self.device), float(r))
        idx = res.neighbors_index.cpu().numpy().astype(np.int64)
        rs = res.neighbors_row_splits.cpu().numpy()
        out = np.empty(X.shape[0], dtype=object)
        for i in range(X.shape[0]):
            out[i] = idx[rs[i]:rs[i + 1]]
        return out


def preprocess_segmentation(data, attr, grid_size, device, proj_splits=("test", "testing"), host_index=False):
    """raw cloud dict -> {'point', 'feat', 'label', 'search_tree'[, 'proj_inds']} like the reference's ``preprocess``."""
    dev = torch.device(device)
    _abi.require_gpu(dev, "preprocess (grid subsample / projection on the HIP ops)")
    points = np.array(data['point'][:, 0:3], dtype=np.float32)
    if 'label' not in data or data['label'] is None:
        labels = np.zeros((points.shape[0],), dtype=np.int32)
    else:
        labels = np.array(data['label'], dtype=np.int32).reshape((-1,))
    feat = None if ('feat' not in data or data['feat'] is None) else np.array(data['feat'], dtype=np.float32)
    p = torch.from_numpy(points).to(dev)
    lab = torch.from_numpy(labels).to(dev)
    if feat is None:
        sub_points, sub_labels = ops.subsample(p, classes=lab, sampleDl=grid_size)
        sub_feat = None
    else:
        sub_points, sub_feat, sub_labels = ops.subsample(p, features=torch.from_numpy(feat).to(dev), classes=lab,
                                                         sampleDl=grid_size)
    out = dict()
    out['point'] = sub_points.cpu().numpy()
    out['feat'] = None if sub_feat is None else sub_feat.cpu().numpy()
    out['label'] = sub_labels.cpu().numpy().astype(np.int32)
    tree = GpuSearchTree(out['point'], dev, host_index=host_index)
    tree._dev = sub_points
    out['search_tree'] = tree
    if attr['split'] in proj_splits:
        proj = ops.knn_search(sub_points, p, 1).neighbors_index
        out['proj_inds'] = proj.reshape(-1).cpu().numpy().astype(np.int32)
    return out


def create_3D_rotations(axis, angle):
    """Rotation matrices from axes [N, 3] and angles [N] (Rodrigues); restates ``create_3D_rotations``
    (ml3d/datasets/utils/operations.py:21-40) with the same float64 operation order."""
    t1 = np.cos(angle)
    t2 = 1 - t1
    t3 = axis[:, 0] * axis[:, 0]
    t6 = t2 * axis[:, 0]
    t7 = t6 * axis[:, 1]
    t8 = np.sin(angle)
    t9 = t8 * axis[:, 2]
    t11 = t6 * axis[:, 2]
    t12 = t8 * axis[:, 1]
    t15 = axis[:, 1] * axis[:, 1]
    t19 = t2 * axis[:, 1] * axis[:, 2]
    t20 = t8 * axis[:, 0]
    t24 = axis[:, 2] * axis[:, 2]
    R = np.stack([t1 + t2 * t3, t7 - t9, t11 + t12, t7 + t9, t1 + t2 * t15, t19 - t20, t11 - t12, t19 + t20,
                  t1 + t2 * t24], axis=1)
    return np.reshape(R, (-1, 3, 3))
