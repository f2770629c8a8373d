"""Batchers of the inference data path — host-side mirror of ``ml3d/torch/dataloaders/{default_batcher,concat_batcher}.py``
for the three hot-path models, for use WITHOUT an Open3D-ML checkout (with one, the reference's own batchers drive the
native model classes unchanged: tools/ref_pipelines.py).  Same class names, constructor arguments and ``collate_fn`` results:

* ``DefaultBatcher`` (RandLA-Net): lists of per-level arrays -> lists of stacked tensors.  Index lists that
  ``RandLANet.transform`` left on the device are stacked there (no host round trip).
* ``ConcatBatcher(device, 'KPFCNN')``: the transformed spheres of the batch items concatenated up to ``batch_limit``
  (concat_batcher.py:41-60), input features chosen by ``in_features_dim`` (:73-104), and the per-layer neighbour / pooling /
  upsampling matrices of ``segmentation_inputs`` (:186-305) built ON THE GPU (``models.kpconv.KPConvBatch``: fixed-radius
  search + grid subsample kernels) instead of the reference's CPU loop.
* ``ConcatBatcher(device, 'PointPillars')``: ``ObjectDetectBatch`` (:487-537).
"""
import numpy as np
import torch


def _stack(values):
    if len(values) == 1 and isinstance(values[0], torch.Tensor):
        # test_batch_size 1 (every in-scope YAML's inference batch): a batch of ONE tensor is that tensor with a leading axis --
        # a view, where torch.stack launches a copy kernel per entry (19 per RandLA-Net item: every level's points, neighbour,
        # pooling and interpolation lists).  The transforms hand out freshly allocated tensors, so nothing else writes to them.
        out = values[0].unsqueeze(0)
    else:
        out = torch.stack([v if isinstance(v, torch.Tensor) else torch.as_tensor(v) for v in values], 0)
    if all(getattr(v, '_ml3d_prefix_of_neighbors', False) for v in values):
        out._ml3d_prefix_of_neighbors = True          # RandLANet.transform's mark on sub_idx (models/randlanet.py, _mark_prefix)
    if len(values) == 1 and getattr(values[0], '_ml3d_arena', None) is not None:
        out._ml3d_arena = values[0]._ml3d_arena       # a batch of ONE device-loop patch stays recognisable to RandLANet.forward
    return out


def default_collate(batch):
    """default_batcher.py:34-85 for what the model transforms produce (tensors, arrays, numbers, strings, dicts, lists)."""
    elem = batch[0]
    if isinstance(elem, (torch.Tensor, np.ndarray)):
        return _stack(batch)
    if isinstance(elem, float):
        return torch.tensor(batch, dtype=torch.float64)
    if isinstance(elem, (int, np.integer)):
        return torch.tensor(batch)
    if isinstance(elem, (str, bytes)):
        return batch
    if isinstance(elem, dict):
        return {k: default_collate([d[k] for d in batch]) for k in elem}
    if isinstance(elem, (list, tuple)):
        if not all(len(e) == len(elem) for e in batch):
            raise RuntimeError('each element in list of batch should be of equal size')
        return [default_collate(list(samples)) for samples in zip(*batch)]
    raise TypeError("default_collate: batch must contain tensors, numpy arrays, numbers, dicts or lists; found %s" % type(elem))


class DefaultBatcher(object):

    def collate_fn(self, batch):
        return default_collate(batch)


def kpconv_input_features(points, columns, in_features_dim):
    """concat_batcher.py:73-104: the network's input features from the per-point columns [xyz | feat] of ``transform``."""
    ones = np.ones_like(points[:, :1], dtype=np.float32)
    d = int(in_features_dim)
    if d == 1:
        return ones
    if d == 2:
        return np.hstack((ones, columns[:, 2:3]))                      # + height
    if d == 3:
        assert columns.shape[1] > 3, "feat from dataset can not be None or try to set in_features_dim = 1, 2, 4"
        return np.hstack((ones, columns[:, 2:4]))                      # + height, reflectance
    if d == 4:
        return np.hstack((ones, columns[:, :3]))                       # + all coordinates
    if d == 5:
        assert columns.shape[1] >= 6, "feat from dataset should have at least 3 dims, or try to set in_features_dim = 1, 2, 4"
        return np.hstack((ones, columns[:, 2:6]))                      # + height, colour
    if d >= 6:
        assert columns.shape[1] > 3, "feat from dataset can not be None or try to set in_features_dim = 1, 2, 4"
        return np.hstack((ones, columns))
    raise ValueError('in_features_dim should be >= 0')


class ObjectDetectBatch:
    """concat_batcher.py:487-537."""

    def __init__(self, batches):
        self.point, self.labels, self.bboxes, self.bbox_objs, self.calib, self.attr = [], [], [], [], [], []
        for batch in batches:
            data = batch['data']
            self.point.append(torch.as_tensor(data['point'], dtype=torch.float32))
            self.labels.append(torch.as_tensor(data['labels'], dtype=torch.int64) if 'labels' in data else None)
            self.bboxes.append(torch.as_tensor(data['bboxes'], dtype=torch.float32) if len(data.get('bboxes', [])) > 0
                               else torch.zeros((0, 7)))
            self.bbox_objs.append(data.get('bbox_objs'))
            self.calib.append(data.get('calib'))

    def pin_memory(self):
        self.point = [p.pin_memory() for p in self.point]
        return self

    def to(self, device):
        self.point = [p.to(device, non_blocking=True) for p in self.point]
        self.labels = [None if l is None else l.to(device) for l in self.labels]
        self.bboxes = [None if b is None else b.to(device) for b in self.bboxes]
        return self


class ConcatBatcher(object):

    def __init__(self, device, model='KPConv'):
        self.device = device
        self.model = model

    def collate_fn(self, batches):
        if self.model in ("KPConv", "KPFCNN"):
            return {'data': self._kpconv(batches), 'attr': []}
        if self.model in ("PointPillars", "PointRCNN"):
            return ObjectDetectBatch(batches)
        raise Exception("ConcatBatcher (MI355X build): model '%s' is outside the hot path (KPFCNN, PointPillars)" % self.model)

    def _kpconv(self, batches):
        from .models.kpconv import KPConvBatch
        cfg = batches[0]['data']['cfg']
        keys = ('p_list', 'f_list', 'l_list', 'p0_list', 's_list', 'R_list', 'r_inds_list', 'r_mask_list', 'val_labels_list')
        acc = {k: [] for k in keys}
        n = 0
        for item in batches:                                  # whole items, until batch_limit points (concat_batcher.py:41-60)
            data = item['data']
            n += sum(p.shape[0] for p in data['p_list'])
            if n > int(cfg.batch_limit):
                break
            for k in keys:
                acc[k] += data[k]
        pts = np.concatenate(acc['p_list'], axis=0).astype(np.float32)
        columns = np.concatenate(acc['f_list'], axis=0)
        lens = [int(p.shape[0]) for p in acc['p_list']]
        b = KPConvBatch(pts, lens, cfg, features=kpconv_input_features(pts, columns, cfg.in_features_dim).astype(np.float32),
                        device=self.device)
        b.labels = torch.from_numpy(np.concatenate([np.atleast_1d(l) for l in acc['l_list']], axis=0).astype(np.int64))
        b.scales = torch.from_numpy(np.array(acc['s_list'], dtype=np.float32))
        b.rots = torch.from_numpy(np.stack(acc['R_list'], axis=0))
        b.frame_inds = torch.from_numpy(np.array([], dtype=np.int32))
        b.frame_centers = torch.from_numpy(np.stack(acc['p0_list'], axis=0))
        b.reproj_inds, b.reproj_masks, b.val_labels = acc['r_inds_list'], acc['r_mask_list'], acc['val_labels_list']
        return b
