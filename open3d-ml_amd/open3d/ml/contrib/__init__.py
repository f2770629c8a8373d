"""``open3d.ml.contrib`` — grid subsampling (``dataprocessing.py:32-49``, ``kpconv.py:2098-2155``) and the box IoUs of
the mAP metric (``ml3d/metrics/mAP.py:85-88``, ``ml3d/datasets/utils/operations.py:7``) on the MI355X.

numpy in, numpy out, like upstream's pybind functions.  Canonical orders: subsampled points come in
ascending voxel key; barycentres are float32 sums in original point order."""
import numpy as np
import torch

from ... import _product as _P


def _f32(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_P.device())


def _lab(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a).astype(np.int32).reshape(-1)).to(_P.device())


def subsample(points, features=None, classes=None, sampleDl=0.1, verbose=0):
    """-> points | (points, features) | (points, classes) | (points, features, classes)"""
    r = _P.ops().subsample(_f32(points), _f32(features), _lab(classes), float(sampleDl))
    if isinstance(r, tuple):
        return tuple(t.cpu().numpy() for t in r)
    return r.cpu().numpy()


def subsample_batch(points, batches_len, features=None, classes=None, sampleDl=0.1, method="barycenters", max_p=0,
                    verbose=0):
    """-> (points, lengths[, features][, classes]); lengths int32 like upstream"""
    if method != "barycenters":
        raise NotImplementedError("subsample_batch: only method='barycenters' (the one the reference uses)")
    lens = np.asarray(batches_len).astype(np.int64).reshape(-1)
    r = _P.ops().subsample_batch(_f32(points), lens.tolist(), _f32(features), _lab(classes), float(sampleDl), int(max_p))
    out = [t.cpu().numpy() for t in r]
    out[1] = out[1].astype(np.int32)
    return tuple(out)


def _boxes(a, cols):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols)
    return torch.from_numpy(a).to(_P.device())


def iou_bev_cuda(boxes_a, boxes_b):
    """rotated bird's-eye-view IoU of every pair: boxes [N, 5] / [M, 5] = (x, z, w, l, yaw) -> float32 [N, M]
    (ml3d/metrics/mAP.py:85: ``iou_bev(pred[:, [0, 2, 3, 5, 6]], target[:, [0, 2, 3, 5, 6]])``)"""
    return _P.ops().iou_bev(_boxes(boxes_a, 5), _boxes(boxes_b, 5)).cpu().numpy()


def iou_3d_cuda(boxes_a, boxes_b):
    """3-D IoU of every pair: boxes [N, 7] / [M, 7] = (x, y, z, w, h, l, yaw), y = bottom of the box (camera frame)
    -> float32 [N, M]  (ml3d/metrics/mAP.py:87: ``iou_3d(pred[:, :7], target[:, :7])``)"""
    return _P.ops().iou_3d(_boxes(boxes_a, 7), _boxes(boxes_b, 7)).cpu().numpy()


# there is one implementation, on the device; the reference picks the *_cpu names when open3d.core.cuda.device_count()
# is 0 (ml3d/metrics/__init__.py:3-9) and operations.py:7 imports iou_bev_cpu unconditionally
iou_bev_cpu = iou_bev_cuda
iou_3d_cpu = iou_3d_cuda
