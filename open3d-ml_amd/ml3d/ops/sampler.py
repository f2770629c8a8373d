"""Patch sampler query, label argmax and the float16 vote update."""
import ctypes as C

import numpy as np
import torch

from .. import _abi
from . import _gates
from ._gates import KnnResult, RadiusResult, VoxelizeResult, _splits, _splits_of_lengths


def _stream():
    return _gates._stream()


def _need_gpu(*tensors):
    return _gates._need_gpu(*tensors)


def _ws(nbytes, device):
    return _gates._ws(nbytes, device)

def nearest_to_center(points, center, k, return_distances=False):
    """The ``k`` points nearest to ``center`` — the ``search_tree.query(center_point, k=num_points)`` of
    SemSegSpatiallyRegularSampler (ml3d/datasets/samplers/semseg_spatially_regular.py:90-91) in the order of the
    reference's sklearn ``KDTree``: ascending FLOAT64 reduced distance (ties: ascending index).  int32 indices [k]
    and, on request, the float64 reduced distances."""
    lib = _abi.get()
    _need_gpu(points)
    points = points.contiguous().float()
    n = points.shape[0]
    dev = points.device
    c = torch.as_tensor(center, dtype=torch.float32).detach().cpu().reshape(-1).contiguous()
    if c.numel() != 3 or not (0 <= int(k) <= n):
        raise RuntimeError("nearest_to_center: center must have 3 elements and 0 <= k <= n_points")
    idx = torch.empty(int(k), dtype=torch.int32, device=dev)
    d2 = torch.empty(int(k), dtype=torch.float64, device=dev) if return_distances else None
    wsb = lib.ml3d_nearest_to_center_workspace_bytes(n)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_nearest_to_center(points.data_ptr(), n, c.data_ptr(), int(k), idx.data_ptr(),
                                        None if d2 is None else d2.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_nearest_to_center")
    return (idx, d2) if return_distances else idx


def device_patch(points, possibility, center_index, perm, k, recenter_dims=(), extra=None, feat_bias=0.0, feat_scale=1.0, out=None):
    """One step of the spatially regular patch loop (semseg_spatially_regular.py:64-111 + randlanet.py:185-212) with nothing
    read back: ``points`` float32 [n, 3], ``possibility`` float64 [n] (bumped IN PLACE), ``center_index`` a 1-element DEVICE
    index tensor (the argmin of the possibilities), ``perm`` the host-drawn shuffle of 0..k-1 as a device int32 tensor,
    ``extra`` optional float32 [n, c] per-point features.  Returns (patch points [k, 3] recentred on ``recenter_dims``,
    features [k, 3 + c], selected cloud indices int32 [k]) -- the arithmetic and its ORDER are numpy's (float32 squared
    distances left to right, sequential float32 column sums for the mean): the patch feeds an exact neighbour search.
    ``out``: optional preallocated (points [k, 3] float32, features [k, 3 + c] float32, indices [k] int32) to write into."""
    lib = _abi.get()
    _need_gpu(points, possibility, center_index, perm)
    dev = points.device
    n = points.shape[0]
    k = int(k)
    if points.dtype != torch.float32 or not points.is_contiguous() or possibility.dtype != torch.float64 or \
            not possibility.is_contiguous() or possibility.numel() != n or perm.dtype != torch.int32 or perm.numel() != k or k > n:
        raise RuntimeError("device_patch: float32 [n, 3] points, float64 [n] possibilities, int32 [k] permutation, k <= n")
    center = points[center_index.reshape(1)].reshape(3).contiguous()          # (a device gather: the centre is never on the host)
    cand = torch.empty(k, dtype=torch.int32, device=dev)
    wsb = lib.ml3d_nearest_to_center_workspace_bytes(n)
    ws = _ws(wsb, dev)
    n_extra = 0 if extra is None else int(extra.shape[1])
    if out is None:
        pts = torch.empty((k, 3), dtype=torch.float32, device=dev)
        sel = torch.empty(k, dtype=torch.int32, device=dev)
        feats = torch.empty((k, 3 + n_extra), dtype=torch.float32, device=dev)
    else:
        pts, feats, sel = out
        if tuple(pts.shape) != (k, 3) or tuple(feats.shape) != (k, 3 + n_extra) or sel.numel() != k or pts.dtype != torch.float32 or \
                feats.dtype != torch.float32 or sel.dtype != torch.int32 or not (pts.is_contiguous() and feats.is_contiguous() and sel.is_contiguous()):
            raise RuntimeError("device_patch: out = (float32 [k, 3], float32 [k, 3 + c], int32 [k]), contiguous")
    scratch = torch.empty(4 * k + 64, dtype=torch.uint8, device=dev)
    ex = None
    mask = 0
    for d in recenter_dims:
        mask |= 1 << int(d)
    with torch.cuda.device(dev):
        rc = lib.ml3d_nearest_to_center_dev(points.data_ptr(), n, center.data_ptr(), k, cand.data_ptr(), None, ws.data_ptr(), wsb,
                                            _stream())
        _abi.check(rc, "ml3d_nearest_to_center_dev")
        rc = lib.ml3d_patch_crop(points.data_ptr(), n, cand.data_ptr(), perm.data_ptr(), center.data_ptr(), k, pts.data_ptr(),
                                 sel.data_ptr(), possibility.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream())
        _abi.check(rc, "ml3d_patch_crop")
        if n_extra:
            ex = extra[sel.long()].contiguous()
        rc = lib.ml3d_patch_recenter(pts.data_ptr(), k, mask, None if ex is None else ex.data_ptr(), n_extra, float(feat_bias),
                                     float(feat_scale), feats.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream())
        _abi.check(rc, "ml3d_patch_recenter")
    return pts, feats, sel


def argmax_labels(scores, out=None):
    """uint8 labels [...] = argmax over the last axis of float32 ``scores`` [..., C <= 256] (first maximum, like
    torch.argmax) -- one pass over the scores instead of torch's generic reduction + dtype cast."""
    lib = _abi.get()
    _need_gpu(scores)
    if scores.dtype != torch.float32 or not scores.is_contiguous() or scores.shape[-1] > 256:
        raise RuntimeError("argmax_labels: contiguous float32 scores with at most 256 classes")
    n = scores.numel() // scores.shape[-1]
    if out is None:
        out = torch.empty(scores.shape[:-1], dtype=torch.uint8, device=scores.device)
    elif out.dtype != torch.uint8 or out.numel() != n or not out.is_contiguous():
        raise RuntimeError("argmax_labels: out must be a contiguous uint8 tensor with one entry per point")
    with torch.cuda.device(scores.device):
        rc = lib.ml3d_argmax_labels(scores.data_ptr(), n, int(scores.shape[-1]), out.data_ptr(), _stream())
    _abi.check(rc, "ml3d_argmax_labels")
    return out


def vote_update(test_probs, point_inds, logits, smooth=0.95):
    """In place: ``test_probs[inds] = smooth * test_probs[inds] + (1 - smooth) * softmax(logits)`` on the float16
    vote accumulator [N_cloud, classes] (ml3d/torch/models/randlanet.py:420-421, 457-462).
    ONE batch item per call: every wave does an unsynchronised read-modify-write of its point's row, so the indices of a
    call MUST be unique.  A patch padded with repeated points (cloud smaller than num_points) is de-duplicated by the caller,
    keeping each point's last occurrence = numpy's fancy-assignment result (``RandLANet.update_probs``).  Patches of a batch
    that share points are applied by calling
    this once per item, in order, on one stream -- what ``RandLANet.update_probs`` / ``KPFCNN.update_probs`` do and what
    the reference's sequential loop (randlanet.py:455-463) means."""
    lib = _abi.get()
    _need_gpu(test_probs, point_inds, logits)
    if test_probs.dtype != torch.float16 or not test_probs.is_contiguous() or test_probs.dim() != 2:
        raise RuntimeError("vote_update: test_probs must be a contiguous float16 [N, classes] tensor")
    C_ = test_probs.shape[1]
    lg = logits.reshape(-1, C_).contiguous().float()
    inds = point_inds.reshape(-1).to(torch.int32).contiguous()
    if inds.numel() != lg.shape[0]:
        raise RuntimeError("vote_update: one index per logits row")
    with torch.cuda.device(test_probs.device):
        rc = lib.ml3d_vote_update(lg.data_ptr(), inds.data_ptr(), lg.shape[0], C_, float(smooth), test_probs.data_ptr(),
                                  test_probs.shape[0], _stream())
    _abi.check(rc, "ml3d_vote_update")
    return test_probs
