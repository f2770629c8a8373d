"""Voxelize and grid subsampling + the per-item rotations of the pooling grids."""
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

def _host3(x):
    t = torch.as_tensor(x, dtype=torch.float32).detach().cpu().contiguous().reshape(-1)
    if t.numel() != 3:
        raise RuntimeError("voxelize: voxel_size / range tensors must have 3 elements")
    return t


def voxelize(points, row_splits, voxel_size, points_range_min, points_range_max,
             max_points_per_voxel=2 ** 62, max_voxels=2 ** 62):
    """``open3d.ml.torch.ops.voxelize`` (ml3d/torch/models/point_pillars.py:354-357).  ``points`` may be the
    strided view ``points[:, :3]`` of an [N, C] tensor (no copy).  voxel_size / range_* are CPU tensors as in
    the reference (point_pillars.py:317-320)."""
    lib = _abi.get()
    _need_gpu(points)
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype != torch.float32:
        raise RuntimeError("voxelize: points must be float32 [N, 3]")
    if points.stride(1) != 1:
        points = points.contiguous()
    stride = points.stride(0) if points.shape[0] > 1 else 3
    dev = points.device
    n = points.shape[0]
    rs = _splits(row_splits, n, dev)
    B = rs.numel() - 1
    vs, mn, mx = _host3(voxel_size), _host3(points_range_min), _host3(points_range_max)
    mp, mv = int(min(max_points_per_voxel, 2 ** 62)), int(min(max_voxels, 2 ** 62))
    wsb = lib.ml3d_voxelize_workspace_bytes(n, B)
    ws = _ws(wsb, dev)
    bs = torch.empty(B + 1, dtype=torch.int64, device=dev)
    stats = torch.empty(2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_voxelize_count(points.data_ptr(), stride, rs.data_ptr(), B, n, vs.data_ptr(), mn.data_ptr(),
                                     mx.data_ptr(), mp, mv, bs.data_ptr(), stats.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_voxelize_count")
        M, K = (int(x) for x in stats.tolist())
        coords = torch.empty((M, 3), dtype=torch.int32, device=dev)
        pidx = torch.empty(K, dtype=torch.int64, device=dev)
        prs = torch.empty(M + 1, dtype=torch.int64, device=dev)
        rc = lib.ml3d_voxelize_fill(B, n, vs.data_ptr(), mn.data_ptr(), mx.data_ptr(), mp, mv, bs.data_ptr(),
                                    coords.data_ptr(), pidx.data_ptr(), prs.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_voxelize_fill")
    return VoxelizeResult(coords, pidx, prs, bs)


def subsample_batch(points, batches_len, features=None, classes=None, sampleDl=0.1, max_p=0, verbose=0):
    """``open3d.ml.contrib.subsample_batch`` (ml3d/torch/models/kpconv.py:2098-2155) for CUDA tensors:
    returns (points, lengths[, features][, classes]) — voxel barycentres per batch item, ascending voxel key."""
    lib = _abi.get()
    _need_gpu(points, features, classes)
    points = points.contiguous().float()
    dev = points.device
    n = points.shape[0]
    rs, total = _splits_of_lengths(batches_len, dev)
    if total != n:
        raise RuntimeError("subsample_batch: batches_len does not sum to the number of points")
    B = rs.numel() - 1
    feats = None if features is None else features.contiguous().float()
    labs = None if classes is None else classes.contiguous().to(torch.int32).reshape(-1)
    wsb = lib.ml3d_subsample_workspace_bytes(n, B)
    ws = _ws(wsb, dev)
    out_len = torch.empty(B, dtype=torch.int64, device=dev)
    stats = torch.empty(2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_subsample_count(points.data_ptr(), rs.data_ptr(), B, n, float(sampleDl), out_len.data_ptr(),
                                      stats.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_subsample_count")
        M, err = (int(x) for x in stats.tolist())
        if err:
            raise RuntimeError("subsample: a batch item spans >= 2^40 voxels at this sampleDl (unsupported)")
        fd = 0 if feats is None else feats.shape[1]
        op = torch.empty((M, 3), dtype=torch.float32, device=dev)
        of = None if feats is None else torch.empty((M, fd), dtype=torch.float32, device=dev)
        ol = None if labs is None else torch.empty(M, dtype=torch.int32, device=dev)
        rc = lib.ml3d_subsample_fill(points.data_ptr(), None if feats is None else feats.data_ptr(), fd,
                                     None if labs is None else labs.data_ptr(), B, n, op.data_ptr(),
                                     None if of is None else of.data_ptr(), None if ol is None else ol.data_ptr(),
                                     ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_subsample_fill")
    if max_p and max_p > 0:     # kpconv.py: keep at most max_p points per batch item
        keep = []
        o = 0
        ln = out_len.tolist()
        for b in range(B):
            keep.append(torch.arange(o, o + min(ln[b], int(max_p)), device=dev))
            o += ln[b]
        keep = torch.cat(keep) if keep else torch.empty(0, dtype=torch.int64, device=dev)
        op = op[keep]
        of = None if of is None else of[keep]
        ol = None if ol is None else ol[keep]
        out_len = torch.clamp(out_len, max=int(max_p))
    out = [op, out_len.to(torch.int32)]
    if of is not None:
        out.append(of)
    if ol is not None:
        out.append(ol)
    return tuple(out)


def subsample(points, features=None, classes=None, sampleDl=0.1, verbose=0):
    """``open3d.ml.contrib.subsample`` (ml3d/datasets/utils/dataprocessing.py:32-49)."""
    r = subsample_batch(points, [points.shape[0]], features, classes, sampleDl)
    r = (r[0],) + tuple(r[2:])
    return r[0] if len(r) == 1 else r


def rotate_points(points, lengths_or_splits, rotations, transpose=False, is_splits=False):
    """Per-item rotation around the pooling grid of ``batch_grid_subsampling`` (kpconv.py:2086-2110)."""
    lib = _abi.get()
    _need_gpu(points, rotations)
    points = points.contiguous().float()
    dev = points.device
    if is_splits:
        rs = lengths_or_splits.to(device=dev, dtype=torch.int64).contiguous()
    else:
        rs = _splits_of_lengths(lengths_or_splits, dev)[0]
    R = rotations.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty_like(points)
    with torch.cuda.device(dev):
        rc = lib.ml3d_rotate_points(points.data_ptr(), rs.data_ptr(), rs.numel() - 1, points.shape[0], R.data_ptr(),
                                    1 if transpose else 0, out.data_ptr(), _stream())
    _abi.check(rc, "ml3d_rotate_points")
    return out


_FORCE_SORTED = False        # tests: True keeps every grid subsampling on the sort-based op (ml3d_subsample_count / _fill)


class _SubsamplePlan:
    """``batch_grid_subsampling`` (points only) in two halves, so that its one host read-back (the number of pooled points, needed
    to allocate them) can be shared with other pending sizes: ``stats`` int64 [2] = (pooled points, overflow flag) and
    ``out_len`` int64 [B] stay on the device until ``resolve``; ``fill`` writes the pooled points."""

    def __init__(self, points, batches_len, sampleDl, rotations=None):
        lib = _abi.get()
        _need_gpu(points)
        self.rotations = rotations
        pts = points.contiguous().float()
        self.src = pts if rotations is None else rotate_points(pts, batches_len, rotations)
        dev = self.src.device
        self.n = self.src.shape[0]
        self.rs, total = _splits_of_lengths(batches_len, dev)
        if total != self.n:
            raise RuntimeError("subsample_batch: batches_len does not sum to the number of points")
        self.B = self.rs.numel() - 1
        self.dl = float(sampleDl)
        self.out_len = torch.empty(self.B, dtype=torch.int64, device=dev)
        self.stats = torch.empty(2, dtype=torch.int64, device=dev)
        self.M = None
        # one workgroup per item, everything in LDS, two launches per call (ml3d_subsample_items_*): whenever every item is small
        # enough -- the KPConv batch build's case (spheres of <= 10 000 points); whole clouds take the sort-based op
        lens = batches_len.tolist() if torch.is_tensor(batches_len) else list(batches_len)
        self.max_item = max([int(v) for v in lens] or [0])
        self.items = (not _FORCE_SORTED) and self.B <= 65535 and self.max_item <= int(lib.ml3d_subsample_items_max_points())
        if self.items:
            with torch.cuda.device(dev):
                rc = lib.ml3d_subsample_items_count(self.src.data_ptr(), self.rs.data_ptr(), self.B, self.n, self.dl,
                                                    self.max_item, self.out_len.data_ptr(), self.stats.data_ptr(), _stream())
            _abi.check(rc, "ml3d_subsample_items_count")
        else:
            self._count_sorted()

    def _count_sorted(self):
        lib = _abi.get()
        dev = self.src.device
        self.items = False
        self.wsb = lib.ml3d_subsample_workspace_bytes(self.n, self.B)
        self.ws = _ws(self.wsb, dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_subsample_count(self.src.data_ptr(), self.rs.data_ptr(), self.B, self.n, self.dl,
                                          self.out_len.data_ptr(), self.stats.data_ptr(), self.ws.data_ptr(), self.wsb, _stream())
        _abi.check(rc, "ml3d_subsample_count")

    def resolve(self, values=None):
        if self.M is None:
            self.M, err = (int(x) for x in (self.stats.tolist() if values is None else values))
            if err == 2 and self.items:          # an item's grid has more cells than the per-item kernel's bitmap: the sort-based op
                self._count_sorted()
                self.M, err = (int(x) for x in self.stats.tolist())
            if err:
                raise RuntimeError("subsample: a batch item spans >= 2^40 voxels at this sampleDl (unsupported)")
        return self

    def fill(self):
        """-> (pooled points [M, 3], pooled lengths int32 [B] on the device)"""
        lib = _abi.get()
        self.resolve()
        dev = self.src.device
        op = torch.empty((self.M, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if self.items:
                rc = lib.ml3d_subsample_items_fill(self.src.data_ptr(), self.rs.data_ptr(), self.B, self.n, self.dl, self.max_item,
                                                   self.out_len.data_ptr(), op.data_ptr(), _stream())
            else:
                rc = lib.ml3d_subsample_fill(self.src.data_ptr(), None, 0, None, self.B, self.n, op.data_ptr(), None, None,
                                             self.ws.data_ptr(), self.wsb, _stream())
        _abi.check(rc, "ml3d_subsample_fill")
        if self.rotations is not None:
            # the pooled lengths stay on the device: their row splits are built there (no read-back for the rotation back)
            rs = torch.zeros(self.B + 1, dtype=torch.int64, device=dev)
            torch.cumsum(self.out_len, 0, out=rs[1:])
            op = rotate_points(op, rs, self.rotations, transpose=True, is_splits=True)
        return op, self.out_len.to(torch.int32)


def grid_subsampling_plan(points, batches_len, sampleDl=0.1, rotations=None):
    """Deferred ``batch_grid_subsampling``: counting enqueued, sizes not read yet (``_SubsamplePlan``)."""
    return _SubsamplePlan(points, batches_len, sampleDl, rotations)


def batch_grid_subsampling(points, batches_len, sampleDl=0.1, rotations=None):
    """``batch_grid_subsampling`` (ml3d/torch/models/kpconv.py:2037-2111, points only) on the GPU.
    ``rotations`` = float32 [B,3,3] grid orientations (what ``random_grid_orient`` draws) or None."""
    return _SubsamplePlan(points, batches_len, sampleDl, rotations).fill()
