"""``ml3d.ops`` — torch front end of libml3d_hip.so (MI355X / gfx950 only).

Mirrors the call surface the reference gets from ``open3d.ml.torch.ops`` /
``open3d.core.nns`` for the inference hot path.  Every function
requires CUDA(HIP) tensors and raises ``RuntimeError`` otherwise: this package has no
CPU path (the CPU oracle lives under /oracle and is test infrastructure only).
All kernels are enqueued on torch's CURRENT stream.
"""
import ctypes as C

import numpy as np
from collections import namedtuple

import torch

from .. import _abi

KnnResult = namedtuple("KnnResult", ["neighbors_index", "neighbors_distance"])


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("ml3d.ops: HIP kernels need tensors on an MI355X device (got %s); "
                               "there is no CPU fallback" % (getattr(t, "device", type(t)),))


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def knn_search(points, queries, k, points_row_splits=None, queries_row_splits=None,
               return_distances=False, index_local=False):
    """Exact k-NN, ascending (d2, index).  Replaces ``NearestNeighborSearch.knn_search``
    (ml3d/datasets/utils/dataprocessing.py:99-103) and ``open3d.ml.torch.ops.knn_search``.

    points [Ns,3] f32, queries [Nq,3] f32 (same tensor object -> self query).
    Returns int32 indices [Nq, k] (-1 padded when an item has < k points) and, on request,
    squared distances."""
    lib = _abi.get()
    _need_gpu(points, queries)
    points = points.contiguous().float()
    same = queries is points or (queries.data_ptr() == points.data_ptr() and queries.shape == points.shape)
    queries = points if same else queries.contiguous().float()
    dev = points.device
    ns, nq = points.shape[0], queries.shape[0]
    if points_row_splits is None:
        points_row_splits = torch.tensor([0, ns], dtype=torch.int64, device=dev)
    if queries_row_splits is None:
        queries_row_splits = points_row_splits if same else torch.tensor([0, nq], dtype=torch.int64, device=dev)
    prs = points_row_splits.to(device=dev, dtype=torch.int64).contiguous()
    qrs = prs if (same and queries_row_splits is points_row_splits) else \
        queries_row_splits.to(device=dev, dtype=torch.int64).contiguous()
    batch = prs.numel() - 1
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d2 = torch.empty((nq, k), dtype=torch.float32, device=dev) if return_distances else None
    wsb = lib.ml3d_knn_workspace_bytes(ns, nq, batch)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_knn_search(points.data_ptr(), prs.data_ptr(), queries.data_ptr(), qrs.data_ptr(), batch,
                                 ns, nq, int(k), 1 if index_local else 0, idx.data_ptr(),
                                 d2.data_ptr() if d2 is not None else None, ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_knn_search")
    return KnnResult(idx, d2 if d2 is not None else torch.empty(0, device=dev))


def pyramid_sizes(n0, ratios):
    n = [int(n0)]
    for r in ratios:
        n.append(n[-1] // int(r))
    return n


def randla_knn_pyramid(points, ratios, k, out=None, workspace=None, tile_order=None):
    """All neighbour searches of ``RandLANet.transform`` (ml3d/torch/models/randlanet.py:218-229)
    for a batch [B, N, 3] in one call.  Returns (neighbor_idx[l] [B,n_l,k], interp_idx[l] [B,n_l,1]),
    int32, item-local.  ``sub_idx[l]`` of the reference is ``neighbor_idx[l][:, :n_{l+1}]``.
    ``tile_order``: optional list of int32 [B * n_l] tensors (one per level) that receive the levels' cell-sorted
    point order for ``randla_forward(..., tile_order=...)``."""
    lib = _abi.get()
    _need_gpu(points)
    if points.dim() != 3 or points.shape[2] != 3 or points.dtype != torch.float32 or not points.is_contiguous():
        raise RuntimeError("randla_knn_pyramid: points must be a contiguous float32 [B, N, 3] tensor")
    B, n0, _ = points.shape
    L = len(ratios)
    n = pyramid_sizes(n0, ratios)
    dev = points.device
    if out is None:
        nbr = [torch.empty((B, n[l], k), dtype=torch.int32, device=dev) for l in range(L)]
        itp = [torch.empty((B, n[l], 1), dtype=torch.int32, device=dev) for l in range(L)]
    else:
        nbr, itp = out
    r = (C.c_int32 * L)(*[int(x) for x in ratios])
    wsb = lib.ml3d_randla_pyramid_workspace_bytes(B, n0, L, r)
    if wsb == 0:
        raise RuntimeError("randla_knn_pyramid: invalid pyramid description")
    ws = workspace if workspace is not None else _ws(wsb, dev)
    if ws.numel() < wsb:
        raise RuntimeError("randla_knn_pyramid: workspace too small")
    t_n = _abi.ptr_table([t.data_ptr() for t in nbr])
    t_i = _abi.ptr_table([t.data_ptr() for t in itp])
    with torch.cuda.device(dev):
        if tile_order is not None:
            # one entry per level; None = no order wanted for that level (the engine orders only the finest levels)
            if len(tile_order) != L or any(t is not None and (t.dtype != torch.int32 or t.numel() != B * n[l] or
                                                              not t.is_contiguous()) for l, t in enumerate(tile_order)):
                raise RuntimeError("randla_knn_pyramid: tile_order must be one contiguous int32 [B * n_l] tensor (or None) "
                                   "per level")
            t_o = _abi.ptr_table([0 if t is None else t.data_ptr() for t in tile_order])
            rc = lib.ml3d_randla_knn_pyramid_ordered(points.data_ptr(), B, n0, L, r, int(k), t_n, t_i, t_o, ws.data_ptr(),
                                                     ws.numel(), _stream(), None)
        else:
            rc = lib.ml3d_randla_knn_pyramid(points.data_ptr(), B, n0, L, r, int(k), t_n, t_i, ws.data_ptr(),
                                             ws.numel(), _stream())
    _abi.check(rc, "ml3d_randla_knn_pyramid")
    return nbr, itp


def randla_forward(desc, params, features, points, neighbor_idx, interp_idx, out=None, workspace=None, tile_order=None):
    """Fused RandLA-Net forward (ml3d/torch/models/randlanet.py:241-298) -> scores [B, N, classes].
    ``tile_order`` (optional, from ``randla_knn_pyramid``): walk each level's attention tiles in that point order
    (same result, better cache locality of the neighbour gathers)."""
    lib = _abi.get()
    _need_gpu(params, features, points, *neighbor_idx, *interp_idx)
    dev = points.device
    B, n0 = int(desc.batch), int(desc.num_points)
    if tuple(points.shape) != (B, n0, 3) or tuple(features.shape) != (B, n0, desc.in_channels):
        raise RuntimeError("randla_forward: points/features shape does not match the descriptor")
    for t in (points, features):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("randla_forward: points/features must be contiguous float32")
    for t in list(neighbor_idx) + list(interp_idx):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("randla_forward: index tensors must be contiguous int32")
    # the C ABI receives bare pointers and derives every extent from the descriptor: the list lengths and the shape of
    # every level are checked HERE (a short list or a wrong K would be an out-of-bounds device read, not an error)
    L, K = int(desc.num_layers), int(desc.num_neighbors)
    sizes = pyramid_sizes(n0, [int(desc.sub_sampling_ratio[i]) for i in range(L)])
    if len(neighbor_idx) != L or len(interp_idx) != L:
        raise RuntimeError("randla_forward: need %d neighbour and %d interpolation index tensors (got %d / %d)"
                           % (L, L, len(neighbor_idx), len(interp_idx)))
    for l in range(L):
        if tuple(neighbor_idx[l].shape) != (B, sizes[l], K):
            raise RuntimeError("randla_forward: neighbor_idx[%d] must be [%d, %d, %d], got %s"
                               % (l, B, sizes[l], K, tuple(neighbor_idx[l].shape)))
        if tuple(interp_idx[l].shape) not in ((B, sizes[l], 1), (B, sizes[l])):
            raise RuntimeError("randla_forward: interp_idx[%d] must be [%d, %d, 1], got %s"
                               % (l, B, sizes[l], tuple(interp_idx[l].shape)))
        if tile_order is not None and (l >= len(tile_order) or
                                       (tile_order[l] is not None and tile_order[l].numel() != B * sizes[l])):
            raise RuntimeError("randla_forward: tile_order[%d] must hold %d rows (or be None)" % (l, B * sizes[l]))
    if out is None:
        out = torch.empty((B, n0, desc.num_classes), dtype=torch.float32, device=dev)
    wsb = lib.ml3d_randla_forward_workspace_bytes(C.byref(desc))
    ws = workspace if workspace is not None else _ws(wsb, dev)
    if ws.numel() < wsb:
        raise RuntimeError("randla_forward: workspace too small")
    t_n = _abi.ptr_table([t.data_ptr() for t in neighbor_idx])
    t_i = _abi.ptr_table([t.data_ptr() for t in interp_idx])
    with torch.cuda.device(dev):
        if tile_order is not None:
            for t in tile_order:
                if t is not None and (t.dtype != torch.int32 or not t.is_contiguous() or t.device != dev):
                    raise RuntimeError("randla_forward: tile_order tensors must be contiguous int32 on the same device")
            t_o = _abi.ptr_table([0 if t is None else t.data_ptr() for t in tile_order])
            rc = lib.ml3d_randla_forward_ordered(C.byref(desc), params.data_ptr(), features.data_ptr(), points.data_ptr(),
                                                 t_n, t_i, t_o, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(), None)
        else:
            rc = lib.ml3d_randla_forward(C.byref(desc), params.data_ptr(), features.data_ptr(), points.data_ptr(),
                                         t_n, t_i, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _abi.check(rc, "ml3d_randla_forward")
    return out


# ---------------------------------------------------------------------------------------------------
# fixed-radius search / ragged_to_dense / voxelize / subsample  
# ---------------------------------------------------------------------------------------------------
RadiusResult = namedtuple("RadiusResult", ["neighbors_index", "neighbors_row_splits", "neighbors_distance"])
VoxelizeResult = namedtuple("VoxelizeResult", ["voxel_coords", "voxel_point_indices", "voxel_point_row_splits",
                                               "voxel_batch_splits"])


# row splits of a list of per-item lengths on the device.  A KPConv batch build asks for the same few lists ~6 times each
# (conv / pool / upsample searches, subsampling and the two rotations of a layer): a small cache keyed by (lengths, device,
# stream) turns ~40 tiny host-to-device copies per batch into ~6.  (The stream is part of the key: a cached tensor is only
# handed to work enqueued on the stream its upload was ordered on.)
_SPLITS_CACHE = {}


def _splits_of_lengths(lengths, dev):
    if torch.is_tensor(lengths):
        lengths = lengths.tolist()
    key = (tuple(int(v) for v in lengths), str(dev), torch.cuda.current_stream(dev).cuda_stream)
    hit = _SPLITS_CACHE.get(key)
    if hit is None:
        host = np.zeros(len(key[0]) + 1, np.int64)
        np.cumsum(np.asarray(key[0], np.int64), out=host[1:])
        hit = (torch.from_numpy(host).to(dev), int(host[-1]))
        if len(_SPLITS_CACHE) >= 64:
            _SPLITS_CACHE.clear()
        _SPLITS_CACHE[key] = hit
    return hit


def _splits(rs, n, dev):
    if rs is None:
        return torch.tensor([0, int(n)], dtype=torch.int64, device=dev)
    return rs.to(device=dev, dtype=torch.int64).contiguous()


class _RadiusPlan:
    """Phase 1 of the fixed-radius search: grid + per-query counts (kept on the device)."""

    def __init__(self, points, queries, radius, points_row_splits, queries_row_splits, defer=False):
        """``defer=True`` leaves the two sizes on the device: read them with ``resolve()`` -- or with ONE host read-back for
        several plans through ``resolve_plans`` (the KPConv batch build has two independent searches per layer)."""
        lib = _abi.get()
        _need_gpu(points, queries)
        self.points = points.contiguous().float()
        self.queries = self.points if queries is points else queries.contiguous().float()
        dev = self.points.device
        self.ns, self.nq = self.points.shape[0], self.queries.shape[0]
        self.prs = _splits(points_row_splits, self.ns, dev)
        self.qrs = _splits(queries_row_splits, self.nq, dev)
        if self.prs.numel() != self.qrs.numel():
            raise RuntimeError("fixed_radius_search: points and queries must have the same batch size")
        self.batch = self.prs.numel() - 1
        self.radius = float(radius)
        self.row_splits = torch.empty(self.nq + 1, dtype=torch.int64, device=dev)
        self.stats = torch.empty(2, dtype=torch.int64, device=dev)
        # room for ~96 neighbours per query before the workspace has to grow for the spill area
        self.ws_total = 96 * self.nq
        self.wsb = lib.ml3d_radius_workspace_bytes(self.ns, self.nq, self.batch, self.ws_total)
        self.ws = _ws(self.wsb, dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_radius_count(self.points.data_ptr(), self.prs.data_ptr(), self.queries.data_ptr(),
                                       self.qrs.data_ptr(), self.batch, self.ns, self.nq, self.radius,
                                       self.row_splits.data_ptr(), self.stats.data_ptr(), self.ws.data_ptr(), self.wsb,
                                       _stream())
        _abi.check(rc, "ml3d_radius_count")
        self.total = self.longest = None
        if not defer:
            self.resolve()

    def resolve(self, values=None):
        """the one host sync of a search (the reference's ``.item()``, kpconv.py:2028) unless ``values`` are handed in"""
        if self.total is None:
            self.total, self.longest = (int(x) for x in (self.stats.tolist() if values is None else values))
            if self.longest < 0 or self.longest >= 2 ** 62:       # the library's overflow flag (int32 scan of the counts wrapped)
                raise RuntimeError("fixed_radius_search: 2^31 or more neighbours in one call (unsupported): split the batch")
        return self

    def fill(self, dense_cols=0, pad_value=0, index_local=False, return_distances=False):
        lib = _abi.get()
        dev = self.points.device
        # the workspace that carries the grid goes back in untouched; when the result is larger than the spill area it was
        # sized for, long rows sort in a separate buffer (no relocation of the grid -- ml3d_hip.h, ml3d_radius_fill)
        spill = _ws(8 * self.total + 8, dev) if self.total > self.ws_total else None
        shape = (self.nq, int(dense_cols)) if dense_cols > 0 else (self.total,)
        idx = torch.empty(shape, dtype=torch.int32, device=dev)
        d2 = torch.empty(shape, dtype=torch.float32, device=dev) if return_distances else None
        with torch.cuda.device(dev):
            rc = lib.ml3d_radius_fill(self.points.data_ptr(), self.prs.data_ptr(), self.queries.data_ptr(),
                                      self.qrs.data_ptr(), self.batch, self.ns, self.nq, self.radius,
                                      self.row_splits.data_ptr(), self.total, 1 if index_local else 0, int(dense_cols),
                                      int(pad_value), idx.data_ptr(), d2.data_ptr() if d2 is not None else None,
                                      self.ws.data_ptr(), self.wsb, spill.data_ptr() if spill is not None else None,
                                      8 * self.total + 8 if spill is not None else 0, _stream())
        _abi.check(rc, "ml3d_radius_fill")
        return idx, d2


def resolve_plans(*plans):
    """One read-back for the sizes of several deferred plans (anything with ``.stats`` [2] int64 and ``.resolve``)."""
    todo = [p for p in plans if p is not None and p.longest is None]
    if todo:
        vals = torch.cat([p.stats for p in todo]).tolist()
        for i, p in enumerate(todo):
            p.resolve(vals[2 * i:2 * i + 2])


def fixed_radius_search(points, queries, radius, points_row_splits=None, queries_row_splits=None,
                        return_distances=False):
    """Functional form of ``open3d.ml.torch.layers.FixedRadiusSearch`` (ml3d/torch/models/kpconv.py:2021-2026):
    ragged neighbours (d2 <= r^2), each row ascending (d2, index), GLOBAL int32 indices."""
    plan = _RadiusPlan(points, queries, radius, points_row_splits, queries_row_splits)
    idx, d2 = plan.fill(return_distances=return_distances)
    return RadiusResult(idx, plan.row_splits, d2 if d2 is not None else torch.empty(0, device=idx.device))


class _DenseRadiusPlan:
    """``batch_neighbors`` in ONE traversal (``ml3d_radius_dense_gather`` / ``_expand``): the search runs once and parks every
    row, sorted and with global indices, in a per-query stash of ``CAP`` entries inside the workspace; what the host must read
    before it can allocate the dense matrix is just ``stats`` = (overflow flag, longest row).  A row longer than ``CAP`` sends
    this one search through the two-phase ``_RadiusPlan`` instead (count -> read -> fill)."""
    CAP = 128

    def __init__(self, points, queries, radius, points_row_splits, queries_row_splits, grid_from=None):
        """``grid_from``: an earlier plan over the SAME support tensor, row splits and radius whose result has been taken
        (``fill_dense`` enqueued): its workspace -- grid included -- is searched again with these queries instead of building
        the grid a second time (the conv and the pool search of a KPConv layer, concat_batcher.py:234-262)."""
        lib = _abi.get()
        _need_gpu(points, queries)
        self.points = points.contiguous().float()
        self.queries = self.points if queries is points else queries.contiguous().float()
        dev = self.points.device
        self.ns, self.nq = self.points.shape[0], self.queries.shape[0]
        self.prs = _splits(points_row_splits, self.ns, dev)
        self.qrs = _splits(queries_row_splits, self.nq, dev)
        if self.prs.numel() != self.qrs.numel():
            raise RuntimeError("fixed_radius_search: points and queries must have the same batch size")
        self.batch = self.prs.numel() - 1
        self.radius = float(radius)
        self.stats = torch.empty(2, dtype=torch.int64, device=dev)
        self.wsb = lib.ml3d_radius_dense_workspace_bytes(self.ns, self.nq, self.batch, self.CAP)
        reuse = isinstance(grid_from, _DenseRadiusPlan) and grid_from.filled and grid_from.points is self.points and \
            grid_from.prs is self.prs and grid_from.radius == self.radius and grid_from.wsb >= self.wsb
        if reuse:
            self.ws, self.wsb = grid_from.ws, grid_from.wsb
        else:
            self.ws = _ws(self.wsb, dev)
        self.filled = False
        with torch.cuda.device(dev):
            rc = lib.ml3d_radius_dense_gather(self.points.data_ptr(), self.prs.data_ptr(), self.queries.data_ptr(),
                                              self.qrs.data_ptr(), self.batch, self.ns, self.nq, self.radius, self.CAP,
                                              1 if reuse else 0, self.stats.data_ptr(), self.ws.data_ptr(), self.wsb, _stream())
        _abi.check(rc, "ml3d_radius_dense_gather")
        self.total = self.longest = None          # (``total`` stays unknown: the dense result never needs it)
        self.fallback = None

    def resolve(self, values=None):
        if self.longest is None:
            overflow, longest = (int(x) for x in (self.stats.tolist() if values is None else values))
            if overflow:                          # some row is longer than the stash: the two-phase search for this one
                self.fallback = _RadiusPlan(self.points, self.queries, self.radius, self.prs, self.qrs)
                longest = self.fallback.longest
            self.longest, self.total = longest, -1
        return self

    def fill_dense(self, cols, pad_value):
        self.filled = True
        if self.fallback is not None:
            return self.fallback.fill(dense_cols=cols, pad_value=pad_value)[0]
        lib = _abi.get()
        dev = self.points.device
        idx = torch.empty((self.nq, int(cols)), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_radius_dense_expand(self.ns, self.nq, self.batch, self.CAP, int(cols), int(pad_value), idx.data_ptr(),
                                              self.ws.data_ptr(), self.wsb, _stream())
        _abi.check(rc, "ml3d_radius_dense_expand")
        return idx


def _one_pass_radius():
    # ML3D_RADIUS_ONE_PASS=0 (read once): the two-phase search for the dense result as well (A/B runs)
    global _ONE_PASS
    try:
        return _ONE_PASS
    except NameError:
        import os
        _ONE_PASS = os.environ.get("ML3D_RADIUS_ONE_PASS", "1") != "0"
        return _ONE_PASS


def radius_plan_dense(queries, supports, q_lengths, s_lengths, radius, grid_from=None):
    """Deferred first half of ``radius_neighbors_dense``: the search is enqueued, its sizes not read yet.  ``grid_from``: an
    already filled plan over the same supports and radius whose grid is reused."""
    dev = supports.device
    prs, qrs = _splits_of_lengths(s_lengths, dev)[0], _splits_of_lengths(q_lengths, dev)[0]
    if _one_pass_radius():
        return _DenseRadiusPlan(supports, queries, radius, prs, qrs, grid_from=grid_from)
    return _RadiusPlan(supports, queries, radius, prs, qrs, defer=True)


def radius_fill_dense(plan, n_supports, max_cols=None):
    """Second half: the dense int32 [Nq, longest] matrix padded with the shadow index."""
    plan.resolve()
    dev = plan.points.device
    cols = plan.longest if max_cols is None else min(plan.longest, int(max_cols))
    if plan.nq == 0 or cols == 0:
        if isinstance(plan, _DenseRadiusPlan):
            plan.filled = True
        return torch.empty((plan.nq, cols), dtype=torch.int32, device=dev)
    if isinstance(plan, _DenseRadiusPlan):
        return plan.fill_dense(cols, n_supports)
    idx, _ = plan.fill(dense_cols=cols, pad_value=n_supports)
    return idx


def radius_neighbors_dense(queries, supports, q_lengths, s_lengths, radius, max_cols=None):
    """``batch_neighbors`` (ml3d/torch/models/kpconv.py:2002-2034) on the GPU: dense int32 [Nq, max_nbrs]
    neighbour matrix padded with the shadow index Ns; search + ragged_to_dense fused in one fill kernel."""
    return radius_fill_dense(radius_plan_dense(queries, supports, q_lengths, s_lengths, radius), supports.shape[0], max_cols)


def ragged_to_dense(values, row_splits, out_col_size, default_value):
    """``open3d.ml.torch.ops.ragged_to_dense`` (kpconv.py:2030, point_pillars.py:364)."""
    lib = _abi.get()
    _need_gpu(values, row_splits)
    values = values.contiguous()
    if values.element_size() not in (4, 8):
        raise RuntimeError("ragged_to_dense: 4- or 8-byte element types only")
    inner = tuple(values.shape[1:])
    elem = values.element_size()
    for s in inner:
        elem *= s
    dev = values.device
    dv = torch.as_tensor(default_value, dtype=values.dtype).to(dev).expand(inner if inner else ()).contiguous()
    rs = row_splits.to(device=dev, dtype=torch.int64).contiguous()
    rows = rs.numel() - 1
    out = torch.empty((rows, int(out_col_size)) + inner, dtype=values.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_ragged_to_dense(values.data_ptr(), rs.data_ptr(), rows, int(out_col_size), elem, dv.data_ptr(),
                                      out.data_ptr(), _stream())
    _abi.check(rc, "ml3d_ragged_to_dense")
    return out


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
            raise RuntimeError("subsample: a batch item spans >= 2^48 voxels at this sampleDl (unsupported)")
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


# ---------------------------------------------------------------------------------------------------
# KPConv blocks
# ---------------------------------------------------------------------------------------------------
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
        self.wsb = lib.ml3d_subsample_workspace_bytes(self.n, self.B)
        self.ws = _ws(self.wsb, dev)
        self.out_len = torch.empty(self.B, dtype=torch.int64, device=dev)
        self.stats = torch.empty(2, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_subsample_count(self.src.data_ptr(), self.rs.data_ptr(), self.B, self.n, float(sampleDl),
                                          self.out_len.data_ptr(), self.stats.data_ptr(), self.ws.data_ptr(), self.wsb, _stream())
        _abi.check(rc, "ml3d_subsample_count")
        self.M = None

    def resolve(self, values=None):
        if self.M is None:
            self.M, err = (int(x) for x in (self.stats.tolist() if values is None else values))
            if err:
                raise RuntimeError("subsample: a batch item spans >= 2^48 voxels at this sampleDl (unsupported)")
        return self

    def fill(self):
        """-> (pooled points [M, 3], pooled lengths int32 [B] on the device)"""
        lib = _abi.get()
        self.resolve()
        dev = self.src.device
        op = torch.empty((self.M, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
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


def kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, weights_kc_o, bias, extent, act=1, slope=0.1,
                 influence=1):
    """KPConv rigid aggregation + folded BN + activation (kpconv.py:1048-1159, 1357-1358).
    weights_kc_o: [15 * cin, cout] (BN-folded), neighb_inds int32 [Nq, H] with shadow index Ns."""
    lib = _abi.get()
    _need_gpu(q_pts, s_pts, neighb_inds, x, kernel_points, weights_kc_o)
    dev = x.device
    nq, ns = q_pts.shape[0], s_pts.shape[0]
    H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
    cin = x.shape[1]
    K = kernel_points.shape[0]
    cout = weights_kc_o.shape[1]
    for t in (q_pts, s_pts, x, kernel_points, weights_kc_o):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("kpconv_rigid: float32 contiguous tensors required")
    if neighb_inds.dtype != torch.int32 or not neighb_inds.is_contiguous():
        raise RuntimeError("kpconv_rigid: neighbour indices must be contiguous int32")
    if weights_kc_o.shape[0] != K * cin:
        raise RuntimeError("kpconv_rigid: weight shape does not match [K * cin, cout]")
    out = torch.empty((nq, cout), dtype=torch.float32, device=dev)
    wsb = lib.ml3d_kpconv_workspace_bytes(nq, cin, cout, K)
    if wsb == 0:
        raise RuntimeError("kpconv_rigid: unsupported configuration (15 kernel points, cin <= 512)")
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_kpconv_rigid(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, x.data_ptr(),
                                   cin, kernel_points.data_ptr(), K, float(extent), int(influence),
                                   weights_kc_o.data_ptr(), None if bias is None else bias.data_ptr(), int(act),
                                   float(slope), cout, out.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_kpconv_rigid")
    return out


def linear(a, weights_t, bias=None, a2=None, gather=None, residual=None, act=0, slope=0.0, residual_gather=None):
    """act([gather(a) | a2] @ weights_t + bias + residual) — UnaryBlock / decoder step (kpconv.py:1288-1293,
    283-286).  gather: int32 [M, H] neighbour matrix whose FIRST column selects the row of ``a`` (closest_pool).
    residual_gather: int32 [M, H] neighbour matrix whose first column selects the ROW OF ``residual`` added to output row m
    (rows >= residual.shape[0], the shadow index, add nothing)."""
    lib = _abi.get()
    _need_gpu(a, weights_t, bias, a2, gather, residual)
    dev = a.device
    k1 = a.shape[1]
    k2 = 0 if a2 is None else a2.shape[1]
    n = weights_t.shape[1]
    if weights_t.shape[0] != k1 + k2:
        raise RuntimeError("linear: weight rows %d != input columns %d" % (weights_t.shape[0], k1 + k2))
    if gather is not None:
        if gather.dtype != torch.int32 or not gather.is_contiguous():
            raise RuntimeError("linear: gather must be contiguous int32")
        m, gstride = gather.shape[0], gather.shape[1] if gather.dim() == 2 else 1
    else:
        m, gstride = a.shape[0], 0
    for t in (a, weights_t, bias, a2, residual):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError("linear: float32 contiguous tensors required")
    rg_stride = 0
    if residual_gather is not None:
        if residual is None or residual_gather.dtype != torch.int32 or not residual_gather.is_contiguous() or \
                residual_gather.shape[0] != m:
            raise RuntimeError("linear: residual_gather must be a contiguous int32 [M, H] matrix next to a residual")
        rg_stride = residual_gather.shape[1] if residual_gather.dim() == 2 else 1
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    wsb = lib.ml3d_linear_workspace_bytes(m, n, k1 + k2)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_linear(a.data_ptr(), k1, k1, None if gather is None else gather.data_ptr(), gstride, a.shape[0],
                             None if a2 is None else a2.data_ptr(), k2, k2, weights_t.data_ptr(),
                             None if bias is None else bias.data_ptr(),
                             None if residual is None else residual.data_ptr(), n,
                             None if residual_gather is None else residual_gather.data_ptr(), rg_stride,
                             0 if residual is None else residual.shape[0], int(act), float(slope),
                             out.data_ptr(), n, m, n, ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_linear")
    return out


def gather_pool(x, inds, mode):
    """mode 'max' = max_pool (kpconv.py:841-858), 'closest' = closest_pool (kpconv.py:821-838)."""
    lib = _abi.get()
    _need_gpu(x, inds)
    if x.dtype != torch.float32 or not x.is_contiguous() or inds.dtype != torch.int32 or not inds.is_contiguous():
        raise RuntimeError("gather_pool: float32 features and int32 indices (contiguous) required")
    out = torch.empty((inds.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_gather_pool(x.data_ptr(), x.shape[0], x.shape[1], inds.data_ptr(), inds.shape[0], inds.shape[1],
                                  0 if mode == "max" else 1, out.data_ptr(), _stream())
    _abi.check(rc, "ml3d_gather_pool")
    return out


# ---------------------------------------------------------------------------------------------------
# PointPillars blocks
# ---------------------------------------------------------------------------------------------------
def pillar_features(points, vox, in_channels, max_num_points, vx, vy, x_offset, y_offset, nx, ny, layers, batch):
    """Fused dense-gather + PillarFeatureNet + PointPillarsScatter (point_pillars.py:359-382, 512-555, 577-616).
    ``vox``: VoxelizeResult of the whole batch; ``layers``: [(wt [cin, units], bias [units]), ...] BN-folded.
    Returns the NHWC canvas [batch, ny, nx, units_last]."""
    lib = _abi.get()
    _need_gpu(points, vox.voxel_coords)
    dev = points.device
    if points.dtype != torch.float32 or points.dim() != 2 or points.stride(1) != 1:
        raise RuntimeError("pillar_features: points must be float32 [N, C] rows")
    stride = points.stride(0) if points.shape[0] > 1 else points.shape[1]
    M = vox.voxel_coords.shape[0]
    nl = len(layers)
    units = (C.c_int32 * nl)(*[int(w.shape[1]) for w, _ in layers])
    cc = int(layers[-1][0].shape[1])
    canvas = torch.empty((int(batch), int(ny), int(nx), cc), dtype=torch.float32, device=dev)
    wsb = lib.ml3d_pillar_features_workspace_bytes(M, int(max_num_points), nl, units)
    ws = _ws(wsb, dev)
    tw = _abi.ptr_table([w.data_ptr() for w, _ in layers])
    tb = _abi.ptr_table([b.data_ptr() for _, b in layers])
    with torch.cuda.device(dev):
        rc = lib.ml3d_pillar_features(points.data_ptr(), stride, int(in_channels), vox.voxel_coords.data_ptr(),
                                      vox.voxel_point_indices.data_ptr(), vox.voxel_point_row_splits.data_ptr(),
                                      vox.voxel_batch_splits.data_ptr(), int(batch), M, int(max_num_points), float(vx),
                                      float(vy), float(x_offset), float(y_offset), int(nx), int(ny), nl, units, tw, tb,
                                      canvas.data_ptr(), cc, ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_pillar_features")
    return canvas


def conv2d_nhwc(x, weights, bias, kh, kw, stride, pad, act=2, slope=0.0, out=None, out_channel_offset=0):
    """Conv2d + folded BN + activation on NHWC maps (SECOND, point_pillars.py:640-682)."""
    lib = _abi.get()
    _need_gpu(x, weights, bias)
    B, H, W, Cin = x.shape
    cout = weights.shape[1]
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, cout), dtype=torch.float32, device=x.device)
    ld = out.shape[3]
    wsb = lib.ml3d_conv2d_workspace_bytes(B, OH, OW, Cin, cout, kh, kw)
    ws = _ws(wsb, x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_conv2d_nhwc(x.data_ptr(), B, H, W, Cin, weights.data_ptr(), None if bias is None else bias.data_ptr(),
                                  kh, kw, stride, pad, act, slope, cout, out.data_ptr() + 4 * out_channel_offset, ld,
                                  ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_conv2d_nhwc")
    return out


def deconv2d_nhwc(x, weights, bias, stride, cout, act=2, slope=0.0, out=None, out_channel_offset=0):
    """ConvTranspose2d(kernel == stride) + folded BN + activation (SECONDFPN, point_pillars.py:712-717, 749)."""
    lib = _abi.get()
    _need_gpu(x, weights, bias)
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, H * stride, W * stride, cout), dtype=torch.float32, device=x.device)
    ld = out.shape[3]
    wsb = lib.ml3d_conv2d_workspace_bytes(B, H, W, Cin, stride * stride * cout, 1, 1)
    ws = _ws(wsb, x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_deconv2d_nhwc(x.data_ptr(), B, H, W, Cin, weights.data_ptr(),
                                    None if bias is None else bias.data_ptr(), stride, act, slope, cout,
                                    out.data_ptr() + 4 * out_channel_offset, ld, ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_deconv2d_nhwc")
    return out


def nhwc_to_nchw(x, channel_offset, channels):
    lib = _abi.get()
    _need_gpu(x)
    B, H, W, ld = x.shape
    out = torch.empty((B, channels, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_nhwc_to_nchw(x.data_ptr(), ld, int(channel_offset), int(channels), B, H * W, out.data_ptr(), _stream())
    _abi.check(rc, "ml3d_nhwc_to_nchw")
    return out


def nms(boxes, scores, nms_overlap_thresh):
    """``open3d.ml.torch.ops.nms`` (ml3d/torch/utils/objdet_helper.py:346): rotated-BEV NMS on boxes
    [N, 5] = (x0, y0, x1, y1, r); returns the kept indices (int64) in descending-score order."""
    lib = _abi.get()
    _need_gpu(boxes, scores)
    boxes = boxes.detach().contiguous().float()
    scores = scores.detach().contiguous().float()
    n = boxes.shape[0]
    dev = boxes.device
    if boxes.dim() != 2 or boxes.shape[1] != 5 or scores.numel() != n:
        raise RuntimeError("nms: boxes must be [N, 5] and scores [N]")
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    wsb = lib.ml3d_nms_workspace_bytes(n)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.ml3d_nms(boxes.data_ptr(), scores.data_ptr(), n, float(nms_overlap_thresh), keep.data_ptr(),
                          count.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_nms")
    return keep[:int(count.item())]


def _head_map(t):
    """[B, ch, H, W] float32 head map in any layout whose (H, W) plane has ONE pixel stride (NCHW tensors, channel slices of
    an NHWC tensor viewed as NCHW) -> (tensor, (batch, channel, pixel) element strides); anything else is made contiguous."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 4:
        raise RuntimeError("pointpillars_boxes: head maps must be [B, channels, H, W]")
    if t.shape[2] > 1 and t.stride(2) != t.shape[3] * t.stride(3):
        t = t.contiguous()
    return t, (t.stride(0), t.stride(1), t.stride(3))


def pointpillars_boxes(cls_scores, bbox_preds, dir_preds, anchors, nms_pre, score_thr, iou_thr, dir_offset=0.0):
    """``Anchor3DHead.get_bboxes`` for the whole batch with no host read-back inside (point_pillars.py:945-1025): head maps
    [B, A*C | A*7 | A*2, H, W] (the reference's NCHW tensors, or NCHW VIEWS of a fused NHWC head tensor -- no copy either
    way), ``anchors`` [H*W*A, 7] -> (rows [B, C*k, 9], total [B] int32): ``rows[b, :total[b]]`` = (x, y, z, w, l, h, yaw,
    score, label) of sample b's detections, class-major in NMS order."""
    lib = _abi.get()
    _need_gpu(cls_scores, bbox_preds, dir_preds, anchors)
    (cls_scores, s_cls), (bbox_preds, s_reg), (dir_preds, s_dir) = (_head_map(t) for t in (cls_scores, bbox_preds, dir_preds))
    anchors = anchors.contiguous().float()
    B, AC, H, W = cls_scores.shape
    A = dir_preds.shape[1] // 2
    C_ = AC // A
    if bbox_preds.shape[1] != A * 7 or anchors.shape[0] != H * W * A or anchors.shape[1] != 7:
        raise RuntimeError("pointpillars_boxes: head maps / anchors do not agree on the anchor count")
    dev = cls_scores.device
    n_anchor = H * W * A
    strides = (C.c_int64 * 9)(*[int(v) for v in s_cls + s_reg + s_dir])
    with torch.cuda.device(dev):
        if n_anchor > int(nms_pre):
            smax = torch.empty((B, n_anchor), dtype=torch.float32, device=dev)
            rc = lib.ml3d_pp_anchor_scores(cls_scores.data_ptr(), strides, B, A, C_, H * W, smax.data_ptr(), _stream())
            _abi.check(rc, "ml3d_pp_anchor_scores")
            cand = torch.topk(smax, int(nms_pre), dim=1)[1].contiguous()
        else:
            cand = torch.arange(n_anchor, dtype=torch.int64, device=dev).repeat(B, 1).contiguous()
        k = cand.shape[1]
        rows = torch.empty((B, C_ * k, 9), dtype=torch.float32, device=dev)
        total = torch.empty(B, dtype=torch.int32, device=dev)
        wsb = lib.ml3d_pp_boxes_workspace_bytes(B, k, C_)
        if wsb == 0:
            raise RuntimeError("pointpillars_boxes: nms_pre = %d candidates per sample is beyond the batched kernel (4096)" % k)
        ws = _ws(wsb, dev)
        rc = lib.ml3d_pp_boxes(cls_scores.data_ptr(), bbox_preds.data_ptr(), dir_preds.data_ptr(), strides, anchors.data_ptr(),
                               cand.data_ptr(), B, k, A, C_, H * W, float(score_thr), float(iou_thr), float(dir_offset),
                               rows.data_ptr(), total.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_pp_boxes")
    return rows, total


# ---------------------------------------------------------------------------------------------------
# patch sampler / vote accumulation
# ---------------------------------------------------------------------------------------------------
def _iou(fn_name, boxes_a, boxes_b, cols):
    lib = _abi.get()
    _need_gpu(boxes_a, boxes_b)
    a = boxes_a.contiguous().float()
    b = boxes_b.contiguous().float()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != cols or b.shape[1] != cols:
        raise RuntimeError("%s: boxes must be float32 [N, %d] / [M, %d]" % (fn_name, cols, cols))
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = getattr(lib, fn_name)(a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], out.data_ptr(), _stream())
    _abi.check(rc, fn_name)
    return out


def iou_bev(boxes_a, boxes_b):
    """``open3d.ml.contrib.iou_bev_*`` (ml3d/metrics/mAP.py:85): rotated bird's-eye-view IoU of every pair;
    boxes [N, 5] / [M, 5] = (x, z, w, l, yaw) -> float32 [N, M]."""
    return _iou("ml3d_iou_bev", boxes_a, boxes_b, 5)


def iou_3d(boxes_a, boxes_b):
    """``open3d.ml.contrib.iou_3d_*`` (ml3d/metrics/mAP.py:87): 3-D IoU of every pair; boxes [N, 7] / [M, 7] =
    (x, y, z, w, h, l, yaw) with y the bottom face (camera frame, y down) -> float32 [N, M]."""
    return _iou("ml3d_iou_3d", boxes_a, boxes_b, 7)


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
