"""Neighbour searches: exact k-NN, the RandLA pyramid, fixed-radius search in its ragged\n(two-phase) and dense (one traversal) forms, ragged_to_dense."""
import ctypes as C
import threading

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


def radius_plan_dense(queries, supports, q_lengths, s_lengths, radius, grid_from=None, long_rows=False):
    """Deferred first half of ``radius_neighbors_dense``: the search is enqueued, its sizes not read yet.  ``grid_from``: an
    already filled plan over the same supports and radius whose grid is reused.  ``long_rows``: the caller expects rows beyond
    the one-traversal form's 128-entry stash (KPConv's deformable layers: deform_radius 6.0, hundreds of neighbours) -- go
    straight to the two-phase search instead of gathering once, overflowing and searching again."""
    dev = supports.device
    prs, qrs = _splits_of_lengths(s_lengths, dev)[0], _splits_of_lengths(q_lengths, dev)[0]
    if not long_rows:
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


# ---- the whole KPConv batch build in one library call (ml3d_kpconv_batch_build) --------------------------------------------
_ARENA_HINT = {}        # (device, 2^k bucket of n_points) -> arena bytes the last build of that size class needed (a sizing hint)
_PINNED = {}            # (device, stream, bytes bucket) -> pinned host scratch for the size read-backs, reused across batches


class KpBatchArena:
    """Result of ``kpconv_batch_build``: the arena tensor (device uint8) + typed views into it, per layer."""

    def __init__(self, arena, out, lengths, points0, has_conv):
        self.arena, self.host_syncs = arena, int(out.host_syncs)
        self.points, self.neighbors, self.pools, self.upsamples, self.lengths = [], [], [], [], []
        dev = arena.device
        e_i = torch.empty((0, 1), dtype=torch.int32, device=dev)

        def view(off, rows, cols, dtype):
            n = int(rows) * int(cols)
            return arena[int(off):int(off) + 4 * n].view(dtype).view(int(rows), int(cols))
        L = int(out.num_layers)
        for l in range(L):
            o = out.layer[l]
            self.points.append(points0 if l == 0 else view(o.points_offset, o.n_points, 3, torch.float32))
        for l in range(L):
            o = out.layer[l]
            n, n_next = int(o.n_points), (int(out.layer[l + 1].n_points) if l + 1 < L else 0)
            if o.conv_offset >= 0:
                self.neighbors.append(view(o.conv_offset, n, o.conv_cols, torch.int32))
            elif has_conv[l]:       # a level of 0 points or rows of 0 columns: the per-layer path's empty [n, cols]
                self.neighbors.append(torch.empty((n, int(o.conv_cols)), dtype=torch.int32, device=dev))
            else:                   # no convolution blocks on this layer: the [0, 1] placeholder
                self.neighbors.append(e_i)
            if l + 1 < L:
                self.pools.append(view(o.pool_offset, n_next, o.pool_cols, torch.int32) if o.pool_offset >= 0
                                  else torch.empty((n_next, int(o.pool_cols)), dtype=torch.int32, device=dev))
                self.upsamples.append(view(o.up_offset, n, o.up_cols, torch.int32) if o.up_offset >= 0
                                      else torch.empty((n, int(o.up_cols)), dtype=torch.int32, device=dev))
            else:
                self.pools.append(e_i)
                self.upsamples.append(e_i)
            self.lengths.append(lengths[l])


KPBATCH_TRACE = []       # measurement hook (bench_models.py): a FIFO of event lists -- each 8 torch.cuda.Event(enable_timing=True),
#                          already recorded once -- one of which the next build takes and records around layer 0's conv search /
#                          expand / subsample count / fill (list.pop is atomic: builds may start on several host threads)


def kpconv_batch_build(points, lengths, radii, dls, has_conv, rotations=None, cap=128, buffers=None):
    """``KPConvBatch.segmentation_inputs`` (ml3d/torch/dataloaders/concat_batcher.py:186-305) for rigid architectures in ONE
    library call: ``radii[l]`` / ``dls[l]`` the conv radius and pooling grid of layer l, ``has_conv[l]`` whether the layer has
    convolution blocks, ``rotations``: per pooling layer a float32 [B,3,3] device tensor or None.  Every layer but the last
    pools.  Returns a ``KpBatchArena`` (matrices identical to the per-layer ``radius_plan_dense`` / ``grid_subsampling_plan``
    calls) or None when some row outgrew the ``cap``-wide stash of the one-traversal search -- the caller then takes the
    per-layer two-phase path for this batch.
    ``buffers``: optional dict a pipeline keeps per build slot -- its ``'workspace'`` and ``'arena'`` tensors (uint8, on the device)
    are reused when large enough and replaced IN the dict when they had to grow, so that a steady stream of batches allocates
    nothing (a fresh multi-hundred-MB block per batch makes the caching allocator grow its per-stream pools for dozens of steps,
    and every growth is a hipMalloc that stalls the step: 20-80 ms on a cold box).  The caller owns the reuse discipline: the
    returned views alias ``buffers['arena']`` until the next build that is handed the same dict."""
    lib = _abi.get()
    _need_gpu(points)
    pts = points.contiguous().float()
    dev = pts.device
    L, B, n0 = len(radii), len(lengths), int(pts.shape[0])
    if L > _abi.KPBATCH_MAX_LAYERS or L < 1:
        raise RuntimeError("kpconv_batch_build: 1 .. %d layers" % _abi.KPBATCH_MAX_LAYERS)
    desc = _abi.KpBatchDesc()
    desc.num_layers, desc.cap = L, int(cap)
    for l in range(L):
        desc.has_conv[l], desc.radius[l], desc.dl[l] = int(bool(has_conv[l])), float(radii[l]), float(dls[l]) if l + 1 < L else 0.0
    try:
        trace = KPBATCH_TRACE.pop(0)
    except IndexError:
        trace = None
    if trace is not None:
        for i, ev in enumerate(trace):
            desc.trace_events[i] = ev.cuda_event
    lens = (C.c_int64 * B)(*[int(v) for v in lengths])
    if rotations is not None and L > 1 and all(isinstance(r, np.ndarray) for r in rotations[:L - 1]):
        # host orientations (the random draws of batch_grid_subsampling): ONE upload for all pooling layers
        stack = torch.from_numpy(np.ascontiguousarray(np.stack(rotations[:L - 1]), dtype=np.float32)).to(dev)
        rot_t = [stack[l] for l in range(L - 1)]
    else:
        rot_t = [None if (rotations is None or rotations[l] is None) else
                 torch.as_tensor(rotations[l], dtype=torch.float32).to(device=dev).contiguous() for l in range(L - 1)]
    rot_p = (C.c_void_p * max(1, L - 1))(*[None if t is None else t.data_ptr() for t in rot_t]) if L > 1 else None
    wsb = lib.ml3d_kpconv_batch_workspace_bytes(n0, B, L, int(cap))
    if wsb == 0:
        raise RuntimeError("kpconv_batch_build: unsupported sizes")
    ws = buffers.get('workspace') if buffers is not None else None
    if ws is None or ws.numel() < wsb or ws.device != dev:
        ws = _ws(int(wsb * 1.1) if buffers is not None else wsb, dev)
        if buffers is not None:
            buffers['workspace'] = ws
    hsb = int(lib.ml3d_kpconv_batch_host_scratch_bytes(B, L))
    stream = _stream()
    # (the stream handle is a ctypes c_void_p; the calling thread is part of the key: the scratch receives this call's size records
    #  while the call blocks on them, so two host threads must never share one -- DESIGN.md §11.11)
    hkey = (str(dev), getattr(stream, "value", stream), (hsb + 4095) // 4096, threading.get_ident())
    pinned = _PINNED.get(hkey)
    if pinned is None:
        if len(_PINNED) >= 16:
            _PINNED.clear()
        pinned = _PINNED[hkey] = torch.empty(((hsb + 4095) // 4096) * 4096, dtype=torch.uint8).pin_memory()
    out = _abi.KpBatchOut()
    out_lens = (C.c_int32 * (L * B))()
    bucket = (str(dev), max(1, n0).bit_length())
    # first guess: what the last batch of this size class took (+25 %), else ~3 matrices of 48 columns over 1.6 N rows + points
    arena_bytes = int(_ARENA_HINT.get(bucket, 0) * 1.25) or int(n0 * 1.6 * (3 * 48 * 4 + 12)) + (1 << 20)
    kept = buffers.get('arena') if buffers is not None else None
    for attempt in range(6):
        if kept is not None and kept.numel() >= arena_bytes and kept.device == dev:
            arena, arena_bytes = kept, kept.numel()
        else:
            arena = torch.empty(arena_bytes, dtype=torch.uint8, device=dev)
            if buffers is not None:
                kept = buffers['arena'] = arena
        with torch.cuda.device(dev):
            rc = lib.ml3d_kpconv_batch_build(pts.data_ptr(), C.addressof(lens), B, n0, C.addressof(desc), rot_p and C.addressof(rot_p),
                                             arena.data_ptr(), arena_bytes, C.addressof(out), C.addressof(out_lens), ws.data_ptr(), ws.numel(),
                                             pinned.data_ptr(), pinned.numel(), stream)
        if rc == -2 and out.arena_used > arena_bytes:        # ML3D_E_WORKSPACE: the arena was short -- grow and redo the batch
            arena_bytes = int(max(2 * arena_bytes, 2 * out.arena_used))
            kept = None
            continue
        break
    if rc == _abi.KPBATCH_FALLBACK:
        return None
    _abi.check(rc, "ml3d_kpconv_batch_build")
    _ARENA_HINT[bucket] = int(out.arena_used)
    lengths_t = [torch.tensor(list(out_lens[l * B:(l + 1) * B]), dtype=torch.int32) for l in range(L)]
    return KpBatchArena(arena, out, lengths_t, pts, [bool(h) for h in has_conv])
