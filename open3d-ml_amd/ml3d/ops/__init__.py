"""``ml3d.ops`` — torch front end of libml3d_hip.so (MI355X / gfx950 only).

Mirrors the call surface the reference gets from ``open3d.ml.torch.ops`` /
``open3d.core.nns`` for the inference hot path.  Every function
requires CUDA(HIP) tensors and raises ``RuntimeError`` otherwise: this package has no
CPU path (the CPU oracle lives under /oracle and is test infrastructure only).
All kernels are enqueued on torch's CURRENT stream.
"""
import ctypes as C
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


def randla_knn_pyramid(points, ratios, k, out=None, workspace=None):
    """All neighbour searches of ``RandLANet.transform`` (ml3d/torch/models/randlanet.py:218-229)
    for a batch [B, N, 3] in one call.  Returns (neighbor_idx[l] [B,n_l,k], interp_idx[l] [B,n_l,1]),
    int32, item-local.  ``sub_idx[l]`` of the reference is ``neighbor_idx[l][:, :n_{l+1}]``."""
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
        rc = lib.ml3d_randla_knn_pyramid(points.data_ptr(), B, n0, L, r, int(k), t_n, t_i, ws.data_ptr(),
                                         ws.numel(), _stream())
    _abi.check(rc, "ml3d_randla_knn_pyramid")
    return nbr, itp


def randla_forward(desc, params, features, points, neighbor_idx, interp_idx, out=None, workspace=None):
    """Fused RandLA-Net forward (ml3d/torch/models/randlanet.py:241-298) -> scores [B, N, classes]."""
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
    if out is None:
        out = torch.empty((B, n0, desc.num_classes), dtype=torch.float32, device=dev)
    wsb = lib.ml3d_randla_forward_workspace_bytes(C.byref(desc))
    ws = workspace if workspace is not None else _ws(wsb, dev)
    if ws.numel() < wsb:
        raise RuntimeError("randla_forward: workspace too small")
    t_n = _abi.ptr_table([t.data_ptr() for t in neighbor_idx])
    t_i = _abi.ptr_table([t.data_ptr() for t in interp_idx])
    with torch.cuda.device(dev):
        rc = lib.ml3d_randla_forward(C.byref(desc), params.data_ptr(), features.data_ptr(), points.data_ptr(),
                                     t_n, t_i, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _abi.check(rc, "ml3d_randla_forward")
    return out
