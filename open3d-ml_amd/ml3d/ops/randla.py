"""The fused RandLA-Net forward."""
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

from .search import pyramid_sizes


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


# ---- training side: random_sample as a differentiable op ------------------------------------------------------
class GatherMaxFunction(torch.autograd.Function):
    """``RandLANet.random_sample`` (randlanet.py:300-327): features [B, n_in, C] point-major, ``neighbor_idx`` int32
    [B, n_in, 16] (the level's neighbour matrix: the pooling rows are its first ``n_out``, randlanet.py:222-223) ->
    [B, n_out, C], the max over the 16 listed neighbours; HIP forward, hand-written HIP backward (the gradient goes to the
    first maximal neighbour, like ``torch.max``)."""

    @staticmethod
    def forward(ctx, features, neighbor_idx, n_out):
        lib = _abi.get()
        _need_gpu(features, neighbor_idx)
        features = features.contiguous()
        B, n_in, c = features.shape
        if neighbor_idx.dtype != torch.int32 or not neighbor_idx.is_contiguous() or tuple(neighbor_idx.shape) != (B, n_in, 16):
            raise RuntimeError("GatherMaxFunction: neighbor_idx must be a contiguous int32 [B, n_in, 16] tensor")
        n_out = int(n_out)
        out = torch.empty((B, n_out, c), dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device):
            rc = lib.ml3d_randla_gather_max(features.data_ptr(), neighbor_idx.data_ptr(), B, n_in, n_out, c, out.data_ptr(), _stream())
        _abi.check(rc, "ml3d_randla_gather_max")
        ctx.save_for_backward(features, neighbor_idx)
        ctx.n_out = n_out
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        features, neighbor_idx = ctx.saved_tensors
        B, n_in, c = features.shape
        g = g.contiguous()
        gf = torch.empty_like(features)
        with torch.cuda.device(features.device):
            rc = lib.ml3d_randla_gather_max_backward(features.data_ptr(), neighbor_idx.data_ptr(), g.data_ptr(), B, n_in, ctx.n_out,
                                                     c, gf.data_ptr(), _stream())
        _abi.check(rc, "ml3d_randla_gather_max_backward")
        return gf, None, None


class AttentivePoolFunction(torch.autograd.Function):
    """The pooling of ``AttentivePooling.forward`` (randlanet.py:622-637) up to its SharedMLP, point-major:
    ``scores``, ``x`` [..., K, C] -> ``sum_k softmax_k(scores) * x`` [..., C].  HIP forward; hand-written HIP backward that
    recomputes the softmax from the saved inputs, so no [rows, K, C] probability tensor lives between the passes."""

    @staticmethod
    def forward(ctx, scores, x):
        lib = _abi.get()
        _need_gpu(scores, x)
        if scores.shape != x.shape or x.dim() < 3:
            raise RuntimeError("AttentivePoolFunction: scores and x must share one [..., K, C] shape")
        scores, x = scores.contiguous().float(), x.contiguous().float()
        k, c = x.shape[-2], x.shape[-1]
        rows = x.numel() // (k * c)
        out = torch.empty(x.shape[:-2] + (c,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_randla_attentive_pool(scores.data_ptr(), x.data_ptr(), rows, k, c, out.data_ptr(), _stream())
        _abi.check(rc, "ml3d_randla_attentive_pool")
        ctx.save_for_backward(scores, x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        scores, x, out = ctx.saved_tensors
        k, c = x.shape[-2], x.shape[-1]
        rows = x.numel() // (k * c)
        g = g.contiguous().float()
        gs, gx = torch.empty_like(scores), torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_randla_attentive_pool_backward(scores.data_ptr(), x.data_ptr(), out.data_ptr(), g.data_ptr(), rows, k, c,
                                                         gs.data_ptr(), gx.data_ptr(), _stream())
        _abi.check(rc, "ml3d_randla_attentive_pool_backward")
        return gs, gx
