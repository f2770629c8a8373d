"""KPConv building blocks: rigid aggregation, Linear with fused gathers / residuals, pools."""
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

def kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, weights_kc_o, bias, extent, act=1, slope=0.1,
                 influence=1, offset_features=None, packed=None):
    """KPConv rigid aggregation + folded BN + activation (kpconv.py:1048-1159, 1357-1358).
    weights_kc_o: [15 * cin, cout] (BN-folded), neighb_inds int32 [Nq, H] with shadow index Ns.
    ``offset_features`` [Nq, 45 | 60]: the DEFORMABLE convolution (kpconv.py:1011-1066) -- kernel point k of query q at
    ``kernel_points[k] + offset_features[q, 3k:3k+3] * extent``, columns 45.. = modulation logits (see ``kpconv_deformable``).
    ``packed`` = ops.pack_bf16x3(weights_kc_o): the [15 cin, cout] contraction runs on the bf16 matrix pipe (float32-equivalent)."""
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
    if offset_features is not None:
        _need_gpu(offset_features)
        if offset_features.dtype != torch.float32 or not offset_features.is_contiguous() or offset_features.dim() != 2 or \
                offset_features.shape[0] != nq or offset_features.shape[1] not in (3 * K, 4 * K):
            raise RuntimeError("kpconv_deformable: offset_features must be contiguous float32 [Nq, 3K] or [Nq, 4K]")
        with torch.cuda.device(dev):
            rc = lib.ml3d_kpconv_deformable(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H,
                                            x.data_ptr(), cin, kernel_points.data_ptr(), K, float(extent), int(influence),
                                            offset_features.data_ptr(), int(offset_features.shape[1]),
                                            weights_kc_o.data_ptr(), None if bias is None else bias.data_ptr(), int(act),
                                            float(slope), cout, out.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_kpconv_deformable")
        return out
    if packed is not None:
        with torch.cuda.device(dev):
            rc = lib.ml3d_kpconv_rigid_bf16x3(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, x.data_ptr(),
                                              cin, kernel_points.data_ptr(), K, float(extent), int(influence),
                                              weights_kc_o.data_ptr(), packed.data_ptr(), None if bias is None else bias.data_ptr(),
                                              int(act), float(slope), cout, out.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_kpconv_rigid_bf16x3")
        return out
    with torch.cuda.device(dev):
        rc = lib.ml3d_kpconv_rigid(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, x.data_ptr(),
                                   cin, kernel_points.data_ptr(), K, float(extent), int(influence),
                                   weights_kc_o.data_ptr(), None if bias is None else bias.data_ptr(), int(act),
                                   float(slope), cout, out.data_ptr(), ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_kpconv_rigid")
    return out


def kpconv_deformable(q_pts, s_pts, neighb_inds, x, kernel_points, weights_kc_o, bias, extent, offset_weights_kc_o,
                      offset_bias, act=1, slope=0.1, influence=1):
    """The deformable KPConv of ``resnetb_deformable`` / ``simple_deformable`` blocks (kpconv.py:1011-1159): the inner rigid
    convolution cin -> 3K (+K modulated) gives each query its kernel-point offsets (``offset_weights_kc_o`` [15 * cin, 45 | 60],
    ``offset_bias``), then the convolution itself runs with the query's own kernel points.  ``KP_influence: linear`` only."""
    off = kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, offset_weights_kc_o, offset_bias, extent, act=0,
                       influence=influence)
    return kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, weights_kc_o, bias, extent, act=act, slope=slope,
                        influence=influence, offset_features=off)


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


# ---- training side: the rigid KPConv as a differentiable op --------------------------------------------------
def kpconv_weighted(q_pts, s_pts, neighb_inds, x, kernel_points, extent, influence=1):
    """wf [Nq, 15 * cin]: the kernel-point-weighted neighbour sums of ``KPConv.forward`` (kpconv.py:1105-1137) on the HIP
    aggregation kernels."""
    lib = _abi.get()
    _need_gpu(q_pts, s_pts, neighb_inds, x, kernel_points)
    nq, ns = q_pts.shape[0], s_pts.shape[0]
    H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
    cin, K = x.shape[1], kernel_points.shape[0]
    wf = torch.empty((nq, K * cin), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_kpconv_weighted(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, x.data_ptr(), cin,
                                      kernel_points.data_ptr(), K, float(extent), int(influence), wf.data_ptr(), _stream())
    _abi.check(rc, "ml3d_kpconv_weighted")
    return wf


def kpconv_weighted_backward(q_pts, s_pts, neighb_inds, grad_wf, cin, kernel_points, extent, influence=1):
    """dx [Ns, cin]: the adjoint of ``kpconv_weighted`` with respect to the features (a scatter-add over the neighbour lists)."""
    lib = _abi.get()
    _need_gpu(q_pts, s_pts, neighb_inds, grad_wf, kernel_points)
    nq, ns = q_pts.shape[0], s_pts.shape[0]
    H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
    K = kernel_points.shape[0]
    dx = torch.empty((ns, int(cin)), dtype=torch.float32, device=grad_wf.device)
    with torch.cuda.device(grad_wf.device):
        rc = lib.ml3d_kpconv_weighted_backward(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, int(cin),
                                               kernel_points.data_ptr(), K, float(extent), int(influence), grad_wf.data_ptr(),
                                               dx.data_ptr(), _stream())
    _abi.check(rc, "ml3d_kpconv_weighted_backward")
    return dx


class KPConvFunction(torch.autograd.Function):
    """``KPConv.forward`` of the rigid branch (kpconv.py:1005-1159, no bias / norm / activation: those are the caller's modules in
    training) with hand-written forward AND backward: forward = HIP aggregation + ``wf @ W``; backward: ``dW = wf^T @ g``,
    ``dwf = g @ W^T`` (this library's MFMA GEMMs: ``ml3d_gemm_tn`` / ``ml3d_linear``) and the HIP scatter ``ml3d_kpconv_weighted_backward`` for the features.  Geometry (points,
    neighbour lists, kernel points) gets no gradient -- the reference does not train it either (kpconv.py:959-963)."""

    @staticmethod
    def forward(ctx, x, weights, q_pts, s_pts, neighb_inds, kernel_points, extent, influence):
        x = x.contiguous()
        K, cin, cout = weights.shape
        wf = kpconv_weighted(q_pts, s_pts, neighb_inds, x, kernel_points, extent, influence)
        ctx.save_for_backward(wf, weights, q_pts, s_pts, neighb_inds, kernel_points)
        ctx.geom = (float(extent), int(influence), int(cin))
        return linear(wf, weights.reshape(K * cin, cout).contiguous())

    @staticmethod
    def backward(ctx, g):
        wf, weights, q_pts, s_pts, neighb_inds, kernel_points = ctx.saved_tensors
        extent, influence, cin = ctx.geom
        K, _, cout = weights.shape
        g = g.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[1]:
            from .train import gemm_tn
            dw = gemm_tn(wf, g).reshape(K, cin, cout)
        if ctx.needs_input_grad[0]:
            dwf = linear(g, weights.reshape(K * cin, cout).t().contiguous())
            dx = kpconv_weighted_backward(q_pts, s_pts, neighb_inds, dwf, cin, kernel_points, extent, influence)
        return dx, dw, None, None, None, None, None, None
