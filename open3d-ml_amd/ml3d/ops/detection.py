"""PointPillars front end, BEV convolutions, box decode + NMS, box IoU."""
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


def pack_bf16x3(weights):
    """Split a [K, N] float weight matrix (K % 32 == 0) once into the three bf16 planes `conv2d_nhwc(..., packed=)` multiplies
    with on the bf16 matrix pipe (float32-equivalent result: include/ml3d_hip.h, ml3d_gemm_pack_bf16x3).  Returns a uint8
    tensor, or None when the matrix is not eligible (the caller keeps the f32 kernel)."""
    lib = _abi.get()
    _need_gpu(weights)
    K, N = int(weights.shape[0]), int(weights.shape[1])
    nbytes = int(lib.ml3d_gemm_pack_bf16x3_bytes(K, N))
    if nbytes == 0:
        return None
    w = weights.contiguous()
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        rc = lib.ml3d_gemm_pack_bf16x3(w.data_ptr(), K, N, packed.data_ptr(), nbytes, _stream())
    _abi.check(rc, "ml3d_gemm_pack_bf16x3")
    return packed


def conv2d_nhwc(x, weights, bias, kh, kw, stride, pad, act=2, slope=0.0, out=None, out_channel_offset=0, packed=None):
    """Conv2d + folded BN + activation on NHWC maps (SECOND, point_pillars.py:640-682).  `packed` = pack_bf16x3(weights) runs the
    same product on the bf16 matrix pipe (three-way split, float32-equivalent); `weights` still gives the shape."""
    lib = _abi.get()
    _need_gpu(x, weights, bias)
    B, H, W, Cin = x.shape
    cout = weights.shape[1]
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, cout), dtype=torch.float32, device=x.device)
    ld = out.shape[3]
    if packed is not None:
        with torch.cuda.device(x.device):
            rc = lib.ml3d_conv2d_nhwc_bf16x3(x.data_ptr(), B, H, W, Cin, packed.data_ptr(),
                                             None if bias is None else bias.data_ptr(), kh, kw, stride, pad, act, slope, cout,
                                             out.data_ptr() + 4 * out_channel_offset, ld, _stream())
        if rc != _abi.E_UNSUPPORTED:
            _abi.check(rc, "ml3d_conv2d_nhwc_bf16x3")
            return out
    wsb = lib.ml3d_conv2d_workspace_bytes(B, OH, OW, Cin, cout, kh, kw)
    ws = _ws(wsb, x.device)
    with torch.cuda.device(x.device):
        rc = lib.ml3d_conv2d_nhwc(x.data_ptr(), B, H, W, Cin, weights.data_ptr(), None if bias is None else bias.data_ptr(),
                                  kh, kw, stride, pad, act, slope, cout, out.data_ptr() + 4 * out_channel_offset, ld,
                                  ws.data_ptr(), wsb, _stream())
    _abi.check(rc, "ml3d_conv2d_nhwc")
    return out


def linear_bf16x3(a, packed, n, bias=None, act=0, slope=0.0, a2=None, residual=None, residual_gather=None):
    """act([a | a2] @ W + bias + residual) on the bf16 matrix pipe, `packed` = pack_bf16x3(W [K, n]) (float32-equivalent: pack_bf16x3).
    ``residual_gather``: int32 [M, H] neighbour matrix whose first column selects the ROW of ``residual`` added to output row m
    (rows >= residual.shape[0], the shadow index, add nothing) -- as ``ops.linear``.
    Returns None when the problem is not eligible (block widths % 32, alignment): the caller keeps ops.linear."""
    lib = _abi.get()
    _need_gpu(a, bias, a2, residual, residual_gather)
    for t in (a, a2, residual):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError("linear_bf16x3: float32 contiguous rows required")
    m, k1 = a.shape
    k2 = 0 if a2 is None else int(a2.shape[1])
    if (k1 % 32) or (k2 % 32):
        return None
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    wsb = int(lib.ml3d_linear_bf16x3_workspace_bytes(m, int(n), k1 + k2))
    ws = _ws(wsb, a.device)
    with torch.cuda.device(a.device):
        if residual_gather is not None:
            if residual is None or residual_gather.dtype != torch.int32 or not residual_gather.is_contiguous() or residual_gather.shape[0] != m:
                raise RuntimeError("linear_bf16x3: residual_gather must be a contiguous int32 [M, H] matrix next to a residual")
            rg_stride = residual_gather.shape[1] if residual_gather.dim() == 2 else 1
            rc = lib.ml3d_linear_bf16x3_gathered(a.data_ptr(), k1, k1, None if a2 is None else a2.data_ptr(), k2, k2, m, packed.data_ptr(),
                                                 None if bias is None else bias.data_ptr(), residual.data_ptr(), int(n),
                                                 residual_gather.data_ptr(), rg_stride, residual.shape[0], int(n), int(act), float(slope),
                                                 out.data_ptr(), int(n), ws.data_ptr(), wsb, _stream())
        else:
            rc = lib.ml3d_linear_bf16x3(a.data_ptr(), k1, k1, None if a2 is None else a2.data_ptr(), k2, k2, m, packed.data_ptr(),
                                        None if bias is None else bias.data_ptr(), None if residual is None else residual.data_ptr(),
                                        int(n), int(n), int(act), float(slope), out.data_ptr(), int(n), ws.data_ptr(), wsb, _stream())
    if rc == _abi.E_UNSUPPORTED:
        return None
    _abi.check(rc, "ml3d_linear_bf16x3")
    return out


def deconv2d_nhwc(x, weights, bias, stride, cout, act=2, slope=0.0, out=None, out_channel_offset=0, packed=None):
    """ConvTranspose2d(kernel == stride) + folded BN + activation (SECONDFPN, point_pillars.py:712-717, 749).  `packed` =
    pack_bf16x3(weights): the same GEMM on the bf16 matrix pipe."""
    lib = _abi.get()
    _need_gpu(x, weights, bias)
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, H * stride, W * stride, cout), dtype=torch.float32, device=x.device)
    ld = out.shape[3]
    if packed is not None:
        with torch.cuda.device(x.device):
            rc = lib.ml3d_deconv2d_nhwc_bf16x3(x.data_ptr(), B, H, W, Cin, packed.data_ptr(), None if bias is None else bias.data_ptr(),
                                               stride, act, slope, cout, out.data_ptr() + 4 * out_channel_offset, ld, _stream())
        if rc != _abi.E_UNSUPPORTED:
            _abi.check(rc, "ml3d_deconv2d_nhwc_bf16x3")
            return out
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


def topk_rows(values, k, with_values=False):
    """The ``nms_pre`` top-k of ``Anchor3DHead.get_bboxes_single`` (``max_scores.topk(self.nms_pre)``, point_pillars.py:985-992)
    for all rows at once: ``values`` float32 [rows, n] (or [n]) -> int64 indices [rows, k] (or [k]) of the k largest of every
    row in descending value, equal values by ascending index, NaN first -- a HIP radix select (``ml3d_topk_rows``), no
    ``torch.topk``.  ``with_values=True`` also returns the values."""
    lib = _abi.get()
    _need_gpu(values)
    v = values.detach()
    flat = v.dim() == 1
    if flat:
        v = v.unsqueeze(0)
    if v.dim() != 2:
        raise RuntimeError("topk_rows: values must be [rows, n] or [n]")
    v = v.contiguous().float()
    rows, n = v.shape
    k = int(k)
    if k < 0 or k > n:
        raise RuntimeError("topk_rows: k = %d outside [0, %d]" % (k, n))
    dev = v.device
    idx = torch.empty((rows, k), dtype=torch.int64, device=dev)
    val = torch.empty((rows, k), dtype=torch.float32, device=dev) if with_values else None
    if rows * k > 0:
        wsb = lib.ml3d_topk_rows_workspace_bytes(rows, n, k)
        if wsb == 0:
            raise RuntimeError("topk_rows: k = %d is beyond the kernel's 4096 candidates per row" % k)
        ws = _ws(wsb, dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_topk_rows(v.data_ptr(), rows, n, k, idx.data_ptr(), val.data_ptr() if with_values else None,
                                    ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_topk_rows")
    if flat:
        idx, val = idx[0], (val[0] if with_values else None)
    return (idx, val) if with_values else idx


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
            cand = topk_rows(smax, int(nms_pre))
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
