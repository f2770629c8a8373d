"""The training side of the segmentation models on hand-written HIP, forward AND backward (csrc/train.hip).

What the reference leaves to ``torch.autograd`` around ``loss.backward()`` (ml3d/torch/pipelines/semantic_segmentation.py:412-437):
every Linear / 1x1 convolution, BatchNorm on the batch statistics (+ LeakyReLU), the index gathers / pools, and the attentive
pooling stage of RandLA-Net.  Each class below is a ``torch.autograd.Function`` whose two passes are C-ABI calls into
``libml3d_hip.so``; torch carries the graph, the optimiser and the loss, nothing else."""
import torch

from .. import _abi
from . import _gates
from .kpconv import linear


def _stream():
    return _gates._stream()


def _need_gpu(*tensors):
    return _gates._need_gpu(*tensors)


def _f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def gemm_tn(a, b, with_col_sums=False):
    """``a^T b`` over the rows: a [m, k], b [m, n] -> [k, n] (+ the column sums of a, [k]) on ``ml3d_gemm_tn``."""
    lib = _abi.get()
    _need_gpu(a, b)
    a, b = _f32(a), _f32(b)
    m, k = a.shape
    n = b.shape[1]
    if b.shape[0] != m:
        raise RuntimeError("gemm_tn: a and b must share their rows")
    c = torch.empty((k, n), dtype=torch.float32, device=a.device)
    sums = torch.empty(k, dtype=torch.float32, device=a.device) if with_col_sums else None
    with torch.cuda.device(a.device):
        rc = lib.ml3d_gemm_tn(a.data_ptr(), k, b.data_ptr(), n, m, k, n, c.data_ptr(), n,
                              None if sums is None else sums.data_ptr(), _stream())
    _abi.check(rc, "ml3d_gemm_tn")
    return (c, sums) if with_col_sums else c


class LinearFunction(torch.autograd.Function):
    """``y = x W^T (+ bias)`` for x [..., in], W [out, in] (``nn.Linear``'s layout; a 1x1 ``Conv2d`` weight ``[out, in, 1, 1]``
    viewed as such): forward on ``ml3d_linear`` (f32 MFMA), backward ``grad_x = grad_y W`` on ``ml3d_linear`` and
    ``grad_W = grad_y^T x``, ``grad_bias = sum grad_y`` on ``ml3d_gemm_tn``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, weight, bias)
        x2 = _f32(x.reshape(-1, x.shape[-1]))
        w = _f32(weight)
        y = linear(x2, w.t().contiguous(), None if bias is None else _f32(bias))
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.lead = tuple(x.shape[:-1])
        return y.reshape(ctx.lead + (w.shape[0],))

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        g2 = _f32(gy.reshape(-1, w.shape[0]))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = linear(g2, w, None).reshape(ctx.lead + (w.shape[1],))
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if ctx.has_bias:
                gw, gb = gemm_tn(g2, x2, with_col_sums=True)
            else:
                gw = gemm_tn(g2, x2)
        return gx, gw, gb


class BatchNormActFunction(torch.autograd.Function):
    """Training-mode batch normalisation over the rows of x [..., C] (``F.batch_norm(..., training=True)``: statistics of THIS
    batch, biased variance) fused with LeakyReLU(``slope``) when ``slope`` is not None: ``ml3d_batchnorm_train_forward /
    _backward``.  ``running_mean / running_var`` are updated in place like torch does (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, slope):
        lib = _abi.get()
        _need_gpu(x, gamma, beta)
        c = x.shape[-1]
        x2 = _f32(x.reshape(-1, c))
        rows = x2.shape[0]
        if rows < 2:
            raise ValueError("BatchNormActFunction: expected more than 1 row per channel when training")
        g = None if gamma is None else _f32(gamma)
        b = None if beta is None else _f32(beta)
        dev = x2.device
        y = torch.empty_like(x2)
        mean, var, invstd = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
        wsb = lib.ml3d_batchnorm_train_workspace_bytes(c)
        ws = _gates._ws(wsb, dev)
        act = 0 if slope is None else 1
        with torch.cuda.device(dev):
            rc = lib.ml3d_batchnorm_train_forward(x2.data_ptr(), rows, c, None if g is None else g.data_ptr(),
                                                  None if b is None else b.data_ptr(), float(eps), act, float(slope or 0.0),
                                                  y.data_ptr(), mean.data_ptr(), var.data_ptr(), invstd.data_ptr(), ws.data_ptr(), wsb,
                                                  _stream())
        _abi.check(rc, "ml3d_batchnorm_train_forward")
        if running_mean is not None and running_var is not None:
            with torch.no_grad():
                if momentum is None:      # torch: cumulative moving average, factor 1 / num_batches_tracked (the caller counts)
                    raise ValueError("BatchNormActFunction: momentum=None (cumulative average) needs num_batches_tracked; use batch_norm_act()")
                mom = float(momentum)
                running_mean.mul_(1.0 - mom).add_(mean.to(running_mean.dtype), alpha=mom)
                running_var.mul_(1.0 - mom).add_(var.to(running_var.dtype), alpha=mom * rows / (rows - 1))
        ctx.save_for_backward(x2, y, g if g is not None else mean.new_empty(0), mean, invstd)
        ctx.has_gamma, ctx.has_beta = gamma is not None, beta is not None
        ctx.act, ctx.slope, ctx.shape = act, float(slope or 0.0), tuple(x.shape)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, gy):
        lib = _abi.get()
        x2, y, g, mean, invstd = ctx.saved_tensors
        rows, c = x2.shape
        g2 = _f32(gy.reshape(rows, c))
        dev = x2.device
        gx = torch.empty_like(x2)
        gg, gb = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(2))
        wsb = lib.ml3d_batchnorm_train_workspace_bytes(c)
        ws = _gates._ws(wsb, dev)
        with torch.cuda.device(dev):
            rc = lib.ml3d_batchnorm_train_backward(x2.data_ptr(), y.data_ptr(), g2.data_ptr(), rows, c,
                                                   g.data_ptr() if ctx.has_gamma else None, mean.data_ptr(), invstd.data_ptr(), ctx.act,
                                                   ctx.slope, gx.data_ptr(), gg.data_ptr(), gb.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_batchnorm_train_backward")
        return (gx.reshape(ctx.shape), gg if ctx.has_gamma else None, gb if ctx.has_beta else None, None, None, None, None, None)


def batch_norm_act(x, bn, slope=None):
    """``bn`` (an ``nn.BatchNorm1d / 2d`` in training mode) over the rows of the point-major x [..., C] + LeakyReLU(slope)."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = bn.momentum
    if momentum is None and bn.track_running_stats:       # torch's cumulative moving average: factor 1 / num_batches_tracked
        momentum = 1.0 / float(max(int(bn.num_batches_tracked), 1)) if bn.num_batches_tracked is not None else 0.0
    return BatchNormActFunction.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                      bn.running_var if bn.track_running_stats else None, momentum, bn.eps, slope)


class GatherRowsFunction(torch.autograd.Function):
    """``x[index]`` for x [n, C] and an int32 row index [m] (a row outside [0, n) reads zeros: the shadow neighbour of
    kpconv.py:809-811) with the scatter-add adjoint: nearest_interpolation (randlanet.py:329-350), closest_pool
    (kpconv.py:821-838)."""

    @staticmethod
    def forward(ctx, x, index):
        lib = _abi.get()
        _need_gpu(x, index)
        x = _f32(x)
        if index.dtype != torch.int32 or index.dim() != 1 or not index.is_contiguous():
            raise RuntimeError("GatherRowsFunction: index must be a contiguous int32 vector")
        n, c = x.shape
        m = index.shape[0]
        out = torch.empty((m, c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_gather_rows(x.data_ptr(), n, c, index.data_ptr(), 1, m, out.data_ptr(), _stream())
        _abi.check(rc, "ml3d_gather_rows")
        ctx.save_for_backward(index)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        index, = ctx.saved_tensors
        g = _f32(g)
        m, c = g.shape
        gx = torch.empty((ctx.n, c), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.ml3d_scatter_add_rows(g.data_ptr(), ctx.n, c, index.data_ptr(), 1, m, gx.data_ptr(), _stream())
        _abi.check(rc, "ml3d_scatter_add_rows")
        return gx, None


class GatherPoolFunction(torch.autograd.Function):
    """``max_pool`` / ``closest_pool`` of KPConv's blocks (kpconv.py:821-858) on ``ml3d_gather_pool`` with the hand-written
    adjoint ``ml3d_gather_pool_backward`` (max: first maximal neighbour in list order; shadow rows count as zeros and take no
    gradient)."""

    @staticmethod
    def forward(ctx, x, inds, mode):
        from .kpconv import gather_pool
        x = _f32(x)
        out = gather_pool(x, inds, mode)
        ctx.save_for_backward(x, inds)
        ctx.mode = 0 if mode == "max" else 1
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        x, inds = ctx.saved_tensors
        g = _f32(g)
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_gather_pool_backward(x.data_ptr(), x.shape[0], x.shape[1], inds.data_ptr(), inds.shape[0], inds.shape[1],
                                               ctx.mode, g.data_ptr(), gx.data_ptr(), _stream())
        _abi.check(rc, "ml3d_gather_pool_backward")
        return gx, None, None


def attention_stage_supported(k, c1, c2, rows=0):
    """The shapes ``ml3d_randla_attention_stage`` takes (train.hip attn_check): K = 16, d = c1 + c2 even and <= 256, and
    ``rows`` = B * N below 2^31 / 16 (32-bit row arithmetic) -- callers fall back to the unfused path otherwise."""
    d = c1 + c2
    return k == 16 and d <= 256 and d % 2 == 0 and (d <= 128 or c1 <= 160) and int(rows) < (1 << 31) // 16


class AttentionStageFunction(torch.autograd.Function):
    """One attentive pooling of ``LocalFeatureAggregation`` in training form as ONE kernel per pass (randlanet.py:596-605, 617,
    631-637): ``f`` [B, N, c1] per-point features, ``enc`` [B, N, K, c2] encoded relative positions, ``idx`` int32 [B, N, K],
    the score Linear's ``weight`` [d, d] / ``bias`` [d] (d = c1 + c2) -> ``sum_k softmax_k(x W^T + b) * x`` [B, N, d] with
    ``x = [f[idx] | enc]``.  Neither pass materialises a [B, N, K, d] tensor; the backward recomputes the softmax."""

    @staticmethod
    def forward(ctx, f, enc, idx, weight, bias):
        lib = _abi.get()
        _need_gpu(f, enc, idx, weight, bias)
        f, enc, w = _f32(f), _f32(enc), _f32(weight)
        b = None if bias is None else _f32(bias)
        B, n, c1 = f.shape
        K, c2 = enc.shape[2], enc.shape[3]
        if idx.dtype != torch.int32 or not idx.is_contiguous() or tuple(idx.shape) != (B, n, K) or tuple(enc.shape[:2]) != (B, n):
            raise RuntimeError("AttentionStageFunction: idx must be a contiguous int32 [B, N, K] tensor matching enc [B, N, K, c2]")
        d = c1 + c2
        if tuple(w.shape) != (d, d):
            raise RuntimeError("AttentionStageFunction: weight must be [%d, %d]" % (d, d))
        wt = w.t().contiguous()
        out = torch.empty((B, n, d), dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            rc = lib.ml3d_randla_attention_stage(f.data_ptr(), enc.data_ptr(), idx.data_ptr(), wt.data_ptr(),
                                                 None if b is None else b.data_ptr(), B, n, K, c1, c2, out.data_ptr(), _stream())
        _abi.check(rc, "ml3d_randla_attention_stage")
        ctx.save_for_backward(f, enc, idx, w, wt, b if b is not None else w.new_empty(0), out)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        f, enc, idx, w, wt, b, out = ctx.saved_tensors
        B, n, c1 = f.shape
        K, c2 = enc.shape[2], enc.shape[3]
        d = c1 + c2
        g = _f32(g)
        gf, genc, gw = torch.empty_like(f), torch.empty_like(enc), torch.empty_like(w)
        gb = torch.empty(d, dtype=torch.float32, device=f.device) if ctx.has_bias else None
        wsb = lib.ml3d_randla_attention_stage_backward_workspace_bytes(B, n, c1, c2)
        ws = _gates._ws(wsb, f.device)
        with torch.cuda.device(f.device):
            rc = lib.ml3d_randla_attention_stage_backward(f.data_ptr(), enc.data_ptr(), idx.data_ptr(), w.data_ptr(), wt.data_ptr(),
                                                          b.data_ptr() if ctx.has_bias else None, out.data_ptr(), g.data_ptr(), B, n, K,
                                                          c1, c2, gf.data_ptr(), genc.data_ptr(), gw.data_ptr(),
                                                          None if gb is None else gb.data_ptr(), ws.data_ptr(), wsb, _stream())
        _abi.check(rc, "ml3d_randla_attention_stage_backward")
        return gf, genc, None, gw, gb


class KPConvDeformedFunction(torch.autograd.Function):
    """The aggregation of a DEFORMABLE KPConv in training (kpconv.py:1011-1066, 1105-1137; linear influence, sum aggregation):
    ``x`` [Ns, cin], ``deformed_kp`` [Nq, 15, 3] (the query's own kernel points, a function of the trained offsets), geometry ->
    ``wf`` [Nq, 15 * cin].  HIP forward and hand-written HIP backward with respect to the features (a scatter) AND the kernel points
    (what trains the offset convolution); the reference's ``[Nq, H, cin]`` neighbour gather exists in neither pass."""

    @staticmethod
    def forward(ctx, x, deformed_kp, q_pts, s_pts, neighb_inds, extent):
        lib = _abi.get()
        _need_gpu(x, deformed_kp, q_pts, s_pts, neighb_inds)
        x, dkp = _f32(x), _f32(deformed_kp)
        # the kernel takes raw pointers: float32 [Nq, 3] / [Ns, 3] points, int32 [Nq, H] neighbour rows, one feature row per support
        q_pts, s_pts = _f32(q_pts), _f32(s_pts)
        if neighb_inds.dtype != torch.int32 or not neighb_inds.is_contiguous():
            neighb_inds = neighb_inds.to(torch.int32).contiguous()
        if q_pts.dim() != 2 or q_pts.shape[1] != 3 or s_pts.dim() != 2 or s_pts.shape[1] != 3:
            raise ValueError("KPConvDeformedFunction: q_pts / s_pts must be [N, 3]")
        if x.dim() != 2 or x.shape[0] != s_pts.shape[0]:
            raise ValueError("KPConvDeformedFunction: x must hold one row per support point (%s vs %d)" % (tuple(x.shape), s_pts.shape[0]))
        if neighb_inds.dim() != 2 or neighb_inds.shape[0] != q_pts.shape[0]:
            raise ValueError("KPConvDeformedFunction: neighb_inds must be [Nq, H]")
        if dkp.dim() != 3 or dkp.shape[0] != q_pts.shape[0] or dkp.shape[2] != 3:
            raise ValueError("KPConvDeformedFunction: deformed_kp must be [Nq, K, 3]")
        nq, ns = q_pts.shape[0], s_pts.shape[0]
        H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
        cin, K = x.shape[1], dkp.shape[1]
        wf = torch.empty((nq, K * cin), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_kpconv_deformed_weighted(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, x.data_ptr(), cin,
                                                   dkp.data_ptr(), K, float(extent), wf.data_ptr(), _stream())
        _abi.check(rc, "ml3d_kpconv_deformed_weighted")
        ctx.save_for_backward(x, dkp, q_pts, s_pts, neighb_inds)
        ctx.extent = float(extent)
        return wf

    @staticmethod
    def backward(ctx, g):
        lib = _abi.get()
        x, dkp, q_pts, s_pts, neighb_inds = ctx.saved_tensors
        nq, ns = q_pts.shape[0], s_pts.shape[0]
        H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
        cin, K = x.shape[1], dkp.shape[1]
        g = _f32(g)
        gx, gk = torch.empty_like(x), torch.zeros_like(dkp)
        with torch.cuda.device(x.device):
            rc = lib.ml3d_kpconv_deformed_weighted_backward(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H,
                                                            x.data_ptr(), cin, dkp.data_ptr(), K, ctx.extent, g.data_ptr(), gx.data_ptr(),
                                                            gk.data_ptr(), _stream())
        _abi.check(rc, "ml3d_kpconv_deformed_weighted_backward")
        return gx, gk, None, None, None, None


class OffsetRegulariserFunction(torch.autograd.Function):
    """``p2p_fitting_regularizer`` of ONE deformable KPConv (kpconv.py:2167-2206, with the ``min_d2`` of kpconv.py:1058-1074) as a
    HIP op: ``deformed_kp`` [Nq, K, 3] + the block's geometry -> float32 [2] = (mean over (q, k) of min_h d^2 / extent^2, mean over
    (q, k) of the repulsive hinge) -- the two L1 terms the reference adds up as ``2 * fitting + repulsive``.  The forward also
    produces the gradient with respect to the kernel points (the other points of the repulsive term detached, as in the reference),
    so the backward is a scaled sum; the reference's [Nq, H, K] distance tensor is never built.  ``min_d2`` [Nq, K] is returned as a
    non-differentiable third output for callers that keep the reference's attribute."""

    @staticmethod
    def forward(ctx, deformed_kp, q_pts, s_pts, neighb_inds, extent, repulse_extent):
        lib = _abi.get()
        _need_gpu(deformed_kp, q_pts, s_pts, neighb_inds)
        dkp = _f32(deformed_kp)
        q_pts, s_pts = _f32(q_pts), _f32(s_pts)
        if neighb_inds.dtype != torch.int32 or not neighb_inds.is_contiguous():
            neighb_inds = neighb_inds.to(torch.int32).contiguous()
        nq, K = dkp.shape[0], dkp.shape[1]
        ns = s_pts.shape[0]
        H = neighb_inds.shape[1] if neighb_inds.dim() == 2 else 0
        if dkp.shape != (nq, K, 3) or q_pts.shape != (nq, 3) or (H and neighb_inds.shape[0] != nq):
            raise RuntimeError("OffsetRegulariserFunction: deformed_kp [Nq, K, 3], q_pts [Nq, 3], neighb_inds [Nq, H] expected")
        dev = dkp.device
        blocks = int(lib.ml3d_kpconv_offset_regulariser_blocks(nq))
        partial = torch.zeros((max(blocks, 1), 2), dtype=torch.float64, device=dev)
        min_d2 = torch.empty((nq, K), dtype=torch.float32, device=dev)
        gfit = torch.empty_like(dkp)
        grep = torch.empty_like(dkp)
        with torch.cuda.device(dev):
            rc = lib.ml3d_kpconv_offset_regulariser(q_pts.data_ptr(), s_pts.data_ptr(), neighb_inds.data_ptr(), nq, ns, H, dkp.data_ptr(), K,
                                                    float(extent), float(repulse_extent), min_d2.data_ptr(), gfit.data_ptr(),
                                                    grep.data_ptr(), partial.data_ptr(), _stream())
        _abi.check(rc, "ml3d_kpconv_offset_regulariser")
        ctx.save_for_backward(gfit, grep)
        ctx.denom = float(max(nq * K, 1))
        ctx.mark_non_differentiable(min_d2)
        return (partial.sum(0) / ctx.denom).to(torch.float32), min_d2

    @staticmethod
    def backward(ctx, g, _g_min):
        gfit, grep = ctx.saved_tensors
        return (gfit * (g[0] / ctx.denom) + grep * (g[1] / ctx.denom)), None, None, None, None, None
