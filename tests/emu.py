"""Test helper: build/load the HOST emulation of libml3d_hip.so (tests/hipemu) and call the C ABI
with numpy buffers.  Lets the CPU suite execute the very same .hip sources the GPU runs."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

from ml3d import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def available():
    cxx = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    return os.path.exists(cxx) or shutil.which(cxx) is not None


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ML3D_EMU_LIB")          # e.g. the AddressSanitizer build of tools/emu_asan.sh
        if not so:
            so = subprocess.check_output([os.path.join(HERE, "hipemu", "build_emu.sh")]).decode().strip().splitlines()[-1]
        _LIB = _abi.bind(C.CDLL(so))
    return _LIB


def knn(points, psplits, queries=None, qsplits=None, k=16, local=False):
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    ps = np.ascontiguousarray(psplits, np.int64)
    if queries is None:
        queries, qs = points, ps
    else:
        queries = np.ascontiguousarray(queries, np.float32)
        qs = np.ascontiguousarray(qsplits, np.int64)
    B = len(ps) - 1
    wsb = L.ml3d_knn_workspace_bytes(len(points), len(queries), B)
    ws = np.zeros(wsb, np.uint8)
    idx = np.full((len(queries), k), -7, np.int32)
    d2 = np.zeros((len(queries), k), np.float32)
    rc = L.ml3d_knn_search(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, len(points),
                           len(queries), k, 1 if local else 0, idx.ctypes.data, d2.ctypes.data, ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    return idx, d2


def pyramid_ordered(points_bn3, ratios, k=16):
    """-> (nbr, itp, order): order[l] = the level's cell-sorted point rows [B * n_l] (ml3d_randla_knn_pyramid_ordered)."""
    L = lib()
    pts = np.ascontiguousarray(points_bn3, np.float32)
    B, n0, _ = pts.shape
    nl = len(ratios)
    r = (C.c_int32 * nl)(*ratios)
    n = [n0]
    for x in ratios:
        n.append(n[-1] // x)
    wsb = L.ml3d_randla_pyramid_workspace_bytes(B, n0, nl, r)
    ws = np.zeros(wsb, np.uint8)
    nbr = [np.full((B, n[l], k), -9, np.int32) for l in range(nl)]
    itp = [np.full((B, n[l], 1), -9, np.int32) for l in range(nl)]
    order = [np.full(B * n[l], -9, np.int32) for l in range(nl)]
    rc = L.ml3d_randla_knn_pyramid_ordered(pts.ctypes.data, B, n0, nl, r, k, _abi.ptr_table([x.ctypes.data for x in nbr]),
                                           _abi.ptr_table([x.ctypes.data for x in itp]),
                                           _abi.ptr_table([x.ctypes.data for x in order]), ws.ctypes.data, wsb, None, None)
    assert rc == 0, rc
    return nbr, itp, order


def pyramid(points_bn3, ratios, k=16):
    L = lib()
    pts = np.ascontiguousarray(points_bn3, np.float32)
    B, n0, _ = pts.shape
    nl = len(ratios)
    r = (C.c_int32 * nl)(*ratios)
    n = [n0]
    for x in ratios:
        n.append(n[-1] // x)
    wsb = L.ml3d_randla_pyramid_workspace_bytes(B, n0, nl, r)
    ws = np.zeros(wsb, np.uint8)
    nbr = [np.full((B, n[l], k), -9, np.int32) for l in range(nl)]
    itp = [np.full((B, n[l], 1), -9, np.int32) for l in range(nl)]
    rc = L.ml3d_randla_knn_pyramid(pts.ctypes.data, B, n0, nl, r, k, _abi.ptr_table([x.ctypes.data for x in nbr]),
                                   _abi.ptr_table([x.ctypes.data for x in itp]), ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    return nbr, itp


def randla_forward(cfg, sd, points, feats, nbr, itp, order=None):
    from ml3d.torch.models import _randla_pack
    L = lib()
    B, N, _ = points.shape
    desc = _abi.make_desc(cfg, B, N)
    off = _abi.randla_param_offsets(L, desc)
    params = _randla_pack.pack(sd, cfg, off)
    wsb = L.ml3d_randla_forward_workspace_bytes(C.byref(desc))
    ws = np.zeros(wsb, np.uint8)
    out = np.zeros((B, N, cfg["num_classes"]), np.float32)
    points = np.ascontiguousarray(points, np.float32)
    feats = np.ascontiguousarray(feats, np.float32)
    assert feats.shape == (B, N, cfg["in_channels"]), (feats.shape, cfg["in_channels"])      # (the library reads in_channels floats per row)
    if order is not None:
        order = [np.ascontiguousarray(o, np.int32) for o in order]
        rc = L.ml3d_randla_forward_ordered(C.byref(desc), params.ctypes.data, feats.ctypes.data, points.ctypes.data,
                                           _abi.ptr_table([x.ctypes.data for x in nbr]),
                                           _abi.ptr_table([x.ctypes.data for x in itp]),
                                           _abi.ptr_table([x.ctypes.data for x in order]), out.ctypes.data, ws.ctypes.data,
                                           wsb, None, None)
        return rc, out
    rc = L.ml3d_randla_forward(C.byref(desc), params.ctypes.data, feats.ctypes.data, points.ctypes.data,
                               _abi.ptr_table([x.ctypes.data for x in nbr]), _abi.ptr_table([x.ctypes.data for x in itp]),
                               out.ctypes.data, ws.ctypes.data, wsb, None)
    return rc, out


def _ws(nbytes):
    return np.zeros(int(nbytes) + 64, np.uint8)


def radius(points, psplits, queries, qsplits, r, dense=False, local=False, with_d2=False, spill_phase=0):
    """-> (neighbors_index, row_splits[, d2]) ragged, or the dense [Nq, max] matrix padded with Ns."""
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    ps = np.ascontiguousarray(psplits, np.int64)
    qs = np.ascontiguousarray(qsplits, np.int64)
    B, ns, nq = len(ps) - 1, len(points), len(queries)
    rs = np.zeros(nq + 1, np.int64)
    stats = np.zeros(2, np.int64)
    wsb = L.ml3d_radius_workspace_bytes(ns, nq, B, 0)
    ws = _ws(wsb)
    rc = L.ml3d_radius_count(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, ns, nq, r,
                             rs.ctypes.data, stats.ctypes.data, ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    total, longest = int(stats[0]), int(stats[1])
    assert total == rs[-1]
    cols = longest if dense else 0
    idx = np.full((nq, cols) if dense else (total,), -5, np.int32)
    d2 = np.zeros(idx.shape, np.float32) if with_d2 else None
    if spill_phase is None:
        # spill == NULL: long rows sort at the tail of ONE workspace sized for the result; the count phase is repeated into it
        wsb2 = L.ml3d_radius_workspace_bytes(ns, nq, B, total)
        ws2 = _ws(wsb2)
        rc = L.ml3d_radius_count(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, ns, nq, r,
                                 rs.ctypes.data, stats.ctypes.data, ws2.ctypes.data, wsb2, None)
        assert rc == 0, rc
        rc = L.ml3d_radius_fill(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, ns, nq, r,
                                rs.ctypes.data, total, 1 if local else 0, cols, ns, idx.ctypes.data,
                                None if d2 is None else d2.ctypes.data, ws2.ctypes.data, wsb2, None, 0, None)
        assert rc == 0, rc
        # ... and a workspace WITHOUT room for the rows is refused instead of overrun
        if total > 0:
            rc = L.ml3d_radius_fill(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, ns, nq, r,
                                    rs.ctypes.data, total, 1 if local else 0, cols, ns, idx.ctypes.data,
                                    None if d2 is None else d2.ctypes.data, ws.ctypes.data, wsb, None, 0, None)
            assert rc != 0
    else:
        # the workspace of the count phase (it carries the grid) goes back in untouched; long rows sort in a separate spill buffer
        spill = np.zeros(total + 2, np.uint64)
        rc = L.ml3d_radius_fill(points.ctypes.data, ps.ctypes.data, queries.ctypes.data, qs.ctypes.data, B, ns, nq, r,
                                rs.ctypes.data, total, 1 if local else 0, cols, ns, idx.ctypes.data,
                                None if d2 is None else d2.ctypes.data, ws.ctypes.data, wsb, spill.ctypes.data + spill_phase,
                                8 * total + 8, None)
        assert rc == 0, rc
    if dense:
        return idx
    return (idx, rs, d2) if with_d2 else (idx, rs)


def ragged_to_dense(values, row_splits, cols, default):
    L = lib()
    values = np.ascontiguousarray(values)
    rs = np.ascontiguousarray(row_splits, np.int64)
    inner = values.shape[1:]
    dv = np.ascontiguousarray(np.broadcast_to(np.asarray(default, values.dtype), inner))
    elem = int(values.dtype.itemsize * int(np.prod(inner, dtype=np.int64)))
    out = np.empty((len(rs) - 1, cols) + tuple(inner), values.dtype)
    rc = L.ml3d_ragged_to_dense(values.ctypes.data, rs.ctypes.data, len(rs) - 1, cols, elem, dv.ctypes.data,
                                out.ctypes.data, None)
    assert rc == 0, rc
    return out


def voxelize(points, row_splits, vs, rmin, rmax, max_points=2**62, max_voxels=2**62):
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    rs = np.ascontiguousarray(row_splits, np.int64)
    vs, rmin, rmax = (np.ascontiguousarray(x, np.float32) for x in (vs, rmin, rmax))
    B, n = len(rs) - 1, len(points)
    stride = points.shape[1]
    wsb = L.ml3d_voxelize_workspace_bytes(n, B)
    ws = _ws(wsb)
    bs = np.zeros(B + 1, np.int64)
    stats = np.zeros(2, np.int64)
    rc = L.ml3d_voxelize_count(points.ctypes.data, stride, rs.ctypes.data, B, n, vs.ctypes.data, rmin.ctypes.data,
                               rmax.ctypes.data, max_points, max_voxels, bs.ctypes.data, stats.ctypes.data,
                               ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    M, K = int(stats[0]), int(stats[1])
    coords = np.full((M, 3), -1, np.int32)
    pidx = np.full(K, -1, np.int64)
    prs = np.full(M + 1, -1, np.int64)
    rc = L.ml3d_voxelize_fill(B, n, vs.ctypes.data, rmin.ctypes.data, rmax.ctypes.data, max_points, max_voxels,
                              bs.ctypes.data, coords.ctypes.data, pidx.ctypes.data, prs.ctypes.data, ws.ctypes.data,
                              wsb, None)
    assert rc == 0, rc
    return coords, pidx, prs, bs


def subsample_batch(points, lengths, dl, features=None, labels=None):
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    rs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    B, n = len(rs) - 1, len(points)
    feats = None if features is None else np.ascontiguousarray(features, np.float32)
    labs = None if labels is None else np.ascontiguousarray(labels, np.int32)
    wsb = L.ml3d_subsample_workspace_bytes(n, B)
    ws = _ws(wsb)
    lens = np.zeros(B, np.int64)
    stats = np.zeros(2, np.int64)
    rc = L.ml3d_subsample_count(points.ctypes.data, rs.ctypes.data, B, n, dl, lens.ctypes.data, stats.ctypes.data,
                                ws.ctypes.data, wsb, None)
    assert rc == 0 and stats[1] == 0, (rc, stats)
    M = int(stats[0])
    op = np.zeros((M, 3), np.float32)
    fd = 0 if feats is None else feats.shape[1]
    of = None if feats is None else np.zeros((M, fd), np.float32)
    ol = None if labs is None else np.zeros(M, np.int32)
    rc = L.ml3d_subsample_fill(points.ctypes.data, None if feats is None else feats.ctypes.data, fd,
                               None if labs is None else labs.ctypes.data, B, n, op.ctypes.data,
                               None if of is None else of.ctypes.data, None if ol is None else ol.ctypes.data,
                               ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    return op, lens, of, ol


def subsample_items(points, lengths, dl):
    """ml3d_subsample_items_count / _fill (one workgroup per batch item) -> (points, lengths, status); status 2 = an item's grid is
    larger than the kernel's bitmap, -4 = an item has more points than the kernel takes (the caller's cue for the sort-based op)."""
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    rs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    B, n = len(rs) - 1, len(points)
    lens = np.zeros(B, np.int64)
    stats = np.zeros(2, np.int64)
    rc = L.ml3d_subsample_items_count(points.ctypes.data, rs.ctypes.data, B, n, dl, int(max(list(lengths) + [0])), lens.ctypes.data,
                                      stats.ctypes.data, None)
    if rc == -4:          # ML3D_E_UNSUPPORTED
        return None, None, -4
    assert rc == 0, rc
    if stats[1]:
        return None, None, int(stats[1])
    op = np.zeros((int(stats[0]), 3), np.float32)
    rc = L.ml3d_subsample_items_fill(points.ctypes.data, rs.ctypes.data, B, n, dl, int(max(list(lengths) + [0])), lens.ctypes.data,
                                     op.ctypes.data, None)
    assert rc == 0, rc
    return op, lens, 0


def kpconv_rigid(q_pts, s_pts, inds, x, kp, weights, extent, bias=None, act=0, slope=0.0, influence=1, offset_features=None,
                 bf16x3=False):
    L = lib()
    q_pts, s_pts, x = (np.ascontiguousarray(a, np.float32) for a in (q_pts, s_pts, x))
    inds = np.ascontiguousarray(inds, np.int32)
    kp = np.ascontiguousarray(kp, np.float32)
    K, cin, cout = weights.shape
    w = np.ascontiguousarray(weights.reshape(K * cin, cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    nq, H = inds.shape
    out = np.zeros((nq, cout), np.float32)
    wsb = L.ml3d_kpconv_workspace_bytes(nq, cin, cout, K)
    ws = _ws(wsb)
    if offset_features is not None:
        off = np.ascontiguousarray(offset_features, np.float32)
        rc = L.ml3d_kpconv_deformable(q_pts.ctypes.data, s_pts.ctypes.data, inds.ctypes.data, nq, len(s_pts), H,
                                      x.ctypes.data, cin, kp.ctypes.data, K, extent, influence, off.ctypes.data,
                                      off.shape[1], w.ctypes.data, None if b is None else b.ctypes.data, act, slope, cout,
                                      out.ctypes.data, ws.ctypes.data, wsb, None)
        return rc, out
    if bf16x3:
        rcp, packed = pack_bf16x3(w)
        if rcp != 0:
            return rcp, None
        rc = L.ml3d_kpconv_rigid_bf16x3(q_pts.ctypes.data, s_pts.ctypes.data, inds.ctypes.data, nq, len(s_pts), H, x.ctypes.data,
                                        cin, kp.ctypes.data, K, extent, influence, w.ctypes.data, packed.ctypes.data,
                                        None if b is None else b.ctypes.data, act, slope, cout, out.ctypes.data, ws.ctypes.data, wsb,
                                        None)
        return rc, out
    rc = L.ml3d_kpconv_rigid(q_pts.ctypes.data, s_pts.ctypes.data, inds.ctypes.data, nq, len(s_pts), H, x.ctypes.data,
                             cin, kp.ctypes.data, K, extent, influence, w.ctypes.data,
                             None if b is None else b.ctypes.data, act, slope, cout, out.ctypes.data, ws.ctypes.data, wsb,
                             None)
    return rc, out


def linear(a, wt, bias=None, a2=None, gather=None, gather_stride=1, residual=None, act=0, slope=0.0, m=None,
           residual_gather=None):
    L = lib()
    a = np.ascontiguousarray(a, np.float32)
    wt = np.ascontiguousarray(wt, np.float32)
    k1 = a.shape[1]
    k2 = 0 if a2 is None else a2.shape[1]
    a2c = None if a2 is None else np.ascontiguousarray(a2, np.float32)
    n = wt.shape[1]
    g = None if gather is None else np.ascontiguousarray(gather, np.int32)
    if m is None:
        m = a.shape[0] if g is None else (g.size // gather_stride)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if residual is None else np.ascontiguousarray(residual, np.float32)
    rg = None if residual_gather is None else np.ascontiguousarray(residual_gather, np.int32)
    out = np.zeros((m, n), np.float32)
    wsb = L.ml3d_linear_workspace_bytes(m, n, k1 + k2)
    ws = _ws(wsb)
    rc = L.ml3d_linear(a.ctypes.data, k1, k1, None if g is None else g.ctypes.data, gather_stride, a.shape[0],
                       None if a2c is None else a2c.ctypes.data, k2, k2, wt.ctypes.data,
                       None if b is None else b.ctypes.data, None if r is None else r.ctypes.data, n,
                       None if rg is None else rg.ctypes.data, 0 if rg is None else (rg.shape[1] if rg.ndim == 2 else 1),
                       0 if r is None else r.shape[0], act, slope, out.ctypes.data, n, m, n, ws.ctypes.data, wsb, None)
    return rc, out


def gather_pool(x, inds, mode):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    inds = np.ascontiguousarray(inds, np.int32)
    out = np.zeros((inds.shape[0], x.shape[1]), np.float32)
    rc = L.ml3d_gather_pool(x.ctypes.data, x.shape[0], x.shape[1], inds.ctypes.data, inds.shape[0], inds.shape[1], mode,
                            out.ctypes.data, None)
    assert rc == 0, rc
    return out


def pillar_features(points, vox, in_ch, P, vx, vy, x_off, y_off, nx, ny, layers, batch):
    """vox = (coords, pidx, prs, bs) from voxelize(); layers = [(wt [cin, units], bias [units]), ...]."""
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    coords, pidx, prs, bs = vox
    units = (C.c_int32 * len(layers))(*[int(w.shape[1]) for w, _ in layers])
    ws_ = [np.ascontiguousarray(w, np.float32) for w, _ in layers]
    bs_ = [np.ascontiguousarray(b, np.float32) for _, b in layers]
    cc = ws_[-1].shape[1]
    canvas = np.full((batch, ny, nx, cc), 7.0, np.float32)
    M = len(coords)
    wsb = L.ml3d_pillar_features_workspace_bytes(M, P, len(layers), units)
    ws = _ws(wsb)
    rc = L.ml3d_pillar_features(points.ctypes.data, points.shape[1], in_ch, coords.ctypes.data, pidx.ctypes.data,
                                prs.ctypes.data, bs.ctypes.data, batch, M, P, vx, vy, x_off, y_off, nx, ny, len(layers),
                                units, _abi.ptr_table([w.ctypes.data for w in ws_]),
                                _abi.ptr_table([b.ctypes.data for b in bs_]), canvas.ctypes.data, cc, ws.ctypes.data, wsb,
                                None)
    return rc, canvas


def conv2d_nhwc(x, w, bias, stride, pad, act=2, kh=3, kw=3):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, H, W, Cin = x.shape
    w = np.ascontiguousarray(w, np.float32)
    cout = w.shape[1]
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = np.zeros((B, OH, OW, cout), np.float32)
    b = np.ascontiguousarray(bias, np.float32)
    wsb = L.ml3d_conv2d_workspace_bytes(B, OH, OW, Cin, cout, kh, kw)
    ws = _ws(wsb)
    rc = L.ml3d_conv2d_nhwc(x.ctypes.data, B, H, W, Cin, w.ctypes.data, b.ctypes.data, kh, kw, stride, pad, act, 0.0, cout,
                            out.ctypes.data, cout, ws.ctypes.data, wsb, None)
    return rc, out


def pack_bf16x3(w):
    """[K, N] float -> the three bf16 planes of ml3d_gemm_pack_bf16x3 (uint8 array), or (rc, None)."""
    L = lib()
    w = np.ascontiguousarray(w, np.float32)
    K, N = w.shape
    nbytes = int(L.ml3d_gemm_pack_bf16x3_bytes(K, N))
    buf = np.zeros(max(nbytes, 16) + 16, np.uint8)
    off = (-buf.ctypes.data) % 16
    packed = buf[off:off + max(nbytes, 16)]
    rc = L.ml3d_gemm_pack_bf16x3(w.ctypes.data, K, N, packed.ctypes.data, nbytes, None)
    return rc, packed


def conv2d_nhwc_bf16x3(x, w, bias, stride, pad, act=2, kh=3, kw=3, out=None, ch_off=0):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, H, W, Cin = x.shape
    cout = w.shape[1]
    rc, packed = pack_bf16x3(w)
    if rc != 0:
        return rc, None
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = np.zeros((B, OH, OW, cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = L.ml3d_conv2d_nhwc_bf16x3(x.ctypes.data, B, H, W, Cin, packed.ctypes.data, None if b is None else b.ctypes.data, kh, kw,
                                   stride, pad, act, 0.0, cout, out.ctypes.data + 4 * ch_off, out.shape[3], None)
    return rc, out


def linear_bf16x3(a, w, bias, act=0, a2=None, residual=None, residual_gather=None):
    L = lib()
    a = np.ascontiguousarray(a, np.float32)
    rc, packed = pack_bf16x3(w)
    if rc != 0:
        return rc, None
    m, k1 = a.shape
    k2 = 0 if a2 is None else a2.shape[1]
    a2 = None if a2 is None else np.ascontiguousarray(a2, np.float32)
    res = None if residual is None else np.ascontiguousarray(residual, np.float32)
    n = w.shape[1]
    out = np.zeros((m, n), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    wsb = L.ml3d_linear_bf16x3_workspace_bytes(m, n, k1 + k2)
    ws = _ws(wsb)
    if residual_gather is not None:
        rg = np.ascontiguousarray(residual_gather, np.int32)
        rc = L.ml3d_linear_bf16x3_gathered(a.ctypes.data, k1, k1, None if a2 is None else a2.ctypes.data, k2, k2, m, packed.ctypes.data,
                                           None if b is None else b.ctypes.data, res.ctypes.data, n, rg.ctypes.data,
                                           rg.shape[1] if rg.ndim == 2 else 1, res.shape[0], n, act, 0.0, out.ctypes.data, n,
                                           ws.ctypes.data, wsb, None)
        return rc, out
    rc = L.ml3d_linear_bf16x3(a.ctypes.data, k1, k1, None if a2 is None else a2.ctypes.data, k2, k2, m, packed.ctypes.data,
                              None if b is None else b.ctypes.data, None if res is None else res.ctypes.data, n, n, act, 0.0,
                              out.ctypes.data, n, ws.ctypes.data, wsb, None)
    return rc, out


def deconv2d_nhwc_bf16x3(x, w, bias, stride, cout, out=None, ch_off=0, act=2):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, H, W, Cin = x.shape
    rc, packed = pack_bf16x3(w)
    if rc != 0:
        return rc, None
    b = np.ascontiguousarray(bias, np.float32)
    if out is None:
        out = np.zeros((B, H * stride, W * stride, cout), np.float32)
    rc = L.ml3d_deconv2d_nhwc_bf16x3(x.ctypes.data, B, H, W, Cin, packed.ctypes.data, b.ctypes.data, stride, act, 0.0, cout,
                                     out.ctypes.data + 4 * ch_off, out.shape[3], None)
    return rc, out


def deconv2d_nhwc(x, w, bias, stride, cout, out=None, ch_off=0, act=2):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, H, W, Cin = x.shape
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(bias, np.float32)
    if out is None:
        out = np.zeros((B, H * stride, W * stride, cout), np.float32)
    ld = out.shape[3]
    wsb = L.ml3d_conv2d_workspace_bytes(B, H, W, Cin, stride * stride * cout, 1, 1)
    ws = _ws(wsb)
    rc = L.ml3d_deconv2d_nhwc(x.ctypes.data, B, H, W, Cin, w.ctypes.data, b.ctypes.data, stride, act, 0.0, cout,
                              out.ctypes.data + 4 * ch_off, ld, ws.ctypes.data, wsb, None)
    return rc, out


def nhwc_to_nchw(x, c0, c):
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    B, H, W, ld = x.shape
    out = np.zeros((B, c, H, W), np.float32)
    rc = L.ml3d_nhwc_to_nchw(x.ctypes.data, ld, c0, c, B, H * W, out.ctypes.data, None)
    assert rc == 0, rc
    return out


def nms(boxes, scores, thr):
    L = lib()
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    n = len(boxes)
    keep = np.full(max(n, 1), -1, np.int64)
    cnt = np.zeros(1, np.int64)
    wsb = L.ml3d_nms_workspace_bytes(n)
    ws = _ws(wsb)
    rc = L.ml3d_nms(boxes.ctypes.data, scores.ctypes.data, n, thr, keep.ctypes.data, cnt.ctypes.data, ws.ctypes.data, wsb,
                    None)
    assert rc == 0, rc
    return keep[:int(cnt[0])]


def box_iou(boxes_a, boxes_b, mode3d):
    """ml3d_iou_bev ([n,5] / [m,5]) or ml3d_iou_3d ([n,7] / [m,7]) -> float32 [n, m]."""
    L = lib()
    a, b = np.ascontiguousarray(boxes_a, np.float32), np.ascontiguousarray(boxes_b, np.float32)
    out = np.full((len(a), len(b)), -7, np.float32)
    fn = L.ml3d_iou_3d if mode3d else L.ml3d_iou_bev
    rc = fn(a.ctypes.data, b.ctypes.data, len(a), len(b), out.ctypes.data, None)
    assert rc == 0, rc
    return out


def topk_rows(values, k, want_values=True):
    """ml3d_topk_rows on a [rows, n] float32 array -> (index int64 [rows, k], value float32 [rows, k] | None)."""
    L = lib()
    v = np.ascontiguousarray(values, np.float32)
    rows, n = v.shape
    idx = np.full((rows, k), -7, np.int64)
    val = np.full((rows, k), -7, np.float32) if want_values else None
    wsb = L.ml3d_topk_rows_workspace_bytes(rows, n, k)
    ws = _ws(max(wsb, 1))
    rc = L.ml3d_topk_rows(v.ctypes.data, rows, n, k, idx.ctypes.data, val.ctypes.data if want_values else None,
                          ws.ctypes.data, wsb, None)
    assert rc == 0, rc
    return idx, val


def nearest_to_center(points, center, k):
    L = lib()
    points = np.ascontiguousarray(points, np.float32)
    c = np.ascontiguousarray(center, np.float32)
    n = len(points)
    idx = np.full(k, -1, np.int32)
    d2 = np.zeros(k, np.float64)
    wsb = L.ml3d_nearest_to_center_workspace_bytes(n)
    ws = _ws(wsb)
    rc = L.ml3d_nearest_to_center(points.ctypes.data, n, c.ctypes.data, k, idx.ctypes.data, d2.ctypes.data, ws.ctypes.data,
                                  wsb, None)
    assert rc == 0, rc
    return idx, d2


def vote_update(probs16, inds, logits, smooth):
    L = lib()
    probs16 = np.ascontiguousarray(probs16, np.float16).copy()
    inds = np.ascontiguousarray(inds, np.int32)
    logits = np.ascontiguousarray(logits, np.float32)
    rc = L.ml3d_vote_update(logits.ctypes.data, inds.ctypes.data, len(inds), probs16.shape[1], smooth, probs16.ctypes.data,
                            probs16.shape[0], None)
    assert rc == 0, rc
    return probs16


def argmax_labels(scores):
    L = lib()
    scores = np.ascontiguousarray(scores, np.float32)
    n = scores.size // scores.shape[-1]
    out = np.full(n, 255, np.uint8)
    rc = L.ml3d_argmax_labels(scores.ctypes.data, n, scores.shape[-1], out.ctypes.data, None)
    assert rc == 0, rc
    return out.reshape(scores.shape[:-1])
