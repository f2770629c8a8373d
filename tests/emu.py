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


def randla_forward(cfg, sd, points, feats, nbr, itp):
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
    rc = L.ml3d_randla_forward(C.byref(desc), params.ctypes.data, feats.ctypes.data, points.ctypes.data,
                               _abi.ptr_table([x.ctypes.data for x in nbr]), _abi.ptr_table([x.ctypes.data for x in itp]),
                               out.ctypes.data, ws.ctypes.data, wsb, None)
    return rc, out
