"""The product's decode sequence (anchor scores -> top-k -> ml3d_pp_boxes, workspaces from torch.empty and FREED after every call, like
ops.pointpillars_boxes) on stream A under a co-running bf16x3 forward on stream B; a copy of every call's candidate list and of its
pp_boxes workspace is kept and compared with a quiet decode.  ML3D_DIAG_LIB picks the library (old = hipMemsetAsync, new = fill kernel),
ML3D_DIAG_WS=zeros zero-fills the workspaces."""
import os, sys, ctypes as C
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
import numpy as np
import torch
import synth_data
from oracle import pointpillars_ref as P
from ml3d import _abi
if os.environ.get("ML3D_DIAG_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["ML3D_DIAG_LIB"])
from ml3d.torch.models import PointPillars
from ml3d.ops import detection as D
alloc = torch.zeros if os.environ.get("ML3D_DIAG_WS") == "zeros" else torch.empty

cfg = P.KITTI_CFG
m = PointPillars(device="cuda:0", **cfg)
m.load_state_dict(P.make_state_dict(cfg, 2024))
m.eval()
A8 = [torch.from_numpy(P.crop_for_cfg(synth_data.kitti_sweep(i), cfg)).cuda() for i in range(16, 24)]
B8 = [torch.from_numpy(P.crop_for_cfg(synth_data.kitti_sweep(i), cfg)).cuda() for i in range(24, 32)]
lib = _abi.get()
with torch.no_grad():
    heads, split = m.head_maps_nhwc(A8)
    heads = heads.clone()
    torch.cuda.synchronize()
nchw = heads.permute(0, 3, 1, 2)
views, off = [], 0
for c in split:
    views.append(nchw[:, off:off + c]); off += c
anchors = m.bbox_head._anchors_for(tuple(views[0].shape[-2:]), heads.device).contiguous().float()
h = m.bbox_head
(cls, s_cls), (reg, s_reg), (dr, s_dir) = (D._head_map(t) for t in views)
Bn, AC, H, W = cls.shape
A = dr.shape[1] // 2
Cc = AC // A
k = int(h.nms_pre)
n_anchor = H * W * A
strides = (C.c_int64 * 9)(*[int(v) for v in s_cls + s_reg + s_dir])
wsb = lib.ml3d_pp_boxes_workspace_bytes(Bn, k, Cc)
twsb = lib.ml3d_topk_rows_workspace_bytes(Bn, n_anchor, k)
al = lambda x: (x + 255) & ~255
Pn, words = Bn * Cc, (k + 63) // 64
regions, o = {}, 0
for name, nbytes in (("order", 4 * Pn * k), ("nvalid", 4 * Pn), ("mask", 8 * Pn * k * words), ("keep", 4 * Pn * k), ("count", 4 * Pn),
                     ("box", 28 * Bn * k), ("bev", 20 * Bn * k), ("score", 4 * Pn * k), ("dirbit", 4 * Bn * k)):
    regions[name] = (o, nbytes); o += al(nbytes)

def decode():
    st = D._stream()
    smax = alloc((Bn, n_anchor), dtype=torch.float32, device="cuda")
    assert lib.ml3d_pp_anchor_scores(cls.data_ptr(), strides, Bn, A, Cc, H * W, smax.data_ptr(), st) == 0
    cand = alloc((Bn, k), dtype=torch.int64, device="cuda")
    tws = alloc(twsb, dtype=torch.uint8, device="cuda")
    assert lib.ml3d_topk_rows(smax.data_ptr(), Bn, n_anchor, k, cand.data_ptr(), None, tws.data_ptr(), twsb, st) == 0
    rows = alloc((Bn, Cc * k, 9), dtype=torch.float32, device="cuda")
    total = alloc(Bn, dtype=torch.int32, device="cuda")
    ws = alloc(wsb, dtype=torch.uint8, device="cuda")
    assert lib.ml3d_pp_boxes(cls.data_ptr(), reg.data_ptr(), dr.data_ptr(), strides, anchors.data_ptr(), cand.data_ptr(), Bn, k, A, Cc, H * W,
                             float(h.score_thr), 0.01, float(h.dir_offset), rows.data_ptr(), total.data_ptr(), ws.data_ptr(), wsb, st) == 0
    base = (-ws.data_ptr()) % 256
    return cand.clone(), ws[base:].clone(), total.clone()       # (the originals are freed here: the next call reuses their blocks)

def parts(wsc):
    return {name: wsc[o: o + nb].cpu().numpy().copy() for name, (o, nb) in regions.items()}

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(sa):
    ref = decode()
torch.cuda.synchronize()
rp = parts(ref[1])
nvalid, count = rp["nvalid"].view(np.int32), rp["count"].view(np.int32)

def cmp(o):
    p = parts(o[1])
    d = {"cand": int((o[0] != ref[0]).sum()), "total": int((o[2] != ref[2]).sum())}
    for name in ("box", "bev", "score", "dirbit", "nvalid", "count"):
        d[name] = int((p[name] != rp[name]).sum())
    od, orf = p["order"].view(np.uint32).reshape(Pn, k), rp["order"].view(np.uint32).reshape(Pn, k)
    d["order"] = sum(int((od[q, :nvalid[q]] != orf[q, :nvalid[q]]).sum()) for q in range(Pn))
    mk, mrf = p["mask"].view(np.uint64).reshape(Pn, k, words), rp["mask"].view(np.uint64).reshape(Pn, k, words)
    bm = [(q, a, w, hex(int(mk[q, a, w])), hex(int(mrf[q, a, w]))) for q in range(Pn) for a in range(int(nvalid[q])) for w in range(a >> 6, words)
          if mk[q, a, w] != mrf[q, a, w]]
    d["mask"] = len(bm)
    kp, krf = p["keep"].view(np.int32).reshape(Pn, k), rp["keep"].view(np.int32).reshape(Pn, k)
    d["keep"] = sum(int((kp[q, :count[q]] != krf[q, :count[q]]).sum()) for q in range(Pn))
    return d, bm

for kind in ("none", "forward", "forward"):
    outs = []
    with torch.no_grad():
        for rep in range(12):
            if kind == "forward":
                with torch.cuda.stream(sb):
                    m.head_maps_nhwc(B8)
            with torch.cuda.stream(sa):
                for _ in range(4):
                    outs.append(decode())
        torch.cuda.synchronize()
    nbad = 0
    for i, o in enumerate(outs):
        d, bm = cmp(o)
        if any(d.values()):
            nbad += 1
            if nbad <= 3:
                print("  ", kind, "decode", i, "differs:", {a: b for a, b in d.items() if b}, "mask (problem, row, word, got, want):", bm[:4], flush=True)
    print(os.environ.get("ML3D_DIAG_LIB", "product lib"), os.environ.get("ML3D_DIAG_WS", "empty"), "| co-runner", kind, ":", nbad, "of", len(outs), "decodes differ", flush=True)
