"""Which kernel of ml3d_pp_boxes is disturbed when the bf16x3 forward co-runs on another stream?  The workspace of every decode is
kept and its regions (box / bev / score / dirbit of pp_decode, order / nvalid of nmsb_order, the defined mask words of nmsb_mask,
keep / count of nmsb_reduce) are compared with a quiet decode of the same maps."""
import os, sys, ctypes as C
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
import numpy as np
import torch
import synth_data
from oracle import pointpillars_ref as P
from ml3d.torch.models import PointPillars
from ml3d import ops, _abi
from ml3d.ops import detection as D

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
strides = (C.c_int64 * 9)(*[int(v) for v in s_cls + s_reg + s_dir])
smax = torch.empty((Bn, H * W * A), dtype=torch.float32, device="cuda")
assert lib.ml3d_pp_anchor_scores(cls.data_ptr(), strides, Bn, A, Cc, H * W, smax.data_ptr(), D._stream()) == 0
cand = D.topk_rows(smax, k)
torch.cuda.synchronize()
wsb = lib.ml3d_pp_boxes_workspace_bytes(Bn, k, Cc)
al = lambda x: (x + 255) & ~255
Pn, words = Bn * Cc, (k + 63) // 64
regions, o = {}, 0
for name, nbytes in (("order", 4 * Pn * k), ("nvalid", 4 * Pn), ("mask", 8 * Pn * k * words), ("keep", 4 * Pn * k), ("count", 4 * Pn),
                     ("box", 28 * Bn * k), ("bev", 20 * Bn * k), ("score", 4 * Pn * k), ("dirbit", 4 * Bn * k)):
    regions[name] = (o, nbytes); o += al(nbytes)

def decode():
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    rows = torch.zeros((Bn, Cc * k, 9), dtype=torch.float32, device="cuda")
    total = torch.zeros(Bn, dtype=torch.int32, device="cuda")
    rc = lib.ml3d_pp_boxes(cls.data_ptr(), reg.data_ptr(), dr.data_ptr(), strides, anchors.data_ptr(), cand.data_ptr(), Bn, k, A, Cc, H * W,
                           float(h.score_thr), 0.01, float(h.dir_offset), rows.data_ptr(), total.data_ptr(), ws.data_ptr(), wsb, D._stream())
    assert rc == 0
    return ws, rows, total

def parts(ws):
    base = (-ws.data_ptr()) % 256
    out = {}
    for name, (o, nb) in regions.items():
        out[name] = ws[base + o: base + o + nb].cpu().numpy().copy()
    return out

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(sa):
    ref = decode()
torch.cuda.synchronize()
rp = parts(ref[0])
nvalid = rp["nvalid"].view(np.int32)
count = rp["count"].view(np.int32)
print("nvalid", nvalid.tolist()); print("count", count.tolist())

def cmp(p):
    d = {}
    for name in ("box", "bev", "score", "dirbit", "nvalid", "count"):
        d[name] = int((p[name] != rp[name]).sum())
    od, orf = p["order"].view(np.uint32).reshape(Pn, k), rp["order"].view(np.uint32).reshape(Pn, k)
    d["order"] = sum(int((od[q, :nvalid[q]] != orf[q, :nvalid[q]]).sum()) for q in range(Pn))
    mk, mrf = p["mask"].view(np.uint64).reshape(Pn, k, words), rp["mask"].view(np.uint64).reshape(Pn, k, words)
    bad_mask = []
    for q in range(Pn):
        for a in range(int(nvalid[q])):
            for w in range(a >> 6, words):
                if mk[q, a, w] != mrf[q, a, w]:
                    bad_mask.append((q, a, w, hex(int(mk[q, a, w])), hex(int(mrf[q, a, w]))))
    d["mask"] = len(bad_mask)
    kp, krf = p["keep"].view(np.int32).reshape(Pn, k), rp["keep"].view(np.int32).reshape(Pn, k)
    d["keep"] = sum(int((kp[q, :count[q]] != krf[q, :count[q]]).sum()) for q in range(Pn))
    return d, bad_mask

for kind in ("none", "forward", "forward", "forward"):
    outs = []
    with torch.no_grad():
        for rep in range(10):
            if kind == "forward":
                with torch.cuda.stream(sb):
                    m.head_maps_nhwc(B8)
            with torch.cuda.stream(sa):
                for _ in range(4):
                    outs.append(decode())
        torch.cuda.synchronize()
    nbad = 0
    for i, o in enumerate(outs):
        d, bm = cmp(parts(o[0]))
        if any(d.values()):
            nbad += 1
            if nbad <= 4:
                print(kind, "decode", i, "differs:", {a: b for a, b in d.items() if b}, "mask words (problem, row, word, got, want):", bm[:6], flush=True)
    print("co-runner", kind, ":", nbad, "of", len(outs), "decodes differ", flush=True)
