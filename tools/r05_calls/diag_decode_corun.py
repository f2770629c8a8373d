"""Which co-runner disturbs the PointPillars decode tail?  Fixed head maps (a quiet forward of 8 sweeps); on stream A 40 decodes
(anchor scores -> top-k -> decode + batched NMS), on stream B one of: nothing / the forward of 8 other sweeps (bf16x3 or f32 per
ML3D_PP_CONV) / a torch matmul loop / a fill loop.  Every decode's smax, candidate list and rows are kept and compared with the quiet ones."""
import os, sys, ctypes as C
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
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
anchors = m.bbox_head._anchors_for(tuple(views[0].shape[-2:]), heads.device)
h = m.bbox_head

def decode():
    cls, s_cls = D._head_map(views[0])
    Bn, AC, H, W = cls.shape
    A = views[2].shape[1] // 2
    Cc = AC // A
    n_anchor = H * W * A
    strides = (C.c_int64 * 9)(*[int(v) for v in s_cls + D._head_map(views[1])[1] + D._head_map(views[2])[1]])
    smax = torch.empty((Bn, n_anchor), dtype=torch.float32, device=heads.device)
    rc = lib.ml3d_pp_anchor_scores(cls.data_ptr(), strides, Bn, A, Cc, H * W, smax.data_ptr(), D._stream())
    assert rc == 0
    cand = D.topk_rows(smax, int(h.nms_pre))
    rows, total = ops.pointpillars_boxes(views[0], views[1], views[2], anchors, h.nms_pre, h.score_thr, 0.01, h.dir_offset)
    return smax, cand, rows, total

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(sa):
    ref = decode()
torch.cuda.synchronize()
big = torch.randn(4096, 4096, device="cuda")
fillbuf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")

def corun(kind):
    if kind == "forward":
        m.head_maps_nhwc(B8)
    elif kind == "matmul":
        for _ in range(12):
            big @ big
    elif kind == "fill":
        for _ in range(40):
            fillbuf.zero_()
    elif kind == "voxelize":
        for _ in range(6):
            m.voxel_layer.voxelize_batch(B8)

for kind in ("none", "forward", "matmul", "fill", "voxelize", "forward"):
    outs = []
    with torch.no_grad():
        for rep in range(10):
            with torch.cuda.stream(sb):
                corun(kind)
            with torch.cuda.stream(sa):
                for _ in range(4):
                    outs.append(decode())
        torch.cuda.synchronize()
    bad = [0, 0, 0, 0]
    for o in outs:
        bad[0] += int(not torch.equal(o[0], ref[0]))
        bad[1] += int(not torch.equal(o[1], ref[1]))
        bad[3] += int(not torch.equal(o[3], ref[3]))
        same_rows = torch.equal(o[3], ref[3]) and all(torch.equal(o[2][b, :int(ref[3][b])], ref[2][b, :int(ref[3][b])]) for b in range(8))
        bad[2] += int(not same_rows)
    print("co-runner %-9s: of %d decodes differ in smax %d, candidates %d, rows %d, totals %d" % (kind, len(outs), *bad), flush=True)
    if bad[1]:
        for o in outs:
            if not torch.equal(o[1], ref[1]):
                d = (o[1] != ref[1])
                print("   first differing candidate lists: sample rows", d.any(1).nonzero().flatten().tolist(), "positions", d.nonzero()[:6].tolist(),
                      "got", o[1][d][:6].tolist(), "want", ref[1][d][:6].tolist())
                break
