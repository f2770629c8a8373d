"""Localise the rare two-lane mismatch (tools/r05_calls/diag_two_lane.py: identical maps outside the pipeline, one sweep of 144
with 179 instead of 181 boxes inside it): every detect() call of the pipeline stashes its head maps and its (rows, total); afterwards
the stashed maps are compared with a quiet-GPU forward of the same sweeps, the stashed rows with a quiet-GPU decode of the stashed maps
and with what the pipeline handed out."""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
import torch
import synth_data
from oracle import pointpillars_ref as P
from ml3d.torch.models import PointPillars
from ml3d.engine import PointPillarsStream

cfg = P.KITTI_CFG
m = PointPillars(device="cuda:0", **cfg)
m.load_state_dict(P.make_state_dict(cfg, 2024))
m.eval()
steps = [[P.crop_for_cfg(synth_data.kitti_sweep(i), cfg) for i in range(16 * s, 16 * s + 16)] for s in range(3)]
hosts = [[torch.from_numpy(c).pin_memory() for c in st] for st in steps]
dev = [[torch.from_numpy(c).cuda() for c in st] for st in steps]
with torch.no_grad():
    quiet = []
    for st in dev:
        heads, split = m.head_maps_nhwc(st)
        quiet.append(heads.clone())
    torch.cuda.synchronize()

stash = []
orig_detect = m.detect

def detect(inputs):
    heads, split = m.head_maps_nhwc(inputs)
    nchw = heads.permute(0, 3, 1, 2)
    views, off = [], 0
    for c in split:
        views.append(nchw[:, off:off + c]); off += c
    rows, total = m.bbox_head.boxes_device(*views)
    stash.append((len(inputs), heads, rows.clone(), total.clone(), torch.cuda.current_stream().cuda_stream))
    return rows, total

m.detect = detect
bad_total = 0
for rep, (lanes, threaded) in enumerate([(1, False)] + [(2, False)] * 8 + [(2, True)] * 4):
    stash.clear()
    pipe = PointPillarsStream(m, "cuda", lanes=lanes, threaded=threaded)
    got = [pipe.submit(h) for h in hosts] + [pipe.flush()]
    got = got[1:]
    torch.cuda.synchronize()
    if rep == 0:
        base = got
    # (a) handed-out detections against the single-lane run
    for s in range(3):
        for i in range(16):
            if not (base[s][2][i].shape == got[s][2][i].shape and torch.equal(base[s][2][i], got[s][2][i])):
                bad_total += 1
                print("rep", rep, (lanes, threaded), "step", s, "sweep", i, "handed-out labels differ:", len(base[s][2][i]), len(got[s][2][i]))
    # (b) stashed maps against the quiet forward; stashed rows against a quiet decode of the stashed maps
    if not threaded:
        order = []
        per = 16 // lanes
        for s in range(3):
            for l in range(lanes):
                order.append((s, l * per, (l + 1) * per))
        for (n, heads, rows, total, stream), (s, lo, hi) in zip(stash, order):
            d = float((heads - quiet[s][lo:hi]).abs().max())
            nchw = heads.permute(0, 3, 1, 2)
            views, off = [], 0
            for c in (heads.shape[3] * 0 + x for x in m.packed_params(m.device)['head_split']):
                views.append(nchw[:, off:off + c]); off += c
            r2, t2 = m.bbox_head.boxes_device(*views)
            torch.cuda.synchronize()
            same_t = torch.equal(total, t2)
            same_r = all(torch.equal(rows[b, :int(total[b])], r2[b, :int(t2[b])]) for b in range(n)) if same_t else False
            if d > 0 or not same_t or not same_r:
                print("rep", rep, (lanes, threaded), "step", s, "sweeps", lo, hi, "maps max|d| vs quiet:", d, "| decode in pipeline == quiet decode:", same_t, same_r,
                      total.tolist(), t2.tolist())
print("mismatching handed-out sweeps:", bad_total)
