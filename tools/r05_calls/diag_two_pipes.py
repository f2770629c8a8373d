"""f32 MFMA against bf16x3 convolutions on the same sweeps: map differences and how the detections differ (counts, unmatched boxes and why)."""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
import torch
import synth_data
from oracle import pointpillars_ref as P
from ml3d.torch.models import PointPillars
cfg = P.KITTI_CFG
sd = P.make_state_dict(cfg, 2024)
clouds = [torch.from_numpy(P.crop_for_cfg(synth_data.kitti_sweep(i), cfg)).cuda() for i in (0, 5, 9, 11, 12, 13)]
res = {}
for path in ("f32", "bf16x3"):
    os.environ["ML3D_PP_CONV"] = path
    m = PointPillars(device="cuda:0", **cfg); m.load_state_dict(sd); m.eval()
    outs = m(clouds)
    res[path] = (outs, m.bbox_head.get_bboxes(*outs))
print("maps max|d|:", [float((a - b).abs().max()) for a, b in zip(res["f32"][0], res["bf16x3"][0])])
for i in range(len(clouds)):
    bf, sf, lf = (res["f32"][1][k][i].cpu() for k in range(3))
    bb, sb, lb = (res["bf16x3"][1][k][i].cpu() for k in range(3))
    un = []
    for j in range(len(lf)):
        same = (lb == lf[j]).nonzero().flatten()
        d = (bb[same] - bf[j]).abs()
        ok = (d.amax(1) <= 1e-3) & ((sb[same] - sf[j]).abs() <= 1e-4)
        if not bool(ok.any()):
            near = d[:, :2].amax(1).argmin() if len(same) else None
            un.append((j, int(lf[j]), float(sf[j]), None if near is None else [round(float(v), 5) for v in d[near]], None if near is None else float((sb[same][near] - sf[j]).abs())))
    print("sweep", i, "f32", len(lf), "bf16x3", len(lb), "identical labels", bool(len(lf) == len(lb) and torch.equal(lf, lb)), "unmatched", len(un), un[:4])
