"""Diagnose tests/test_gpu_pointpillars.py::test_two_lane_stream_returns_the_single_lane_detections at HEAD (run under
ML3D_PP_CONV=bf16x3 / f32, ML3D_CONV_WINDOW is a test-hook only): (1) are a sweep's head maps independent of the batch it rides
in, (2) which lane configuration / sweep differs and by how much."""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "open3d-ml_amd")]
import numpy as np
import torch
import synth_data
from oracle import pointpillars_ref as P
from ml3d.torch.models import PointPillars
from ml3d.engine import PointPillarsStream

cfg = P.KITTI_CFG
m = PointPillars(device="cuda:0", **cfg)
m.load_state_dict(P.make_state_dict(cfg, 2024))
m.eval()
clouds = [P.crop_for_cfg(synth_data.kitti_sweep(i), cfg) for i in range(16)]
dev = [torch.from_numpy(c).cuda() for c in clouds]
with torch.no_grad():
    full = [t.clone() for t in m(dev)]
    for rep in range(2):
        again = m(dev)
        print("rerun", rep, [float((a - b).abs().max()) for a, b in zip(full, again)])
    for lo, hi in ((0, 8), (8, 16), (0, 4), (5, 6), (3, 16)):
        part = m(dev[lo:hi])
        d = [(a[lo:hi] - b).abs().amax(dim=(1, 2, 3)).cpu().numpy() for a, b in zip(full, part)]
        print("batch [%d:%d) vs full, per-sweep max|d| cls/reg/dir:" % (lo, hi))
        for x in d:
            print("   ", np.array2string(x, precision=3))
    # where do they differ, if they do
    part = m(dev[8:16])
    dd = (full[0][8:16] - part[0]).abs()
    if float(dd.max()) > 0:
        nz = torch.nonzero(dd > 0.5 * dd.max())
        print("largest cls differences at (sweep, c/y, y/x, x/c):", nz[:12].cpu().tolist(), "of", int((dd > 0).sum()), "nonzero; shape", tuple(dd.shape))
steps = [[P.crop_for_cfg(synth_data.kitti_sweep(i), cfg) for i in range(16 * s, 16 * s + 16)] for s in range(3)]
hosts = [[torch.from_numpy(c).pin_memory() for c in st] for st in steps]
runs = []
for lanes, threaded in ((1, False), (2, False), (2, True), (1, False)):
    pipe = PointPillarsStream(m, "cuda", lanes=lanes, threaded=threaded)
    got = [pipe.submit(h) for h in hosts] + [pipe.flush()]
    runs.append(got[1:])
for r, other in enumerate(runs[1:]):
    bad = 0
    for s, (one, two) in enumerate(zip(runs[0], other)):
        for i in range(16):
            same = one[2][i].shape == two[2][i].shape and torch.equal(one[2][i], two[2][i])
            if not same:
                bad += 1
                print("run", r + 1, "step", s, "sweep", i, "labels differ: n =", len(one[2][i]), len(two[2][i]))
                if bad <= 3:
                    print("   scores one", one[1][i][:12].tolist()); print("   scores two", two[1][i][:12].tolist())
            else:
                ds = float((one[1][i] - two[1][i]).abs().max()) if len(one[1][i]) else 0.0
                db = float((one[0][i] - two[0][i]).abs().max()) if len(one[0][i]) else 0.0
                if ds > 1e-6 or db > 1e-5:
                    print("run", r + 1, "step", s, "sweep", i, "scores/boxes differ", ds, db)
    print("run", r + 1, "mismatching sweeps:", bad)
