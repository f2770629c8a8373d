"""Quick timing of the KPConv and PointPillars inference paths on one MI355X (development aid).
usage: python tools/perf_models.py [kpconv|pointpillars|all] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def kpconv(iters, nspheres=4):
    import synth_weights as W
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    cfg = dict(W.TORONTO3D_CFG)
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(W.kpconv_state_dict(cfg, 1))
    spheres = [synth_data.toronto3d_sphere(100 + i) for i in range(nspheres)]
    pts = torch.from_numpy(np.concatenate(spheres)).cuda()
    lens = [len(s) for s in spheres]
    np.random.seed(0)
    batch = KPConvBatch(pts, lens, cfg, device="cuda:0")
    t_b = timeit(lambda: KPConvBatch(pts, lens, cfg, device="cuda:0"), iters)
    t_f = timeit(lambda: m(batch), iters)
    print("kpconv: %d spheres (%d pts): batcher %.3f ms, forward %.3f ms -> %.1f spheres/s" %
          (nspheres, sum(lens), t_b, t_f, nspheres / (t_b + t_f) * 1e3))


def pointpillars(iters, nframes=2):
    import synth_weights as W
    from ml3d.torch.models.point_pillars import PointPillars
    cfg = W.POINTPILLARS_KITTI_CFG
    m = PointPillars(device="cuda:0", **cfg)
    m.load_state_dict(W.pointpillars_state_dict(cfg, 1))
    clouds = [torch.from_numpy(W.crop_for_cfg(synth_data.kitti_sweep(i), cfg)).cuda() for i in range(nframes)]
    t = timeit(lambda: m(clouds), iters)
    print("pointpillars: %d frames (%s pts): forward %.3f ms -> %.1f frames/s" %
          (nframes, [len(c) for c in clouds], t, nframes / t * 1e3))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if what in ("kpconv", "all"):
        kpconv(iters)
    if what in ("pointpillars", "all"):
        pointpillars(iters)
