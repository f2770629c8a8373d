"""GPU box: run ONLY the op that `workloads.*.roofline` of the bench line prices, a few times, so that a rocprofv3 --pmc pass
attributes HBM traffic to it alone.  usage: python tools/roofline_ops.py kp|pp [reps]
  kp  the first resnet block's KPConv of the 64-sphere Toronto3D batch (32 -> 32 channels, 640 000 queries): kp_agg_gemm32
      (aggregation + product in one kernel), the op bench_models.run_kpconv times as `kpconv_rigid` call #1
  pp  SECOND's second convolution of 16 KITTI sweeps (3x3, 64 -> 64, stride 1, 248 x 216): bench_models.run_pointpillars's
      `conv2d_nhwc` call #1, on the bf16x3 path the model runs (pp_f32: the f32 MFMA kernel on the same problem)
  radius | subsample | voxelize | pillars   the four HBM-bound primitives of SURVEY.md §8(d) at the batch shapes of the bench
      (64 Toronto3D spheres: layer-0 conv search r = 0.2 m / layer-0 pooling grid 0.16 m; 8 KITTI sweeps = one lane's launch:
      voxelize / pillar gather + PFN + canvas scatter), each ALONE in the process, `launches` times after its inputs are on
      the device -- the `roofline_other` entries of `bench.py --workload kpconv | pointpillars`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W
from ml3d import ops

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
if which == "kp":
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    cfg = dict(W.TORONTO3D_CFG)
    m = KPFCNN(**cfg, device=dev)
    m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
    spheres = [synth_data.toronto3d_sphere(i) for i in range(64)]
    np.random.seed(0)
    batch = KPConvBatch(np.concatenate(spheres), [len(s) for s in spheres], cfg, device=dev)
    P = m.packed_params(dev)
    c = P['enc'][1]['conv']
    x = torch.randn((batch.points[0].shape[0], 32), device=dev)
    run = lambda: ops.kpconv_rigid(batch.points[0], batch.points[0], batch.neighbors[0], x, c['kp'], c['w'], c['b'], c['extent'],
                                   1, 0.2, 1)
    units = 64
elif which in ("radius", "subsample"):
    spheres = [synth_data.toronto3d_sphere(i) for i in range(64)]
    lens = [len(s) for s in spheres]
    pts = torch.from_numpy(np.concatenate(spheres)).to(dev)
    units = 64
    if which == "radius":
        run = lambda: ops.radius_neighbors_dense(pts, pts, lens, lens, 0.2)
    else:
        from ml3d.torch.models.kpconv import random_grid_rotations
        np.random.seed(0)
        R = torch.from_numpy(random_grid_rotations(64)).to(dev)
        run = lambda: ops.batch_grid_subsampling(pts, lens, 0.16, R)
elif which in ("voxelize", "pillars"):
    from ml3d.torch.models.point_pillars import PointPillars
    cfg = W.POINTPILLARS_KITTI_CFG
    m = PointPillars(device=dev, **cfg)
    m.load_state_dict(W.pointpillars_state_dict(cfg, 2024))
    clouds = [torch.from_numpy(W.crop_for_cfg(synth_data.kitti_sweep(i), cfg)).to(dev) for i in range(8)]
    units = 8
    name = "voxelize" if which == "voxelize" else "pillar_features"
    calls, origs = {}, {}

    class _Captured(Exception):
        pass
    for nm in ("voxelize", "pillar_features"):          # the model's own call, captured with its arguments; the forward stops there
        origs[nm] = getattr(ops, nm)                     # (nothing downstream of the op runs in this process: its PMC pass sees the op alone)
        def grab(*a, _n=nm, **k):
            if _n == name:
                calls[_n] = (a, k)
                raise _Captured()
            return origs[_n](*a, **k)
        setattr(ops, nm, grab)
    try:
        m(clouds)
    except _Captured:
        pass
    torch.cuda.synchronize()
    a, k = calls[name]
    run = lambda: origs[name](*a, **k)
else:
    x = torch.randn((16, 248, 216, 64), device=dev)
    w = torch.randn((9 * 64, 64), device=dev) * 0.05
    b = torch.randn(64, device=dev)
    # pp: the path the model runs (bf16x3 weights, gemm_tile_bf3); pp_f32: the f32 MFMA kernel (gemm_tile2) on the same problem
    pk = ops.pack_bf16x3(w) if which != "pp_f32" else None
    run = lambda: ops.conv2d_nhwc(x, w, b, 3, 3, 1, 1, act=2, packed=pk)
    units = 16
run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    ev[0].record()
    run()
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("%s: %d units per launch, median %.4f ms over %d launches" % (which, units, float(np.median(ts)), reps))
# (for tools/make_traffic.py --op: the op ran reps + 1 times in this process)
print("LAUNCHES %d" % (reps + 1))
