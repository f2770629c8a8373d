"""GPU box: run ONLY the op that `workloads.*.roofline` of the bench line prices, a few times, so that a rocprofv3 --pmc pass
attributes HBM traffic to it alone.  usage: python tools/roofline_ops.py kp|pp [reps]
  kp  the first resnet block's KPConv of the 64-sphere Toronto3D batch (32 -> 32 channels, 640 000 queries): kp_agg_gemm32
      (aggregation + product in one kernel), the op bench_models.run_kpconv times as `kpconv_rigid` call #1
  pp  SECOND's second convolution of 16 KITTI sweeps (3x3, 64 -> 64, stride 1, 248 x 216): bench_models.run_pointpillars's
      `conv2d_nhwc` call #1"""
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
else:
    x = torch.randn((16, 248, 216, 64), device=dev)
    w = torch.randn((9 * 64, 64), device=dev) * 0.05
    b = torch.randn(64, device=dev)
    run = lambda: ops.conv2d_nhwc(x, w, b, 3, 3, 1, 1, act=2)
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
