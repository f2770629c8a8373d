"""GPU box: the KPFCNN forward ALONE on one prebuilt 64-sphere batch, `reps` times (ms per forward printed), for a rocprofv3
--kernel-trace whose per-kernel SEQUENCE of one forward tools/trace_sequence.py prints.  usage: python tools/kp_forward_seq.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W
from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = dict(W.TORONTO3D_CFG)
m = KPFCNN(**cfg, device=dev)
m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
spheres = [synth_data.toronto3d_sphere(i) for i in range(64)]
np.random.seed(0)
batch = KPConvBatch(torch.from_numpy(np.concatenate(spheres)).to(dev), [len(s) for s in spheres], cfg, device=dev)
print("levels", [int(p.shape[0]) for p in batch.points], "cols", [tuple(t.shape) for t in batch.neighbors])
for _ in range(3):
    m(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m(batch)
torch.cuda.synchronize()
print("forward alone: %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
t0 = time.perf_counter()
for _ in range(reps):
    np.random.seed(0)
    KPConvBatch(torch.from_numpy(np.concatenate(spheres[:1] * 0 + spheres)).to(dev) if False else batch.points[0], [len(s) for s in spheres], cfg, device=dev)
torch.cuda.synchronize()
print("build alone: %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
