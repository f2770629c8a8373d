"""GPU box: `steps` pipelined KPConv steps (the bench's 64-sphere batch) and nothing else -- the process rocprofv3 traces for
tools/trace_timeline.py.  ML3D_KP_BUILDERS = builds in flight.  usage: python tools/kp_steps.py [steps]"""
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
from ml3d.engine import KPConvPipeline, KPConvPipelineN
from ml3d.torch.models.kpconv import KPFCNN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
cfg = dict(W.TORONTO3D_CFG)
m = KPFCNN(**cfg, device=dev)
m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
spheres = [synth_data.toronto3d_sphere(i) for i in range(64)]
lens = [len(s) for s in spheres]
host = torch.from_numpy(np.concatenate(spheres)).pin_memory()
np.random.seed(0)
b = max(1, int(os.environ.get("ML3D_KP_BUILDERS", "2")))
pipe = KPConvPipelineN(m, cfg, dev, builders=b) if b > 1 else KPConvPipeline(m, cfg, dev)
for _ in range(8):
    pipe.submit(host.to(dev, non_blocking=True), lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.submit(host.to(dev, non_blocking=True), lens)
pipe.flush()
torch.cuda.synchronize()
print("builders %d: %.3f ms per step" % (b, (time.perf_counter() - t0) / steps * 1e3))
