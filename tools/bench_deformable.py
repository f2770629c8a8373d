"""GPU box: KPFCNN with the deformable blocks of kpconv_parislille3d.yml (synth_weights.PARISLILLE3D_CFG) -- batch build +
forward of a batch of synthetic input spheres, native (HIP ops) against the CPU oracle on one sphere.  One JSON line.
usage: python tools/bench_deformable.py [spheres_per_batch] [steps]"""
import json
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
from ml3d.engine import KPConvPipeline
from ml3d.torch.models.kpconv import KPFCNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = dict(W.PARISLILLE3D_CFG)
dev = torch.device("cuda:0")
sd = W.kpconv_state_dict(cfg, 21)
m = KPFCNN(**cfg, device=dev)
m.load_state_dict(sd)
m.eval()
spheres = [synth_data.toronto3d_sphere(100 + i) for i in range(B)]
lens = [len(s) for s in spheres]
host = torch.from_numpy(np.concatenate(spheres)).pin_memory()
np.random.seed(0)
pipe = KPConvPipeline(m, cfg, dev)
for _ in range(4):
    pipe.submit(host.to(dev, non_blocking=True), lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.submit(host.to(dev, non_blocking=True), lens)
res = pipe.flush()
res.wait()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"workload": "KPFCNN kpconv_parislille3d.yml architecture (5 deformable blocks), %d synthetic 10000-point spheres per step: "
                   "H2D + GPU batch build + forward" % B, "spheres_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3}
from oracle import kpconv_ref as K
sp = spheres[0]
t0 = time.perf_counter()
seg = K.segmentation_inputs(sp, [len(sp)], cfg)
ref = K.forward(sd, cfg, K.to_torch_batch(seg), torch.ones((len(sp), 1)))
out["cpu_oracle_spheres_per_s"] = 1.0 / (time.perf_counter() - t0)
out["cpu_threads"] = int(torch.get_num_threads())
print(json.dumps(out))
