"""GPU box: the REAL graphed patch loop, piecewise.  usage: python tools/r05_calls/debug_graphs2.py patch|both [patches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights
import bench
from ml3d.torch.dataloaders import DefaultBatcher
from ml3d.torch.models import RandLANet

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
cfg = dict(bench.CFG, grid_size=0.06, augment={"recenter": {"dim": [0, 1]}})
m = RandLANet(**cfg, device=dev, seed=5)
m.load_state_dict(synth_weights.randlanet_state_dict(bench.CFG, 2024))
sweep = synth_data.lidar_sweep(5000)
m.inference_begin(dict(point=sweep, feat=None, label=np.zeros(sweep.shape[0], np.int32)))
collate = DefaultBatcher().collate_fn
attr = {"split": "test"}
if mode == "patch":
    m._forward_graphed = lambda inputs: None
for i in range(n):
    d = m.transform(m.inference_data, attr)
    torch.cuda.synchronize(); print("patch", i, "transform ok", m._dev_loop.get("graph_failed"), flush=True)
    inputs = collate([{"data": d, "attr": attr}])
    sc = m(inputs["data"])
    torch.cuda.synchronize(); print("patch", i, "forward ok", m._dev_loop.get("fwd_failed"), float(sc.abs().max()), flush=True)
    m.update_probs(inputs, sc, m.test_probs)
    torch.cuda.synchronize(); print("patch", i, "votes ok", flush=True)
print("done", mode)
