"""GPU box helper: only the model-class API latency loop of bench.py (for rocprofv3 kernel tables of the patch loop)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import torch

import bench
import synth_weights

dev = torch.device("cuda:0")
sd = synth_weights.randlanet_state_dict(bench.CFG, 2024)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
print(json.dumps(bench.latency(dev, sd, frames_timed=n, frames_warm=10)))
