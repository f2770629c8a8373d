"""GPU box helper: run only the RandLA neighbour pyramid of a 64-frame batch a few times (for rocprofv3 counter passes
and A/B timing of the k-NN kernels without the forward).  usage: python tools/knn_only.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights
from ml3d.engine import RandLAInferenceEngine, make_trace

CFG = dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG)
B, N = int(os.environ.get("ML3D_BENCH_BATCH", 64)), CFG["num_points"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
base = np.stack([synth_data.semantickitti_patch(i, N) for i in range(8)])
rng = np.random.default_rng(77)
frames = np.empty((B, N, 3), np.float32)
for b in range(B):
    f = base[b % 8]
    if b >= 8:
        a = rng.uniform(0, 2 * np.pi)
        rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        f = (f @ rot.T)[rng.permutation(N)]
    frames[b] = f
dev = torch.device("cuda:0")
eng = RandLAInferenceEngine(CFG, synth_weights.randlanet_state_dict(CFG, 2024), B, N, dev)
pts = torch.from_numpy(frames).to(dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); b.record()
out = {}
for tag in [0, 100, 101, 102, 103]:
    ts = []
    for _ in range(reps):
        eng.neighbors(pts, make_trace(tag, a, b))
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out[tag] = float(np.median(ts))
print("knn_only", " ".join("%d=%.3f" % kv for kv in out.items()))
