"""GPU box experiment: bench.py's RandLA step dealt to `lanes` independent RandLAFrameStream pipelines (each: upload / search /
forward streams + its own argmax on a post stream), `frames` frames per step in total.  usage: python tools/randla_lanes.py lanes frames [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import bench
import synth_weights
from ml3d import dist as mdist
from ml3d.engine import RandLAFrameStream

lanes, B = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
CFG = bench.CFG
N = CFG["num_points"]
dev = torch.device("cuda:0")
Bl = B // lanes
frames = bench.synthetic_batch(0, B, N, 8)
sd = synth_weights.randlanet_state_dict(CFG, 2024)
streams = [RandLAFrameStream(CFG, sd, Bl, N, dev, overlap=True) for _ in range(lanes)]
hosts = [torch.from_numpy(frames).pin_memory(), torch.from_numpy(np.ascontiguousarray(frames[::-1])).pin_memory()]
gathers = [mdist.PredictionGather(Bl, N, CFG["num_classes"], dev) for _ in range(lanes)]
posts = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
labelled = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(lanes)]
step_no = 0


def one_step():
    global step_no
    slot = step_no & 1
    for li, st in enumerate(streams):
        if step_no >= 2:
            st.compute_stream.wait_event(labelled[li][slot])
        scores = st.submit(hosts[slot][li * Bl:(li + 1) * Bl], None, None, None, None)
        with torch.cuda.stream(posts[li]):
            posts[li].wait_stream(st.compute_stream)
            gathers[li].push(scores)
            labelled[li][slot].record(posts[li])
    step_no += 1


for _ in range(5):
    one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    one_step()
for st in streams:
    st.synchronize()
for p in posts:
    p.synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("lanes %d x %d frames: %.1f frames/s, %.3f ms per step" % (lanes, Bl, B * K / dt, dt / K * 1e3))
