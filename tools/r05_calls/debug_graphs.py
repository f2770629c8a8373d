"""GPU box: which captured piece of the patch loop misbehaves.  usage: python tools/r05_calls/debug_graphs.py <piece>
pieces: argmin | patch | pyramid | labels | forward  (each captures ONLY that piece into a graph, replays it 3 times, checks vs eager)"""
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
from ml3d import ops, _abi
from ml3d.torch.models import RandLANet

piece = sys.argv[1]
dev = torch.device("cuda:0")
cfg = dict(bench.CFG, grid_size=0.06, augment={"recenter": {"dim": [0, 1]}})
m = RandLANet(**cfg, device=dev, seed=5)
m.load_state_dict(synth_weights.randlanet_state_dict(bench.CFG, 2024))
m.use_graphs = False
sweep = synth_data.lidar_sweep(5000)
m.inference_begin(dict(point=sweep, feat=None, label=np.zeros(sweep.shape[0], np.int32)))
st = m._dev_loop
st['layout'] = m._arena_layout()
lay, nbytes = st['layout']
arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
v = m._arena_views(arena, lay)
k = int(cfg["num_points"])
perm = torch.arange(k, dtype=torch.int32, device=dev)
m._patch_into(v, perm)          # eager once: valid contents everywhere
torch.cuda.synchronize()
L = cfg["num_layers"]
center = torch.argmin(st['possibility']).reshape(1)


def f_argmin():
    return torch.argmin(st['possibility']).reshape(1)


def f_patch():
    ops.device_patch(st['points'], st['possibility'], center, perm, k, st['dims'], st['feat'], st['bias'], st['scale'],
                     out=(v['pts'], v['feats'], v['sel']))


def f_pyramid():
    ops.randla_knn_pyramid(v['pts'][None], cfg["sub_sampling_ratio"], cfg["num_neighbors"],
                           out=([v['nbr%d' % l] for l in range(L)], [v['itp%d' % l] for l in range(L)]))


def f_labels():
    torch.index_select(st['label'], 0, v['sel'].long(), out=v['labels'])


desc = _abi.make_desc(m.cfg, 1, k)
scores = torch.empty((1, k, cfg["num_classes"]), dtype=torch.float32, device=dev)
params = m.packed_params(dev)


def f_forward():
    ops.randla_forward(desc, params, v['feats'][None], v['pts'][None], [v['nbr%d' % l] for l in range(L)],
                       [v['itp%d' % l] for l in range(L)], out=scores)


fn = dict(argmin=f_argmin, patch=f_patch, pyramid=f_pyramid, labels=f_labels, forward=f_forward)[piece]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
before = arena.clone(); sc0 = scores.clone()
print(piece, "eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
torch.cuda.synchronize()
print(piece, "captured", flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print(piece, "replay", i, "ok", flush=True)
if piece in ("pyramid", "labels"):
    print("same as eager:", torch.equal(before, arena))
if piece == "forward":
    print("same as eager:", torch.equal(sc0, scores))
