"""GPU box: the patch loop with only ONE part of the per-patch sequence captured, the rest eager, data changing between replays.
usage: python tools/r05_calls/debug_graphs3.py crop|pyramid|labels|argmin [patches]"""
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
from ml3d import ops
from ml3d.torch.models import RandLANet

part = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
k, L = int(cfg["num_points"]), cfg["num_layers"]
perm = torch.empty(k, dtype=torch.int32, device=dev)
center = torch.zeros(1, dtype=torch.int64, device=dev)


def f_argmin():
    center.copy_(torch.argmin(st['possibility']).reshape(1))


def f_crop():
    ops.device_patch(st['points'], st['possibility'], center, perm, k, st['dims'], st['feat'], st['bias'], st['scale'],
                     out=(v['pts'], v['feats'], v['sel']))


import ctypes as C
from ml3d import _abi
_r = (C.c_int32 * L)(*[int(x) for x in cfg["sub_sampling_ratio"]])
PWS = torch.empty(int(_abi.get().ml3d_randla_pyramid_workspace_bytes(1, k, L, _r)), dtype=torch.uint8, device=dev) \
    if os.environ.get("STATIC_WS", "1") == "1" else None


def f_pyramid():
    ops.randla_knn_pyramid(v['pts'][None], cfg["sub_sampling_ratio"], cfg["num_neighbors"],
                           out=([v['nbr%d' % l] for l in range(L)], [v['itp%d' % l] for l in range(L)]), workspace=PWS)


def f_labels():
    torch.index_select(st['label'], 0, v['sel'].long(), out=v['labels'])


parts = [("argmin", f_argmin), ("crop", f_crop), ("pyramid", f_pyramid), ("labels", f_labels)]
graphs = {}
for i in range(n):
    perm.copy_(torch.from_numpy(m.rng.permutation(k).astype(np.int32)))
    for name, fn in parts:
        if name == part:
            if name not in graphs:
                keep = st['possibility'].clone()
                fn(); torch.cuda.synchronize()
                st['possibility'].copy_(keep)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                graphs[name] = g
            graphs[name].replay()
        else:
            fn()
        torch.cuda.synchronize()
        print("patch", i, name, "ok (graph)" if name == part else "ok", flush=True)
    print("patch", i, "centre", int(center), "nbr checksum", int(v['nbr0'].long().sum()), flush=True)
print("done", part)
