"""GPU box: where the HOST time of a pipelined KPConv step goes (6.0 ms of kernels in an 8.35 ms step).
Runs the bench's own pipeline (KPConvPipeline: build of step i+1 under the forward of step i) for `steps` steps and prints
(1) wall per step, (2) per step: seconds inside blocking read-backs (Tensor.tolist / .item), inside library calls (ctypes),
inside torch.empty / tensor constructors, number of each, (3) a cProfile of the same loop, top 30 by cumulative time.
usage: python tools/kp_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W
from ml3d import _abi
from ml3d.engine import KPConvPipeline
from ml3d.torch.models.kpconv import KPFCNN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
cfg = dict(W.TORONTO3D_CFG)
m = KPFCNN(**cfg, device=dev)
m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
spheres = [synth_data.toronto3d_sphere(i) for i in range(64)]
lens = [len(s) for s in spheres]
host = torch.from_numpy(np.concatenate(spheres)).pin_memory()
np.random.seed(0)
pipe = KPConvPipeline(m, cfg, dev, threaded=bool(int(os.environ.get("ML3D_KP_THREADED", "0"))))


def step():
    pipe.submit(host.to(dev, non_blocking=True), lens)


for _ in range(8):
    step()
torch.cuda.synchronize()

acc = {"sync_s": 0.0, "sync_n": 0, "lib_s": 0.0, "lib_n": 0}
orig_tolist = torch.Tensor.tolist


def tolist(self):
    t0 = time.perf_counter()
    r = orig_tolist(self)
    if self.is_cuda:
        acc["sync_s"] += time.perf_counter() - t0
        acc["sync_n"] += 1
    return r


torch.Tensor.tolist = tolist
lib = _abi.get()
wrapped = {}
for sym in _abi.SYMBOLS:
    fn = getattr(lib, sym)

    def call(*a, _f=fn):
        t0 = time.perf_counter()
        r = _f(*a)
        acc["lib_s"] += time.perf_counter() - t0
        acc["lib_n"] += 1
        return r
    wrapped[sym] = fn
    try:
        setattr(lib, sym, call)
    except Exception:
        pass
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
print("wall per step %.3f ms | blocking read-backs %.3f ms in %.1f calls | library calls %.3f ms in %.1f calls | rest (python, torch "
      "allocations, stream bookkeeping) %.3f ms" % (wall * 1e3, acc["sync_s"] / steps * 1e3, acc["sync_n"] / steps,
                                                     acc["lib_s"] / steps * 1e3, acc["lib_n"] / steps,
                                                     (wall - (acc["sync_s"] + acc["lib_s"]) / steps) * 1e3))
for sym, fn in wrapped.items():
    try:
        setattr(lib, sym, fn)
    except Exception:
        pass
torch.Tensor.tolist = orig_tolist
prof = cProfile.Profile()
prof.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
prof.disable()
pstats.Stats(prof, stream=sys.stdout).sort_stats("tottime").print_stats(28)
