"""Workload for tools/emu_asan.sh: k-NN (batched, ragged), the RandLA forward on every attention width incl. tiles that
straddle clouds, with and without a tile order."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import emu
import synth_data
from ml3d import _abi
from oracle import randlanet_ref as R

if os.environ.get("ML3D_EMU_LIB"):
    emu._LIB = _abi.bind(C.CDLL(os.environ["ML3D_EMU_LIB"]))
pts = synth_data.uniform_cloud(1, 3000)
idx, _ = emu.knn(pts, [0, 1000, 1000, 3000], k=16)
print("knn ok", idx.shape)
CFGS = [(dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
              dim_features=8, dim_output=[16, 64, 128, 256]), 2, 1024),
        (dict(num_neighbors=16, num_layers=3, num_classes=7, sub_sampling_ratio=[4, 4, 4], in_channels=3,
              dim_features=8, dim_output=[16, 64, 128]), 3, 1100)]
for cfg, B, N in CFGS:
    p = synth_data.uniform_cloud(5, B * N).reshape(B, N, 3)
    sd = R.make_state_dict(cfg, 3)
    nbr, itp, order = emu.pyramid_ordered(p, cfg["sub_sampling_ratio"])
    rc, a = emu.randla_forward(cfg, sd, p, p.copy(), nbr, itp)
    rc2, b = emu.randla_forward(cfg, sd, p, p.copy(), nbr, itp, order=order)
    assert rc == 0 and rc2 == 0 and np.array_equal(a, b)
    print("forward ok", cfg["dim_output"], B, N)
print("done: no AddressSanitizer report above means clean")
