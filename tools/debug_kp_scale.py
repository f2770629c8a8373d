"""GPU box: find the first op of the KPFCNN forward whose rows for batch item 0 differ between a 64-sphere batch and the same
sphere alone (same grid rotations).  usage: python tools/debug_kp_scale.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W
from ml3d import ops
from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(W.TORONTO3D_CFG)
dev = torch.device("cuda:0")
m = KPFCNN(**cfg, device=dev)
m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
spheres = [synth_data.toronto3d_sphere(i) for i in range(B)]
lens = [len(s) for s in spheres]
np.random.seed(0)
big = KPConvBatch(np.concatenate(spheres), lens, cfg, device=dev)
one = KPConvBatch(spheres[0], lens[:1], cfg, rotations=[None if R is None else R[:1] for R in big.rotations], device=dev)
n_item0 = [int(l[0]) for l in one.lengths]
print("layer sizes of item 0:", n_item0, " batch columns:", [tuple(t.shape) for t in big.neighbors],
      " single columns:", [tuple(t.shape) for t in one.neighbors])

log = {}


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        r = orig(*a, **k)
        log[tag[0]].append((name, r, [tuple(x.shape) if torch.is_tensor(x) else x for x in a[:6]]))
        return r
    return orig, f


tag = ["big"]
origs = {}
for nm in ("kpconv_rigid", "linear", "gather_pool"):
    origs[nm], f = wrap(nm)
    setattr(ops, nm, f)
log["big"], log["one"] = [], []
out_big = m(big)
tag[0] = "one"
out_one = m(one)
torch.cuda.synchronize()
for i, ((n1, r1, a1), (n2, r2, a2)) in enumerate(zip(log["big"], log["one"])):
    rows = r2.shape[0]
    d = (r1[:rows] - r2).abs().max().item()
    ref = r2.abs().max().item()
    print("%2d %-13s rows %6d / %7d cols %4d  max|d| %.3e  (max|x| %.3e)  args big %s | one %s" % (
        i, n1, rows, r1.shape[0], r2.shape[1], d, ref, a1, a2))
print("logits max|d| item 0:", (out_big[:lens[0]] - out_one).abs().max().item())
