"""ASan workload, round 5: the k-NN scan that reads past a run's end (GRID_SORTED_SLACK), the one-call KPConv batch build, the
degenerate NMS boxes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import emu, emu_runtime, synth_data, synth_weights as W
from ml3d import _abi
prod = emu_runtime.install("ml3d")
if os.environ.get("ML3D_EMU_LIB"):
    emu._LIB = _abi._lib
for n in (1, 2, 3, 5, 64, 999, 3000):
    pts = synth_data.uniform_cloud(n, n)
    idx, _ = emu.knn(pts, [0, n], k=16)
print("knn sizes ok")
p = synth_data.uniform_cloud(5, 2 * 1024).reshape(2, 1024, 3)
emu.pyramid_ordered(p, [4, 4, 4, 4]); print("pyramid ok")
from ml3d.torch.models.kpconv import KPConvBatch
cfg = dict(W.TORONTO3D_CFG)
spheres = [synth_data.toronto3d_sphere(3, 900, radius=1.5), np.zeros((0, 3), np.float32), synth_data.toronto3d_sphere(4, 700, radius=1.5), np.array([[9., 9., 9.]], np.float32)]
np.random.seed(1)
b = KPConvBatch(np.concatenate(spheres).astype(np.float32), [len(s) for s in spheres], cfg, device="cpu")
print("kpbatch ok", getattr(b, "host_syncs", None), [int(x.shape[0]) for x in b.points])
rng = np.random.default_rng(12)
bx = rng.random((64, 5), dtype=np.float32) * 5; bx[:, 2:4] += bx[:, :2]
bx[3, 0] = np.inf; bx[7] = [np.nan, 0, np.inf, 1, .3]; bx[9, 4] = np.nan
emu.nms(bx, rng.random(64, dtype=np.float32), 0.3); print("nms ok")
print("done: no AddressSanitizer report above means clean")
