"""ASan workload: the bf16x3 kernels (window-staged 3 x 3 convolution, general implicit-GEMM convolution, dense rows / transposed
convolution / split-K) on exact-size numpy buffers: an out-of-bounds store would corrupt a neighbouring allocation on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import emu, emu_runtime
from ml3d import _abi
prod = emu_runtime.install("ml3d")
if os.environ.get("ML3D_EMU_LIB"):
    emu._LIB = _abi._lib
rng = np.random.default_rng(1)
for cin, cout, hw, nb, stride in [(32, 64, (13, 11), 2, 1), (64, 128, (3, 140), 1, 1), (32, 200, (9, 16), 3, 1), (64, 40, (1, 37), 2, 1),
                                  (32, 64, (45, 1), 2, 1), (64, 72, (31, 27), 3, 1), (64, 128, (12, 14), 2, 2), (96, 132, (31, 9), 2, 2)]:
    x = rng.standard_normal((nb,) + hw + (cin,)).astype(np.float32)
    wk = (rng.standard_normal((9 * cin, cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    rc, out = emu.conv2d_nhwc_bf16x3(x, wk, b, stride, 1)
    assert rc == 0
    print("conv", cin, cout, hw, nb, stride, "ok", flush=True)
for stride in (1, 2, 4):
    x = rng.standard_normal((2, 6, 5, 64)).astype(np.float32)
    wk = (rng.standard_normal((64, stride * stride * 32)) * 0.1).astype(np.float32)
    big = np.zeros((2, 6 * stride, 5 * stride, 80), np.float32)
    rc, out = emu.deconv2d_nhwc_bf16x3(x, wk, rng.standard_normal(32).astype(np.float32), stride, 32, out=big, ch_off=48)
    assert rc == 0
    print("deconv", stride, "ok", flush=True)
for m, k1, k2, n, res in [(300, 384, 0, 72, False), (129, 32, 64, 128, True), (40, 512, 512, 64, False), (2000, 96, 0, 40, True)]:
    a = rng.standard_normal((m, k1)).astype(np.float32)
    a2 = rng.standard_normal((m, k2)).astype(np.float32) if k2 else None
    w = (rng.standard_normal((k1 + k2, n)) * 0.1).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32) if res else None
    rc, out = emu.linear_bf16x3(a, w, rng.standard_normal(n).astype(np.float32), act=1, a2=a2, residual=r)
    assert rc == 0
    print("linear", m, k1, k2, n, "ok", flush=True)
print("done: no AddressSanitizer report above means clean")
