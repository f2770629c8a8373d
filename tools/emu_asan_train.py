"""ASan workload: the training ops of csrc/train.hip (gemm_tn, BatchNorm forward / backward, gathers / pools, the fused attention stage at every
width class with more tiles than workgroups, the deformed KPConv aggregation) through their autograd Functions on exact-size tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import emu_runtime
emu_runtime.install("ml3d")
from ml3d import ops
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))
for m, k, n in ((1000, 70, 33), (5, 16, 16), (513, 130, 200)):
    ops.gemm_tn(T(rng.standard_normal((m, k))), T(rng.standard_normal((m, n))), with_col_sums=True)
print("gemm_tn ok", flush=True)
for shape, cout, bias in (((3, 50, 16, 10), 8, True), ((700, 64), 128, False)):
    x = T(rng.standard_normal(shape)).requires_grad_(True)
    w = T(rng.standard_normal((cout, shape[-1]))).requires_grad_(True)
    b = T(rng.standard_normal(cout)).requires_grad_(True) if bias else None
    ops.LinearFunction.apply(x, w, b).sum().backward()
for shape, slope in (((4, 30, 16, 8), 0.2), ((500, 300), None), ((10, 5), 0.0)):
    c = shape[-1]
    x = T(rng.standard_normal(shape)).requires_grad_(True)
    gam, bet = T(rng.random(c) + 0.5).requires_grad_(True), T(rng.standard_normal(c)).requires_grad_(True)
    ops.BatchNormActFunction.apply(x, gam, bet, T(np.zeros(c)), T(np.ones(c)), 0.01, 1e-6, slope).square().sum().backward()
print("linear / batchnorm ok", flush=True)
x = T(rng.standard_normal((40, 12))).requires_grad_(True)
ops.GatherRowsFunction.apply(x, torch.from_numpy(rng.integers(0, 41, 100).astype(np.int32))).sum().backward()
for mode in ("max", "closest"):
    ops.GatherPoolFunction.apply(x, torch.from_numpy(rng.integers(0, 41, (30, 7)).astype(np.int32)), mode).sum().backward()
print("gathers ok", flush=True)
for B, n, c1, c2 in ((2, 37, 8, 8), (1, 2100, 8, 8), (1, 1100, 32, 32), (2, 530, 40, 56), (1, 70, 128, 128), (1, 21, 6, 10), (1, 9, 150, 106)):
    f = T(rng.standard_normal((B, n, c1))).requires_grad_(True)
    enc = T(rng.standard_normal((B, n, 16, c2))).requires_grad_(True)
    idx = torch.from_numpy(rng.integers(0, n, (B, n, 16)).astype(np.int32))
    d = c1 + c2
    w, b = T(rng.standard_normal((d, d)) * 0.3).requires_grad_(True), T(rng.standard_normal(d)).requires_grad_(True)
    ops.AttentionStageFunction.apply(f, enc, idx, w, b).square().sum().backward()
    print("attention stage ok", B, n, c1, c2, flush=True)
for nq, ns, H, cin in ((40, 60, 9, 8), (25, 25, 14, 70), (7, 30, 5, 130), (3, 0, 0, 4)):
    q, s_ = T(rng.random((nq, 3))), T(rng.random((ns, 3)))
    inds = torch.from_numpy(rng.integers(0, ns + 3, (nq, H)).astype(np.int32))
    x = T(rng.standard_normal((ns, cin))).requires_grad_(True)
    dkp = T(rng.standard_normal((nq, 15, 3)) * 0.2).requires_grad_(True)
    ops.KPConvDeformedFunction.apply(x, dkp, q, s_, inds, 0.35).square().sum().backward()
print("deformed aggregation ok", flush=True)
print("done: no AddressSanitizer report above means clean")
