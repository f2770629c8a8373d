"""CPU: the KPConv building blocks (kpconv.hip + gemm.hip) through the host emulator vs the oracle's
PyTorch restatement of the reference ops (oracle/kpconv_ref.py) — float tolerance 1e-4."""
import os

import numpy as np
import pytest
import torch

import emu
import synth_data
from oracle import kpconv_ref as K

pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")
TOL = 1e-4


def _layer(seed, n, r):
    p = synth_data.toronto3d_sphere(seed, n)
    inds = K.batch_neighbors(p, p, [len(p)], [len(p)], r)
    return p, inds


@pytest.mark.parametrize("cin,cout,n", [(1, 64, 700), (5, 16, 300), (2, 32, 333), (4, 128, 300), (5, 96, 301), (3, 160, 200),
                                        (32, 32, 600), (24, 40, 260), (64, 64, 300),
                                        (128, 128, 130), (256, 64, 70)])
def test_kpconv_rigid_matches_reference_ops(cin, cout, n):
    rng = np.random.default_rng(cin * 1000 + cout)
    p, inds = _layer(21, n, 0.2)
    x = rng.standard_normal((len(p), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, cin, cout)) * (1.0 / np.sqrt(cin * 4))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    rc, out = emu.kpconv_rigid(p, p, inds, x, kp, w, 0.08, bias=b, act=1, slope=0.2)
    assert rc == 0
    ref = K.kpconv_rigid(torch.from_numpy(p), torch.from_numpy(p), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08)
    ref = torch.nn.functional.leaky_relu(ref + torch.from_numpy(b), 0.2).numpy()
    assert np.abs(out - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_kpconv_strided_queries_and_shadow_only_rows():
    rng = np.random.default_rng(3)
    s = synth_data.toronto3d_sphere(22, 900)
    q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.16)[0], [[50, 50, 50]]]).astype(np.float32)  # last: no nbrs
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], 0.2)
    assert (inds[-1] == len(s)).all()
    x = rng.standard_normal((len(s), 32)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, 32, 48)) * 0.1).astype(np.float32)
    rc, out = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08)
    ref = K.kpconv_rigid(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08).numpy()
    assert rc == 0 and np.abs(out - ref).max() <= TOL and (out[-1] == 0).all()


@pytest.mark.parametrize("cin,n,r", [(16, 700, 0.3), (32, 1500, 0.3), (64, 800, 0.35), (128, 500, 0.3), (256, 300, 0.25)])
def test_kpconv_mfma_aggregation_rows_wider_than_64_shadows_anywhere(cin, n, r):
    """cin in {16, 32, 64, 128, 256} aggregates on the matrix unit (kp_agg_mfma: one wave per query, lane = (kernel point,
    neighbour of a group of four)).  Rows of 65 .. 128 columns (second index register), the columns of every row shuffled so that
    shadow entries sit ANYWHERE (the walk ends at the last real column, not at the first shadow), strided queries, a query with
    only shadow neighbours."""
    rng = np.random.default_rng(cin)
    s = synth_data.toronto3d_sphere(22, n)
    q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.1)[0], [[50, 50, 50]]]).astype(np.float32)
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], r)
    assert 40 < inds.shape[1] <= 128
    inds = np.take_along_axis(inds, np.argsort(rng.random(inds.shape), axis=1), 1)
    x = rng.standard_normal((len(s), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, cin, 32)) * (0.5 / np.sqrt(cin))).astype(np.float32)
    rc, out = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08)
    ref = K.kpconv_rigid(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08).numpy()
    assert rc == 0 and np.abs(out - ref).max() <= TOL * max(1.0, np.abs(ref).max()) and (out[-1] == 0).all()


@pytest.mark.parametrize("cin,n,r,modulated", [(32, 1500, 0.3, False), (64, 900, 0.55, False), (16, 700, 0.3, True),
                                                 (128, 500, 0.3, True), (512, 260, 0.3, False)])
def test_kpconv_deformable_matches_the_reference_formulation(cin, n, r, modulated):
    """``ml3d_kpconv_deformable`` (kpconv.py:1011-1159): the inner rigid convolution's output moves each query's kernel points,
    optional modulations scale the weighted features.  Against the oracle's restatement of the reference branch INCLUDING its
    in-range pruning and top-k reordering of the neighbour lists (which only drops zero-influence neighbours).  The 0.55 m case
    has rows wider than 128 columns (walked 128 at a time); strided queries, a query with only shadow neighbours."""
    rng = np.random.default_rng(cin + 7)
    s = synth_data.toronto3d_sphere(22, n)
    q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.1)[0], [[50, 50, 50]]]).astype(np.float32)
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], r)
    assert (inds.shape[1] > 128) == (r > 0.5)
    x = rng.standard_normal((len(s), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    od = 60 if modulated else 45
    w = (rng.standard_normal((15, cin, 32)) * (0.5 / np.sqrt(cin))).astype(np.float32)
    ow = (rng.standard_normal((15, cin, od)) * (0.12 / np.sqrt(cin))).astype(np.float32)
    ob = (rng.standard_normal(od) * 0.1).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    rc, off = emu.kpconv_rigid(q, s, inds, x, kp, ow, 0.08, bias=ob)                     # the inner convolution
    assert rc == 0
    rc, out = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08, bias=b, act=1, slope=0.1, offset_features=off)
    assert rc == 0
    t = torch.from_numpy
    ref = K.kpconv_deformable(t(q), t(s), t(inds).long(), t(x), t(kp), t(w), 0.08, t(ow), t(ob), modulated)
    ref = torch.nn.functional.leaky_relu(ref + t(b), 0.1).numpy()
    assert np.abs(out - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    # the offsets really move the result (a rigid run differs)
    rc, rigid = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08, bias=b, act=1, slope=0.1)
    assert np.abs(rigid - ref).max() > 100 * TOL
    # unsupported corners say so
    rc, _ = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08, influence=2, offset_features=off)
    assert rc == emu._abi.E_UNSUPPORTED if hasattr(emu._abi, "E_UNSUPPORTED") else rc != 0


_FUSED32_CASE = r"""
import os, sys
ROOT = %(root)r
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import emu, synth_data
from oracle import kpconv_ref as K
rng = np.random.default_rng(3)
s = synth_data.toronto3d_sphere(22, 4300)
q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.05)[0], [[50, 50, 50]]]).astype(np.float32)   # last: no neighbours
if len(q) %% 16 == 0:
    q = q[1:]
inds = K.batch_neighbors(q, s, [len(q)], [len(s)], 0.12)
assert (inds[-1] == len(s)).all() and len(q) > 16 * 256                 # more tiles than the 256 workgroups of the launch
x = rng.standard_normal((len(s), 32)).astype(np.float32)
kp = K.synthetic_kernel_points(0.2)
w = (rng.standard_normal((15, 32, 32)) * 0.1).astype(np.float32)
for infl, name in ((1, "linear"), (2, "gaussian")):
    rc, out = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08, act=2, influence=infl)
    tq, ts, ti, tx, tk, tw = (torch.from_numpy(a) for a in (q, s, inds.astype(np.int64), x, kp, w))
    if infl == 1:
        ref = K.kpconv_rigid(tq, ts, ti, tx, tk, tw, 0.08)
    else:                           # radius_gaussian (kpconv.py:1119-1125): exp(-d2 / (2 sigma^2 + eps)), sigma = 0.3 extent
        nb = torch.cat((ts, torch.zeros_like(ts[:1]) + 1e6), 0)[ti] - tq.unsqueeze(1)
        sq = ((nb.unsqueeze(2) - tk) ** 2).sum(3)
        wgt = torch.exp(-sq / (2 * (0.3 * 0.08) ** 2 + 1e-9)).transpose(1, 2)
        wf = torch.matmul(wgt, torch.cat((tx, torch.zeros_like(tx[:1])), 0)[ti]).permute(1, 0, 2)
        ref = torch.matmul(wf, tw).sum(0)
    ref = torch.relu(ref).numpy()
    assert rc == 0 and np.abs(out - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())) and (out[-1] == 0).all(), name
# rows wider than 128 columns (a dense cloud under a wide radius): the aggregation walks them 128 columns at a time, and the
# one-kernel block re-reads the neighbours' positions per 128-column slab
q2 = q[::37]
inds2 = K.batch_neighbors(q2, s, [len(q2)], [len(s)], 0.45)
assert inds2.shape[1] > 128
rc, out = emu.kpconv_rigid(q2, s, inds2, x, kp, w, 0.3, act=2, influence=1)
ref = torch.relu(K.kpconv_rigid(torch.from_numpy(q2), ts, torch.from_numpy(inds2.astype(np.int64)), tx, tk, tw, 0.3)).numpy()
assert rc == 0 and np.abs(out - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
print("ok")
"""


def test_kpconv_32_to_32_block_strided_queries_shadow_rows():
    """cin = cout = 32 (the KPConv of the full-resolution resnet blocks): MFMA aggregation + GEMM on 4 000+ strided queries, a
    query with only shadow neighbours, a ragged last tile, no bias, ReLU, linear and gaussian influence."""
    import subprocess
    import sys
    emu.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUSED32_CASE % {"root": root}], capture_output=True, text=True, timeout=900,
                       cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("cin,act", [(1, 0), (4, 2)])
def test_kpconv_first_layer_fused_kernel_strided_shadow_rows_no_bias(cin, act):
    """cin <= 5 with 32 | cout <= 128 runs as ONE kernel (kp_small_fused: influences + weighted sum + the [15 cin] x [cout]
    product + activation); queries != supports, a query with only shadow neighbours, a ragged last wave (nq % 8 != 0)."""
    rng = np.random.default_rng(30 + cin)
    s = synth_data.toronto3d_sphere(23, 800)
    q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.16)[0], [[50, 50, 50]]]).astype(np.float32)
    if len(q) % 8 == 0:
        q = q[1:]
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], 0.2)
    if cin == 4:        # shadows ANYWHERE in the rows (the walk is bounded by the wave's last real column, not its first shadow)
        inds = np.take_along_axis(inds, np.argsort(rng.random(inds.shape), axis=1), 1)
    x = rng.standard_normal((len(s), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, cin, 64)) * 0.3).astype(np.float32)
    rc, out = emu.kpconv_rigid(q, s, inds, x, kp, w, 0.08, act=act, slope=0.1)
    ref = K.kpconv_rigid(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08)
    if act == 2:
        ref = torch.relu(ref)
    assert rc == 0 and np.abs(out - ref.numpy()).max() <= TOL * max(1.0, float(ref.abs().max())) and (out[-1] == 0).all()


@pytest.mark.parametrize("m,k,n", [(300, 64, 128), (70, 1536, 96), (129, 15, 64), (64, 36, 8), (5, 3072, 512)])
def test_linear_bias_act_residual(m, k, n):
    rng = np.random.default_rng(m + k + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    wt = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    rc, out = emu.linear(a, wt, bias=b, residual=res, act=1, slope=0.1)
    y = a.astype(np.float64) @ wt + b + res
    ref = np.where(y > 0, y, 0.1 * y)
    assert rc == 0 and np.abs(out - ref).max() <= TOL


def test_linear_fused_upsample_concat_is_decoder_step():
    """NearestUpsampleBlock + cat + UnaryBlock (kpconv.py:283-286): [x[up[:,0]] | skip] @ W."""
    rng = np.random.default_rng(8)
    xc = rng.standard_normal((40, 64)).astype(np.float32)           # coarse features
    skip = rng.standard_normal((200, 32)).astype(np.float32)
    up = rng.integers(0, 41, (200, 7)).astype(np.int32)              # 40 == shadow -> zeros
    wt = (rng.standard_normal((96, 48)) * 0.1).astype(np.float32)
    rc, out = emu.linear(xc, wt, a2=skip, gather=up, gather_stride=7, act=1, slope=0.2)
    xs = K.closest_pool(torch.from_numpy(xc), torch.from_numpy(up).long())
    y = torch.cat([xs, torch.from_numpy(skip)], 1) @ torch.from_numpy(wt)
    ref = torch.nn.functional.leaky_relu(y, 0.2).numpy()
    assert rc == 0 and np.abs(out - ref).max() <= TOL


@pytest.mark.parametrize("m,kc,ks,n", [(200, 64, 32, 48), (1000, 128, 64, 64), (77, 256, 128, 128)])
def test_linear_split_decoder_step_gathered_residual(m, kc, ks, n):
    """The decoder step split by linearity (KPFCNN._upsample_concat_unary): (x W_x)[up[:, 0]] + skip W_skip + b, the coarse
    product entering the fine GEMM's epilogue through the upsampling index; shadow index (= number of coarse rows) adds zeros."""
    rng = np.random.default_rng(m)
    nc = max(3, m // 4)
    xc = rng.standard_normal((nc, kc)).astype(np.float32)
    skip = rng.standard_normal((m, ks)).astype(np.float32)
    up = rng.integers(0, nc + 1, (m, 5)).astype(np.int32)           # nc == shadow
    up[3, 0] = nc
    wt = (rng.standard_normal((kc + ks, n)) * 0.1).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    rc0, coarse = emu.linear(xc, wt[:kc])
    rc1, out = emu.linear(skip, wt[kc:], bias=b, residual=coarse, residual_gather=up, act=1, slope=0.2)
    xs = K.closest_pool(torch.from_numpy(xc), torch.from_numpy(up).long())
    y = torch.cat([xs, torch.from_numpy(skip)], 1) @ torch.from_numpy(wt) + torch.from_numpy(b)
    ref = torch.nn.functional.leaky_relu(y, 0.2).numpy()
    assert rc0 == 0 and rc1 == 0 and np.abs(out - ref).max() <= TOL
    rc2, fused = emu.linear(xc, wt, bias=b, a2=skip, gather=up, gather_stride=5, act=1, slope=0.2)
    assert rc2 == 0 and np.abs(out - fused).max() <= TOL


def test_gather_pools_match_reference_ops():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((500, 24)).astype(np.float32)
    inds = rng.integers(0, 501, (130, 9)).astype(np.int32)
    inds[7] = 500                                                    # a row of shadow indices only
    assert np.array_equal(emu.gather_pool(x, inds, 0), K.max_pool(torch.from_numpy(x), torch.from_numpy(inds).long()).numpy())
    assert np.array_equal(emu.gather_pool(x, inds, 1), K.closest_pool(torch.from_numpy(x), torch.from_numpy(inds).long()).numpy())


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 300), (128, 128, 130), (256, 64, 70), (64, 200, 140), (32, 32, 300)])
def test_kpconv_rigid_contraction_on_the_bf16x3_path(cin, cout, n):
    """ml3d_kpconv_rigid_bf16x3: the [15 cin, cout] contraction through gemm_tile_bf3 (three-way bf16 split), split along K where the
    problem has few row tiles (K = 960 .. 3840 here: 2 .. 30 slices + gemm_reduce); cin = 32 stays in the fused kernel and must give
    the float path's result bit for bit."""
    rng = np.random.default_rng(cin * 1000 + cout + 1)
    p, inds = _layer(21, n, 0.2)
    x = rng.standard_normal((len(p), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, cin, cout)) * (1.0 / np.sqrt(cin * 4))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    rc, out = emu.kpconv_rigid(p, p, inds, x, kp, w, 0.08, bias=b, act=1, slope=0.2, bf16x3=True)
    rc32, out32 = emu.kpconv_rigid(p, p, inds, x, kp, w, 0.08, bias=b, act=1, slope=0.2)
    assert rc == 0 and rc32 == 0
    ref = K.kpconv_rigid(torch.from_numpy(p), torch.from_numpy(p), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08)
    ref = torch.nn.functional.leaky_relu(ref + torch.from_numpy(b), 0.2).numpy()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(out - ref).max() <= TOL * scale
    if cin == 32:
        assert np.array_equal(out, out32)
    else:
        assert np.abs(out - out32).max() <= 2e-5 * scale
