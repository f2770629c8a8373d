"""CPU: the training side on hand-written HIP (csrc/train.hip, ``ml3d.ops.train``) executed on the HOST EMULATION of
the same ``.hip`` sources: every new differentiable op against torch's autograd through the reference's formulation, then the two
segmentation models in ``.train()`` mode -- every Linear, BatchNorm, gather / pool and attention stage on the HIP ops in BOTH passes --
against ONE training forward + backward of the REAL reference modules (tests/golden/train_{randlanet,kpconv}.npz, written by
oracle/gen_golden_train.py from /root/reference).  The -m gpu twins are tests/test_gpu_training.py."""
import os
import subprocess
import sys

import pytest

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")

_PRELUDE = r'''
import os, sys
ROOT = %(root)r
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
import emu_runtime
emu_runtime.install("ml3d")
'''


def _run(body, **env):
    emu.lib()
    r = subprocess.run([sys.executable, "-c", _PRELUDE % {"root": ROOT} + body], capture_output=True, text=True, timeout=1500,
                       cwd="/tmp", env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_training_ops_match_torch_autograd():
    """``ops.gemm_tn``, ``LinearFunction``, ``BatchNormActFunction`` (statistics, running buffers, fused LeakyReLU), ``GatherRowsFunction``,
    ``GatherPoolFunction`` (max with shadow rows / closest) and the fused ``AttentionStageFunction`` (d = 16 / 64 / 128 / 256 and an
    uneven c1 / c2 split, with and without bias; randlanet.py:596-605, 617, 631-637 written out in torch as the reference does):
    outputs and every input gradient."""
    _run(r'''
from ml3d import ops
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))
# gemm_tn
for m, k, n in ((1000, 70, 33), (5, 16, 16), (4097, 130, 200), (64, 8, 8)):
    a, b = T(rng.standard_normal((m, k))), T(rng.standard_normal((m, n)))
    c, s = ops.gemm_tn(a, b, with_col_sums=True)
    ref = a.double().t() @ b.double()
    assert (c - ref).abs().max() <= 1e-5 * max(1, m ** 0.5), (m, k, n, float((c - ref).abs().max()))
    assert (s - a.double().sum(0)).abs().max() <= 1e-4
print("gemm_tn ok")
# Linear
for shape, cout, bias in (((3, 50, 16, 10), 8, True), ((700, 64), 128, False), ((2, 9, 33), 5, True)):
    x = T(rng.standard_normal(shape)).requires_grad_(True)
    w = T(rng.standard_normal((cout, shape[-1])) * 0.3).requires_grad_(True)
    b = T(rng.standard_normal(cout)).requires_grad_(True) if bias else None
    g = T(rng.standard_normal(shape[:-1] + (cout,)))
    ref = F.linear(x, w, b); ref.backward(g)
    want = [x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()]
    x.grad = None; w.grad = None
    if b is not None: b.grad = None
    out = ops.LinearFunction.apply(x, w, b); out.backward(g)
    assert (out - ref).abs().max() <= 1e-5
    got = [x.grad, w.grad, None if b is None else b.grad]
    for a_, b_ in zip(got, want):
        if b_ is not None:
            assert (a_ - b_).abs().max() <= 2e-5 * max(1.0, float(b_.abs().max())), float((a_ - b_).abs().max())
print("linear ok")
# BatchNorm + act
for shape, slope in (((4, 30, 16, 8), 0.2), ((500, 64), None), ((3, 7, 300), 0.1), ((10, 5), 0.0), ((2, 512), 0.1), ((3, 1024), None), ((3, 340), 0.2)):   # (last three: fewer grid threads than channels -- ADVICE r5)
    c = shape[-1]
    x = T(rng.standard_normal(shape) * 2 + 1).requires_grad_(True)
    gam = T(rng.random(c) + 0.5).requires_grad_(True); bet = T(rng.standard_normal(c)).requires_grad_(True)
    rm, rv = T(rng.standard_normal(c)), T(rng.random(c) + 0.5)
    rm2, rv2 = rm.clone(), rv.clone()
    g = T(rng.standard_normal(shape))
    y = F.batch_norm(x.reshape(-1, c), rm, rv, gam, bet, True, 0.01, 1e-6).reshape(shape)
    ref = y if slope is None else F.leaky_relu(y, slope)
    ref.backward(g)
    want = [x.grad.clone(), gam.grad.clone(), bet.grad.clone()]
    x.grad = None; gam.grad = None; bet.grad = None
    out = ops.BatchNormActFunction.apply(x, gam, bet, rm2, rv2, 0.01, 1e-6, slope)
    out.backward(g)
    few = x.numel() // c < 4        # 2-3 rows: 1 / sqrt(var) of near-equal rows amplifies float rounding (torch's own sums are float there)
    assert (out - ref).abs().max() <= (5e-4 if few else 2e-5), float((out - ref).abs().max())
    assert (rm - rm2).abs().max() <= 1e-6 and (rv - rv2).abs().max() <= 1e-6
    for a_, b_ in zip([x.grad, gam.grad, bet.grad], want):
        assert a_.shape == b_.shape and torch.isfinite(a_).all()
        assert (a_ - b_).abs().max() <= (2e-3 if few else 5e-5) * max(1.0, float(b_.abs().max())), (shape, float((a_ - b_).abs().max()), float(b_.abs().max()))
print("bn ok")
# gathers
x = T(rng.standard_normal((40, 12))).requires_grad_(True)
idx = torch.from_numpy(rng.integers(0, 41, 100).astype(np.int32))
g = T(rng.standard_normal((100, 12)))
pad = torch.cat([x, torch.zeros_like(x[:1])])
ref = pad[idx.long()]; ref.backward(g); want = x.grad.clone(); x.grad = None
out = ops.GatherRowsFunction.apply(x, idx); out.backward(g)
assert torch.equal(out, ref) and (x.grad - want).abs().max() <= 1e-5
x.grad = None
for mode in ("max", "closest"):
    inds = torch.from_numpy(rng.integers(0, 41, (30, 7)).astype(np.int32))
    g = T(rng.standard_normal((30, 12)))
    pad = torch.cat([x, torch.zeros_like(x[:1])])
    ref = pad[inds.long()].max(1)[0] if mode == "max" else pad[inds[:, 0].long()]
    ref.backward(g); want = x.grad.clone(); x.grad = None
    out = ops.GatherPoolFunction.apply(x, inds, mode); out.backward(g)
    assert torch.equal(out, ref), mode
    assert (x.grad - want).abs().max() <= 1e-5, (mode, float((x.grad - want).abs().max()))
    x.grad = None
print("gathers ok")
# attention stage
for B, n, c1, c2, bias in ((2, 37, 8, 8, True), (1, 130, 32, 32, True), (2, 9, 64, 64, False), (1, 5, 128, 128, True), (1, 21, 6, 10, True),
                           (1, 2100, 8, 8, True), (2, 530, 40, 56, True), (1, 4200, 8, 8, False)):      # (the last three: more tiles than persistent workgroups; 17 and 33 chunks of private partial sums)
    K, d = 16, c1 + c2
    f = T(rng.standard_normal((B, n, c1))).requires_grad_(True)
    enc = T(rng.standard_normal((B, n, K, c2))).requires_grad_(True)
    idx = torch.from_numpy(rng.integers(0, n, (B, n, K)).astype(np.int32))
    w = T(rng.standard_normal((d, d)) * 0.3).requires_grad_(True)
    b = T(rng.standard_normal(d)).requires_grad_(True) if bias else None
    g = T(rng.standard_normal((B, n, d)))
    x = torch.cat([f[torch.arange(B)[:, None, None], idx.long()], enc], -1)
    s = F.linear(x, w, b)
    ref = (torch.softmax(s, dim=-2) * x).sum(-2)
    ref.backward(g)
    want = [f.grad.clone(), enc.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()]
    f.grad = None; enc.grad = None; w.grad = None
    if b is not None: b.grad = None
    out = ops.AttentionStageFunction.apply(f, enc, idx, w, b)
    out.backward(g)
    assert (out - ref).abs().max() <= 2e-5, (d, float((out - ref).abs().max()))
    got = [f.grad, enc.grad, w.grad, None if b is None else b.grad]
    for name, a_, b_ in zip("f enc w b".split(), got, want):
        if b_ is not None:
            tol = 5e-5 * max(1.0, float(b_.abs().max())) if name != "b" else 2e-4
            assert (a_ - b_).abs().max() <= tol, (d, name, float((a_ - b_).abs().max()), float(b_.abs().max()))
    print("attention stage ok", B, n, c1, c2, flush=True)
print("ok")
''')


_MODEL_CHECK = r'''
loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
def check(m, logits, loss, g, n_grads, tol):
    assert logits.requires_grad and np.abs(logits.detach().numpy() - g["logits"]).max() <= 1e-4
    assert abs(float(loss) - float(g["loss"])) <= 1e-5
    loss.backward()
    named = dict(m.named_parameters())
    checked, worst = 0, 0.0
    for key in g.files:
        if key.startswith("grad:"):
            want, got = g[key], named[key[5:]].grad.numpy()
            assert got.shape == want.shape
            err = float(np.abs(got - want).max())
            assert err <= max(2e-6, tol * float(np.abs(want).max())), (key, err, float(np.abs(want).max()))
            worst = max(worst, err / max(1e-12, float(np.abs(want).max())))
            checked += 1
    assert checked == n_grads
    return worst
'''


@pytest.mark.parametrize("path", ["hip", "torch"])
def test_randlanet_training_on_hip_ops_matches_the_reference(path):
    """RandLANet in ``.train()`` mode against the REAL reference's training forward + backward (logits <= 1e-4, loss <= 1e-5, 12
    parameter gradients <= 1e-3 of each tensor's largest entry, the BatchNorm running mean) with ``ML3D_TRAIN_OPS=hip`` (Linear /
    BatchNorm / gathers / fused attention stages on csrc/train.hip) and ``=torch`` (the A/B side)."""
    _run(_MODEL_CHECK + r'''
from oracle import randlanet_ref as R
from oracle.gen_golden_train import RANDLA_TRAIN_CFG, randla_train_inputs
from ml3d.torch.models import RandLANet
g = np.load(os.path.join(ROOT, "tests", "golden", "train_randlanet.npz"))
cfg = dict(RANDLA_TRAIN_CFG)
m = RandLANet(**cfg, device="cpu")
m.load_state_dict(R.make_state_dict(cfg, 55))
m.train()
m.fc1[2].eval()
pts, feats, labels = randla_train_inputs()
logits = m({"coords": [torch.from_numpy(pts)], "features": torch.from_numpy(feats)})
loss, lab, _ = m.get_loss(loss_obj, logits, {"data": {"labels": torch.from_numpy(labels)}}, "cpu")
assert int(lab.numel()) == int(g["n_valid"])
worst = check(m, logits, loss, g, 12, 1e-3)
assert np.abs(m.bn0.running_mean.numpy() - g["running_mean:bn0"]).max() <= 1e-5
print("ok, worst relative gradient error %.2g" % worst)
''', ML3D_TRAIN_OPS=path)


@pytest.mark.parametrize("path", ["hip", "torch"])
def test_kpfcnn_training_on_hip_ops_matches_the_reference(path):
    """KPFCNN (rigid) in ``.train()`` mode against the REAL reference's training forward + backward, both paths of ``ML3D_TRAIN_OPS``;
    KPConv's weight products run on ``ml3d_linear`` / ``ml3d_gemm_tn`` in either."""
    _run(_MODEL_CHECK + r'''
from ml3d.torch.dataloaders import kpconv_input_features
from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
from oracle import kpconv_ref as K
from oracle.gen_golden_train import TRAIN_CFG, train_inputs
g = np.load(os.path.join(ROOT, "tests", "golden", "train_kpconv.npz"))
cfg = dict(TRAIN_CFG)
m = KPFCNN(**cfg, device="cpu")
m.load_state_dict(K.make_state_dict(cfg, 77))
spheres, cols, labels = train_inputs()
pts, columns = np.concatenate(spheres), np.concatenate(cols)
np.random.seed(31)
batch = KPConvBatch(pts, [len(s) for s in spheres], cfg, features=kpconv_input_features(pts, columns, cfg["in_features_dim"]).astype(np.float32),
                    device="cpu")
batch.labels = torch.from_numpy(np.concatenate(labels).astype(np.int64))
m.train()
logits = m(batch)
loss, lab, scores = m.get_loss(loss_obj, logits, {"data": batch}, "cpu")
assert int(lab.numel()) == int(g["n_valid"])
worst = check(m, logits, loss, g, 12, 1e-3)
rm = dict(m.named_buffers())["encoder_blocks.0.batch_norm.batch_norm.running_mean"].numpy()
assert np.abs(rm - g["running_mean:encoder_blocks.0"]).max() <= 1e-5
print("ok, worst relative gradient error %.2g" % worst)
''', ML3D_TRAIN_OPS=path)


def test_unsupported_stage_shapes_fall_back_and_bad_arguments_are_refused():
    """A stage wider than the fused kernel (d = 512 of the 5-layer configs' last level) keeps the unfused ops inside the HIP path and still
    agrees with the torch-autograd path; the C entries refuse what they cannot run (ML3D_E_UNSUPPORTED = -4) or do not understand
    (ML3D_E_INVALID = -1) instead of mis-computing."""
    _run(r'''
import ctypes as C
from ml3d import ops, _abi
from ml3d.torch.models import RandLANet
assert ops.attention_stage_supported(16, 8, 8) and ops.attention_stage_supported(16, 128, 128) and ops.attention_stage_supported(16, 100, 28)
assert not ops.attention_stage_supported(16, 256, 256) and not ops.attention_stage_supported(8, 8, 8) and not ops.attention_stage_supported(16, 7, 8)
assert not ops.attention_stage_supported(16, 200, 56)            # (c1 > 160 above d = 128: the f half of the direct term must fit in LDS)
lib = _abi.get()
z = np.zeros(64, np.float32)
i = np.zeros(64, np.int32)
p = lambda a: a.ctypes.data
assert lib.ml3d_randla_attention_stage(p(z), p(z), p(i), p(z), None, 1, 1, 8, 2, 2, p(z), None) == -4          # K != 16
assert lib.ml3d_randla_attention_stage(p(z), p(z), p(i), p(z), None, 1, 1, 16, 300, 300, p(z), None) == -4     # d > 256
assert lib.ml3d_randla_attention_stage(p(z), p(z), p(i), p(z), None, 1, 1, 16, 3, 2, p(z), None) == -4         # d odd
assert lib.ml3d_randla_attention_stage(p(z), p(z), p(i), p(z), None, 1, 1, 16, 0, 2, p(z), None) == -1
assert lib.ml3d_randla_attention_stage(None, p(z), p(i), p(z), None, 1, 1, 16, 2, 2, p(z), None) == -1
assert lib.ml3d_gemm_tn(p(z), 4, p(z), 4, 2, 4, 4, None, 4, None, None) == -1 and lib.ml3d_gemm_tn(p(z), 2, p(z), 4, 2, 4, 4, p(z), 4, None, None) == -1
assert lib.ml3d_batchnorm_train_forward(p(z), 0, 4, None, None, 1e-5, 0, 0.0, p(z), p(z), p(z), p(z), p(z), 1024, None) == -1
assert lib.ml3d_batchnorm_train_forward(p(z), 4, 4, None, None, 1e-5, 0, 0.0, p(z), p(z), p(z), p(z), p(z), 8, None) == -2      # workspace too small
assert lib.ml3d_gather_rows(p(z), 4, 4, p(i), 0, 4, p(z), None) == -1
# a 2-layer net whose second level is 512 wide: stage 1 fused, stage 2 on the unfused Functions -- both inside ML3D_TRAIN_OPS=hip
cfg = dict(num_neighbors=16, num_layers=2, num_points=256, num_classes=5, sub_sampling_ratio=[4, 4], in_channels=3, dim_features=8,
           dim_output=[16, 512])
torch.manual_seed(3)
m = RandLANet(**cfg, device="cpu")
m.train()
m.fc1[2].eval()
rng = np.random.default_rng(2)
pts = torch.from_numpy(rng.random((2, 256, 3)).astype(np.float32) * 4)
lab = torch.from_numpy(rng.integers(0, 5, (2, 256)))
res = {}
for path in ("torch", "hip"):
    os.environ["ML3D_TRAIN_OPS"] = path
    m.zero_grad(set_to_none=True)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    logits = m({"coords": [pts], "features": pts.clone()})
    loss = F.cross_entropy(logits.reshape(-1, 5), lab.reshape(-1))
    loss.backward()
    res[path] = (logits.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None})
    m.load_state_dict(sd)
a, b = res["torch"], res["hip"]
assert (a[0] - b[0]).abs().max() <= 1e-4 * max(1.0, float(a[0].abs().max()))
assert set(a[1]) == set(b[1]) and len(a[1]) > 40
worst = max(float((a[1][k] - b[1][k]).abs().max()) / max(1e-6, float(a[1][k].abs().max())) for k in a[1] if float(a[1][k].abs().max()) > 1e-7)
assert worst <= 2e-3, worst
print("ok", worst)
''')


def test_deformed_kpconv_aggregation_matches_torch_autograd():
    """``ops.KPConvDeformedFunction`` (the deformable KPConv's aggregation with per-query kernel points, kpconv.py:1011-1066, 1105-1137)
    against the reference's formulation on torch's autograd -- [Nq, H, K] distances, clamped linear influences, the [Nq, H, Cin] gather, one
    matmul: the output, the feature gradient and the gradient of the deformed kernel points (what trains the offset convolution); shadow
    neighbours in the rows, cin below / across / above one 64-lane chunk.  (The model-level pin is
    tests/test_emulated_api.py::test_deformable_kpfcnn_training_forward_regulariser_and_gradients_match_the_reference, which runs on this path.)"""
    _run(r'''
from ml3d import ops
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))
for nq, ns, H, cin in ((40, 60, 9, 8), (25, 25, 14, 70), (7, 30, 5, 130)):
    q = T(rng.random((nq, 3))); s_ = T(rng.random((ns, 3)))
    inds = torch.from_numpy(rng.integers(0, ns + 3, (nq, H)).astype(np.int32))       # (values >= ns: shadow neighbours)
    x = T(rng.standard_normal((ns, cin))).requires_grad_(True)
    kp = T(rng.standard_normal((15, 3)) * 0.15)
    dkp = (kp[None] + T(rng.standard_normal((nq, 15, 3)) * 0.05)).requires_grad_(True)
    ext = 0.35
    g = T(rng.standard_normal((nq, 15 * cin)))
    far = torch.cat([s_, torch.zeros_like(s_[:1]) + 1e6], 0)
    ii = inds.long().clamp(max=ns)
    nb = far[ii] - q.unsqueeze(1)
    sq = ((nb.unsqueeze(2) - dkp.unsqueeze(1)) ** 2).sum(3)
    w = torch.clamp(1 - torch.sqrt(sq) / ext, min=0.0).transpose(1, 2)
    nx = torch.cat([x, torch.zeros_like(x[:1])], 0)[ii]
    ref = torch.matmul(w, nx).reshape(nq, 15 * cin)
    ref.backward(g)
    want = [x.grad.clone(), dkp.grad.clone()]
    x.grad = None; dkp.grad = None
    out = ops.KPConvDeformedFunction.apply(x, dkp, q, s_, inds, ext)
    out.backward(g)
    assert (out - ref).abs().max() <= 2e-5, float((out - ref).abs().max())
    for name, a, b in zip(("x", "dkp"), (x.grad, dkp.grad), want):
        assert (a - b).abs().max() <= 5e-5 * max(1.0, float(b.abs().max())), (name, float((a - b).abs().max()), float(b.abs().max()))
    print("deformed aggregation ok", nq, ns, H, cin, "max|grad dkp|", float(want[1].abs().max()), flush=True)
for nq, ns, H, cin in ((3, 0, 0, 4), (0, 5, 3, 4), (4, 6, 0, 4)):                   # no supports / no queries / no neighbour columns
    q, s_ = T(rng.random((nq, 3))), T(rng.random((ns, 3)))
    inds = torch.from_numpy(rng.integers(0, ns + 3, (nq, H)).astype(np.int32))
    x = T(rng.standard_normal((ns, cin))).requires_grad_(True)
    dkp = T(rng.standard_normal((nq, 15, 3)) * 0.2).requires_grad_(True)
    out = ops.KPConvDeformedFunction.apply(x, dkp, q, s_, inds, 0.35)
    out.square().sum().backward()
    assert out.shape == (nq, 15 * cin) and float(out.abs().sum()) == 0.0 and float(dkp.grad.abs().sum()) == 0.0 and float(x.grad.abs().sum()) == 0.0
print("ok")
''')


def test_randlanet_with_the_semantickitti_widths_matches_the_reference():
    """RandLANet with randlanet_semantickitti.yml's widths (stages of 16 / 64 / 128 / 256 channels: all four LDS classes of the fused
    attention kernels) in ``.train()`` mode on the HIP ops against the REAL reference's training forward + backward
    (tests/golden/train_randlanet_wide.npz): logits, loss, 39 gradients -- the score Linears' weights of all eight attentive poolings among
    them -- within 1e-3 of each tensor's largest entry, one running variance."""
    _run(_MODEL_CHECK + r'''
from oracle import randlanet_ref as R
from oracle.gen_golden_train import RANDLA_WIDE_TRAIN_CFG, randla_wide_train_inputs
from ml3d.torch.models import RandLANet
g = np.load(os.path.join(ROOT, "tests", "golden", "train_randlanet_wide.npz"))
cfg = dict(RANDLA_WIDE_TRAIN_CFG)
m = RandLANet(**cfg, device="cpu")
m.load_state_dict(R.make_state_dict(cfg, 56))
m.train()
m.fc1[2].eval()
pts, feats, labels = randla_wide_train_inputs()
logits = m({"coords": [torch.from_numpy(pts)], "features": torch.from_numpy(feats)})
loss, lab, _ = m.get_loss(loss_obj, logits, {"data": {"labels": torch.from_numpy(labels)}}, "cpu")
assert int(lab.numel()) == int(g["n_valid"])
worst = check(m, logits, loss, g, 39, 1e-3)
assert np.abs(m.encoder[3].pool2.mlp.batch_norm.running_var.numpy() - g["running_var:encoder.3.pool2.mlp"]).max() <= 1e-5
print("ok, worst relative gradient error %.2g" % worst)
''', ML3D_TRAIN_OPS="hip")


def test_offset_regulariser_matches_the_reference_formulation_on_torch_autograd():
    """``ops.OffsetRegulariserFunction`` (round 6: ``p2p_fitting_regularizer``, kpconv.py:2167-2206, with the ``min_d2`` of
    kpconv.py:1058-1074, value + gradient in one HIP kernel) against the reference's formulation written out in torch -- the
    [Nq, H, K] distance tensor, its min over the neighbours, the per-kernel-point loop with detached others: both L1 terms,
    ``min_d2`` and the gradient with respect to the deformed kernel points; shadow neighbours in the rows, a block of 16 queries
    plus a partial one, K = 15 and a smaller K."""
    _run(r'''
from ml3d import ops
rng = np.random.default_rng(1)
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))
l1 = torch.nn.L1Loss()
for nq, ns, H, K in ((40, 60, 9, 15), (16, 25, 14, 15), (7, 30, 5, 6), (5, 9, 0, 15)):
    q = T(rng.random((nq, 3))); s_ = T(rng.random((ns, 3)))
    inds = torch.from_numpy(rng.integers(0, ns + 3, (nq, H)).astype(np.int32))       # (values >= ns: shadow neighbours)
    dkp = (T(rng.standard_normal((K, 3)) * 0.3)[None] + T(rng.standard_normal((nq, K, 3)) * 0.1)).requires_grad_(True)
    ext, rep_ext = 0.35, 1.2
    far = torch.cat([s_, torch.zeros_like(s_[:1]) + 1e6], 0)
    if H:
        nb = far[inds.long().clamp(max=ns)] - q.unsqueeze(1)
        min_d2 = ((nb.unsqueeze(2) - dkp.unsqueeze(1)) ** 2).sum(3).min(1)[0]
    else:
        min_d2 = torch.zeros((nq, K)) + 0 * dkp.sum()
    d2 = min_d2 / ext ** 2
    fitting = l1(d2, torch.zeros_like(d2))
    locs = dkp / ext
    repulsive = 0
    for i in range(K):
        others = torch.cat([locs[:, :i], locs[:, i + 1:]], 1).detach()
        dist = torch.sqrt(((others - locs[:, i:i + 1]) ** 2).sum(2))
        rep = (torch.clamp_max(dist - rep_ext, 0.0) ** 2).sum(1)
        repulsive = repulsive + l1(rep, torch.zeros_like(rep)) / K
    ref = 2 * fitting + repulsive
    ref.backward()
    want = dkp.grad.clone(); dkp.grad = None
    terms, md2 = ops.OffsetRegulariserFunction.apply(dkp, q, s_, inds, ext, rep_ext)
    got = 2 * terms[0] + terms[1]
    got.backward()
    assert abs(float(terms[0]) - float(fitting)) <= 1e-5 * max(1.0, abs(float(fitting))), (float(terms[0]), float(fitting))
    assert abs(float(terms[1]) - float(repulsive)) <= 1e-5 * max(1.0, abs(float(repulsive))), (float(terms[1]), float(repulsive))
    if H:
        assert (md2 - min_d2.detach()).abs().max() <= 1e-5 * max(1.0, float(min_d2.abs().max()))
    assert (dkp.grad - want).abs().max() <= 2e-5 * max(1e-3, float(want.abs().max())), (float((dkp.grad - want).abs().max()), float(want.abs().max()))
    print("regulariser ok", nq, ns, H, K, float(ref), flush=True)
print("ok")
''')
