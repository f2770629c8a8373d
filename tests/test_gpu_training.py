"""GPU: the training side (narrow) of all three models.  Rigid KPConv: KPFCNN in train mode -- every KPConv through
``ops.KPConvFunction`` (HIP aggregation forward, hand-written HIP scatter backward), BatchNorm on batch statistics -- against ONE
forward + backward of the REAL reference KPFCNN on PyTorch-CPU (tests/golden/train_kpconv.npz, oracle/gen_golden_train.py:
semantic_segmentation.py:412-437's loss.backward() on the reference's module)."""
import os

import numpy as np
import pytest
import torch

from oracle import kpconv_ref as K
from oracle.gen_golden_train import TRAIN_CFG, train_inputs

pytestmark = pytest.mark.gpu


def _setup():
    from ml3d.torch.dataloaders import kpconv_input_features
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    cfg = dict(TRAIN_CFG)
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(K.make_state_dict(cfg, 77))
    m.to("cuda:0")                       # (the pipeline's model.to(device), semantic_segmentation.py:333: trainable parameters on the GPU)
    spheres, cols, labels = train_inputs()
    pts, columns = np.concatenate(spheres), np.concatenate(cols)
    np.random.seed(31)
    batch = KPConvBatch(pts, [len(s) for s in spheres], cfg, features=kpconv_input_features(pts, columns, cfg["in_features_dim"]).astype(np.float32),
                        device="cuda:0")
    batch.labels = torch.from_numpy(np.concatenate(labels).astype(np.int64))
    return cfg, m, batch


@pytest.mark.parametrize("train_ops", ["hip", "torch"])
def test_one_training_step_forward_and_gradients_match_the_reference(golden_dir, monkeypatch, train_ops):
    """(``ML3D_TRAIN_OPS=hip``, the default: Linear / BatchNorm + LeakyReLU / pools on csrc/train.hip in both passes; ``torch``: those
    modules on torch's autograd -- the A/B side.  KPConv itself is ``ops.KPConvFunction`` in either.)"""
    monkeypatch.setenv("ML3D_TRAIN_OPS", train_ops)
    g = np.load(os.path.join(golden_dir, "train_kpconv.npz"))
    cfg, m, batch = _setup()
    m.train()
    logits = m(batch)
    assert logits.requires_grad and np.abs(logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-4
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    loss, labels, scores = m.get_loss(loss_obj, logits, {"data": batch}, "cuda:0")
    assert int(labels.numel()) == int(g["n_valid"]) and abs(float(loss) - float(g["loss"])) <= 1e-5
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if not key.startswith("grad:"):
            continue
        want = g[key]
        got = named[key[5:]].grad.detach().cpu().numpy()
        assert got.shape == want.shape
        # gradients here are O(1e-2): 1e-4 of the largest entry of each tensor (atomics reorder the scatter sums)
        assert np.abs(got - want).max() <= max(2e-6, 1e-3 * float(np.abs(want).max())), (key, float(np.abs(got - want).max()), float(np.abs(want).max()))
        checked += 1
    assert checked == 12
    rm = dict(m.named_buffers())["encoder_blocks.0.batch_norm.batch_norm.running_mean"].cpu().numpy()
    assert np.abs(rm - g["running_mean:encoder_blocks.0"]).max() <= 1e-5


def test_an_optimisation_step_lowers_the_loss_and_inference_sees_the_new_weights():
    cfg, m, batch = _setup()
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    cfgp = type("P", (), dict(learning_rate=0.05, deform_lr_factor=0.1, momentum=0.0, weight_decay=0.0, scheduler_gamma=1.0))()
    m.eval()
    with torch.no_grad():
        before = m(batch).clone()
    m.train()
    opt, _ = m.get_optimizer(cfgp)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss, _, _ = m.get_loss(loss_obj, m(batch), {"data": batch}, "cuda:0")
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[2] < losses[0]
    m.eval()
    with torch.no_grad():
        after = m(batch)
    assert (after - before).abs().max() > 1e-3          # the fused inference kernels repacked the trained weights


@pytest.mark.parametrize("train_ops", ["hip", "torch"])
def test_randlanet_training_forward_and_gradients_match_the_reference(golden_dir, monkeypatch, train_ops):
    """RandLANet in train mode (BatchNorm on batch statistics, random_sample through ops.GatherMaxFunction = HIP forward +
    hand-written HIP backward, the HIP neighbour pyramid feeding the indices) against one forward + backward of the REAL
    reference module (tests/golden/train_randlanet.npz).  ``ML3D_TRAIN_OPS=hip`` (default): every Linear, BatchNorm + LeakyReLU,
    nearest_interpolation and the FUSED attention stages on csrc/train.hip in both passes; ``torch``: rounds 3-4's formulation."""
    monkeypatch.setenv("ML3D_TRAIN_OPS", train_ops)
    from oracle import randlanet_ref as R
    from oracle.gen_golden_train import RANDLA_TRAIN_CFG, randla_train_inputs
    from ml3d.torch.models import RandLANet
    g = np.load(os.path.join(golden_dir, "train_randlanet.npz"))
    cfg = dict(RANDLA_TRAIN_CFG)
    m = RandLANet(**cfg, device="cuda:0")
    m.load_state_dict(R.make_state_dict(cfg, 55))
    m.to("cuda:0")
    m.train()
    m.fc1[2].eval()                       # (Dropout: a device-specific random stream; off on both sides)
    pts, feats, labels = randla_train_inputs()
    logits = m({"coords": [torch.from_numpy(pts).cuda()], "features": torch.from_numpy(feats).cuda()})
    assert logits.requires_grad and np.abs(logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-4
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    loss, lab, _ = m.get_loss(loss_obj, logits, {"data": {"labels": torch.from_numpy(labels)}}, "cuda:0")
    assert int(lab.numel()) == int(g["n_valid"]) and abs(float(loss) - float(g["loss"])) <= 1e-5
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("grad:"):
            want, got = g[key], named[key[5:]].grad.detach().cpu().numpy()
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= max(2e-6, 1e-3 * float(np.abs(want).max())), (key, float(np.abs(got - want).max()), float(np.abs(want).max()))
            checked += 1
    assert checked == 12
    assert np.abs(m.bn0.running_mean.cpu().numpy() - g["running_mean:bn0"]).max() <= 1e-5
    # and the op alone: the gradient of random_sample lands on the first maximal neighbour
    from ml3d import ops
    f = torch.randn((2, 64, 8), device="cuda", requires_grad=True)
    idx = torch.randint(0, 64, (2, 64, 16), device="cuda", dtype=torch.int32)
    out = ops.GatherMaxFunction.apply(f, idx, 16)
    ref = f[torch.arange(2, device="cuda")[:, None, None], idx[:, :16].long()].max(2)[0]
    assert torch.equal(out, ref)
    gr = torch.randn_like(out)
    out.backward(gr)
    g1 = f.grad.clone(); f.grad = None
    ref.backward(gr)
    assert (g1 - f.grad).abs().max() <= 1e-6


def test_pointpillars_training_forward_and_gradients_match_the_reference(golden_dir):
    """``PointPillars`` in ``.train()`` mode on the MI355X (voxelize on the HIP ops; PFN / scatter / SECOND / FPN / heads on
    torch's autograd with this class's own modules) against the REAL reference model's training forward + backward
    (tests/golden/train_pointpillars.npz): head maps, the three ``get_loss`` terms, parameter gradients."""
    import synth_weights
    from ml3d.torch.models import PointPillars
    from oracle import pointpillars_ref as P
    from oracle.gen_golden_train import PP_LOSS_CFG, pp_train_inputs
    g = np.load(os.path.join(golden_dir, "train_pointpillars.npz"))
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    m = PointPillars(device="cuda:0", loss=PP_LOSS_CFG, **cfg)
    m.load_state_dict(P.make_state_dict(cfg, 21))
    m.to("cuda:0")
    m.train()
    clouds, boxes, labels = pp_train_inputs()

    class In:
        point = [torch.from_numpy(c).cuda() for c in clouds]
        bboxes = [b.cuda() for b in boxes]
    In.labels = [l.cuda() for l in labels]
    maps = m(In)
    for name, t in zip(("cls", "reg", "dir"), maps):
        want = g[name]
        assert t.requires_grad and np.abs(t.detach().cpu().numpy()[:, :, ::2, ::2] - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), name
    terms = m.get_loss(maps, In)
    got = np.array([float(terms["loss_cls"]), float(terms["loss_bbox"]), float(terms["loss_dir"])])
    assert np.abs(got - g["loss"]).max() <= 1e-4 * np.abs(g["loss"]).max(), (got, g["loss"])
    sum(terms.values()).backward()
    named = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("grad:"):
            want, have = g[key], named[key[5:]].grad.detach().cpu().numpy()
            # (5e-3 of each tensor's largest entry: the pseudo-trained heads saturate -- loss_cls is O(400) -- so the BatchNorm
            #  bias gradients are sums of large terms that cancel, and MIOpen's reduction order is not PyTorch-CPU's: 3e-3
            #  measured on backbone.blocks.2.1.bias, <= 1e-3 everywhere on the CPU run of tests/test_emulated_api.py)
            assert np.abs(have - want).max() <= 5e-3 * float(np.abs(want).max()), (key, float(np.abs(have - want).max()))
            checked += 1
    assert checked == 11
    m.eval()
    with torch.no_grad():
        out = m(In)
    assert out[0].shape == maps[0].shape and not out[0].requires_grad


@pytest.mark.parametrize("train_ops", ["hip", "torch"])
def test_deformable_kpfcnn_training_matches_the_reference(golden_dir, monkeypatch, train_ops):
    """KPFCNN with three DEFORMABLE, modulated blocks in ``.train()`` mode on the MI355X (offset convolutions through
    ``ops.KPConvFunction``; the deformed convolutions through ``ops.KPConvDeformedFunction`` -- HIP aggregation + hand-written adjoint
    with respect to the features and the deformed kernel points -- with ``ML3D_TRAIN_OPS=hip``, on torch's autograd with ``=torch``)
    against the REAL reference's training forward + backward (tests/golden/train_kpconv_deform.npz): logits, cross entropy, the
    point-to-point offset regulariser, gradients."""
    monkeypatch.setenv("ML3D_TRAIN_OPS", train_ops)
    from ml3d.torch.dataloaders import kpconv_input_features
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    from oracle.gen_golden_train import DEFORM_TRAIN_CFG, deform_train_inputs
    g = np.load(os.path.join(golden_dir, "train_kpconv_deform.npz"))
    cfg = dict(DEFORM_TRAIN_CFG)
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(K.make_state_dict(cfg, 78))
    m.to("cuda:0")
    spheres, cols, labels = deform_train_inputs()
    pts, columns = np.concatenate(spheres), np.concatenate(cols)
    np.random.seed(32)
    batch = KPConvBatch(pts, [len(s) for s in spheres], cfg,
                        features=kpconv_input_features(pts, columns, cfg["in_features_dim"]).astype(np.float32), device="cuda:0")
    batch.labels = torch.from_numpy(np.concatenate(labels).astype(np.int64))
    m.train()
    logits = m(batch)
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-4 * max(1.0, float(np.abs(g["logits"]).max()))
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    loss, lab, scores = m.get_loss(loss_obj, logits, {"data": batch}, "cuda:0")
    assert abs(float(m.output_loss) - float(g["output_loss"])) <= 1e-5
    assert abs(float(m.reg_loss) - float(g["reg_loss"])) <= 1e-4 * float(g["reg_loss"])
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("grad:"):
            want, have = g[key], named[key[5:]].grad.detach().cpu().numpy()
            assert np.abs(have - want).max() <= 2e-3 * float(np.abs(want).max()), (key, float(np.abs(have - want).max()))
            checked += 1
    assert checked == 11
    # the validation loop of run_train: an EVAL-mode forward (fused inference kernels), then get_loss -- the regulariser must be
    # the CURRENT batch's (kpconv.py:1058,1074 set min_d2 / deformed_KP in eval mode too), not the training forward's 7.0
    m.eval()
    with torch.no_grad():
        logits_e = m(batch)
        m.get_loss(loss_obj, logits_e, {"data": batch}, "cuda:0")
    assert np.abs(logits_e.cpu().numpy() - g["eval_logits"]).max() <= 1e-4 * max(1.0, float(np.abs(g["eval_logits"]).max()))
    assert abs(float(m.output_loss) - float(g["eval_output_loss"])) <= 1e-5
    assert abs(float(m.reg_loss) - float(g["eval_reg_loss"])) <= 1e-4 * float(g["eval_reg_loss"])
    assert abs(float(g["eval_reg_loss"]) - float(g["reg_loss"])) > 1.0          # (the two really differ on this fixture)



# ---- the training ops of csrc/train.hip against torch's own CUDA autograd, at sizes that span many workgroups --------------------
def _close(a, b, tol):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_linear_batchnorm_gather_functions_match_torch_autograd():
    """``ops.LinearFunction`` (forward ml3d_linear, backward ml3d_linear + ml3d_gemm_tn), ``ops.BatchNormActFunction`` (batch
    statistics in double, fused LeakyReLU, running buffers), ``ops.GatherRowsFunction`` / ``ops.GatherPoolFunction`` against the
    same expressions on torch's autograd (rocBLAS / MIOpen-free elementwise reference), 2e5 rows."""
    from ml3d import ops
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for shape, cout, bias in (((4, 11264, 16, 10), 8, True), ((200000, 64), 128, False), ((3, 999, 96), 19, True)):
        x = rn(*shape).requires_grad_(True)
        w = (rn(cout, shape[-1]) * 0.3).requires_grad_(True)
        b = rn(cout).requires_grad_(True) if bias else None
        gy = rn(*shape[:-1], cout)
        ref = F.linear(x.double(), w.double(), None if b is None else b.double())
        ref.backward(gy.double())
        want = [x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()]
        x.grad = w.grad = None
        if b is not None:
            b.grad = None
        out = ops.LinearFunction.apply(x, w, b)
        out.backward(gy)
        assert _close(out, ref.float(), 2e-5)
        rows = x.numel() // shape[-1]
        for got, wnt in zip([x.grad, w.grad, None if b is None else b.grad], want):
            if wnt is not None:
                assert _close(got, wnt, 2e-6 * max(1.0, rows ** 0.5)), (shape, float((got - wnt).abs().max()), float(wnt.abs().max()))
    for shape, slope in (((4, 2816, 16, 64), 0.2), ((300000, 32), None), ((7, 5), 0.1), ((2, 512), 0.1), ((3, 1024), None)):
        c = shape[-1]
        x = (rn(*shape) * 2 + 1).requires_grad_(True)
        gam, bet = (torch.rand(c, device="cuda", generator=g) + 0.5).requires_grad_(True), rn(c).requires_grad_(True)
        rm, rv = rn(c), torch.rand(c, device="cuda", generator=g) + 0.5
        rm2, rv2 = rm.clone(), rv.clone()
        gy = rn(*shape)
        y = F.batch_norm(x.reshape(-1, c), rm, rv, gam, bet, True, 0.01, 1e-6).reshape(shape)
        ref = y if slope is None else F.leaky_relu(y, slope)
        ref.backward(gy)
        want = [x.grad.clone(), gam.grad.clone(), bet.grad.clone()]
        x.grad = gam.grad = bet.grad = None
        out = ops.BatchNormActFunction.apply(x, gam, bet, rm2, rv2, 0.01, 1e-6, slope)
        out.backward(gy)
        few = x.numel() // c < 4    # 2-3 rows: 1 / sqrt(var) of near-equal rows amplifies float rounding
        assert _close(out, ref, 5e-4 if few else 2e-5) and _close(rm2, rm, 1e-5) and _close(rv2, rv, 1e-5)
        for got, wnt in zip([x.grad, gam.grad, bet.grad], want):
            assert got.shape == wnt.shape and torch.isfinite(got).all()
            assert _close(got, wnt, 2e-3 if few else 2e-4), (shape, float((got - wnt).abs().max()), float(wnt.abs().max()))
    x = rn(5000, 48).requires_grad_(True)
    idx = torch.randint(0, 5001, (70000,), device="cuda", generator=g).to(torch.int32)
    gy = rn(70000, 48)
    pad = torch.cat([x, torch.zeros_like(x[:1])])
    ref = pad[idx.long()]
    ref.backward(gy)
    want = x.grad.clone(); x.grad = None
    out = ops.GatherRowsFunction.apply(x, idx)
    out.backward(gy)
    assert torch.equal(out, ref) and _close(x.grad, want, 1e-5)
    x.grad = None
    for mode in ("max", "closest"):
        inds = torch.randint(0, 5001, (20000, 23), device="cuda", generator=g).to(torch.int32)
        gy = rn(20000, 48)
        pad = torch.cat([x, torch.zeros_like(x[:1])])
        ref = pad[inds.long()].max(1)[0] if mode == "max" else pad[inds[:, 0].long()]
        ref.backward(gy)
        want = x.grad.clone(); x.grad = None
        out = ops.GatherPoolFunction.apply(x, inds, mode)
        out.backward(gy)
        assert torch.equal(out, ref) and _close(x.grad, want, 1e-5), mode
        x.grad = None


@pytest.mark.parametrize("B,n,c1,c2,bias", [(2, 11264, 8, 8, True), (4, 2816, 32, 32, True), (2, 1500, 64, 64, False), (3, 704, 128, 128, True),
                                            (1, 333, 6, 10, True)])
def test_fused_attention_stage_matches_the_unfused_formulation(B, n, c1, c2, bias):
    """``ops.AttentionStageFunction`` (one kernel per pass, no [B, N, K, d] tensor) against the reference's formulation on torch's
    autograd -- gather, concat, Linear, softmax over K, weighted sum (randlanet.py:596-605, 617, 631-637) -- in float64: the four
    stage widths of randlanet_semantickitti.yml at multi-workgroup sizes and an uneven split; output and every gradient."""
    from ml3d import ops
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + c1)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    K, d = 16, c1 + c2
    f, enc = rn(B, n, c1).requires_grad_(True), rn(B, n, K, c2).requires_grad_(True)
    idx = torch.randint(0, n, (B, n, K), device="cuda", generator=g).to(torch.int32)
    w = (rn(d, d) * 0.3).requires_grad_(True)
    b = rn(d).requires_grad_(True) if bias else None
    gy = rn(B, n, d)
    x = torch.cat([f.double()[torch.arange(B, device="cuda")[:, None, None], idx.long()], enc.double()], -1)
    s = F.linear(x, w.double(), None if b is None else b.double())
    ref = (torch.softmax(s, dim=-2) * x).sum(-2)
    ref.backward(gy.double())
    want = [f.grad.clone(), enc.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone()]
    f.grad = enc.grad = w.grad = None
    if b is not None:
        b.grad = None
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = ops.AttentionStageFunction.apply(f, enc, idx, w, b)
    out.backward(gy)
    torch.cuda.synchronize()
    # nothing of size [B, N, K, d] was allocated by the two passes: the peak stays below ONE such tensor + the outputs they return
    # (+ the backward's transient workspace, which does not grow with the number of points)
    assert torch.cuda.max_memory_allocated() - base <= 4 * (B * n * K * d + 2 * B * n * K * c2 + 4 * B * n * d) + (176 << 20)         # (+ the <= 165 MB transient workspace of private grad_weight partials)
    assert _close(out, ref.float(), 2e-5), float((out - ref.float()).abs().max())
    for name, got, wnt in zip("f enc w b".split(), [f.grad, enc.grad, w.grad, None if b is None else b.grad], want):
        if wnt is not None:
            tol = 1e-4 if name != "b" else 1e-3          # (the bias gradient is a sum of terms that cancel exactly in exact arithmetic)
            assert float((got - wnt).abs().max()) <= tol * max(1.0, float(wnt.abs().max())), (name, float((got - wnt).abs().max()), float(wnt.abs().max()))


def test_deformed_kpconv_aggregation_matches_torch_autograd():
    """``ops.KPConvDeformedFunction`` against the reference's formulation on torch's autograd: output, feature gradient and the gradient of
    the per-query kernel points, 20 000 queries, shadow neighbours in the rows.  (float32 reference: the kernel follows the reference's
    operation sequence for the influence, so the clamp's mask -- where the kernel-point gradient is discontinuous -- is the same bit for
    bit; the sums differ by rounding.  The kernel-point gradient is additionally held in the L1 norm.)"""
    from ml3d import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for nq, ns, H, cin in ((20000, 20000, 24, 64), (3000, 9000, 37, 200), (500, 500, 6, 8)):
        q = torch.rand(nq, 3, device="cuda", generator=g)
        s_ = torch.rand(ns, 3, device="cuda", generator=g)
        inds = torch.randint(0, ns + 3, (nq, H), device="cuda", generator=g).to(torch.int32)
        x = rn(ns, cin).requires_grad_(True)
        kp = rn(15, 3) * 0.05
        dkp = (kp[None] + rn(nq, 15, 3) * 0.02).requires_grad_(True)
        ext = 0.12
        gy = rn(nq, 15 * cin)
        far = torch.cat([s_, torch.zeros_like(s_[:1]) + 1e6], 0)
        ii = inds.long().clamp(max=ns)
        nb = far[ii] - q.unsqueeze(1)
        sq = ((nb.unsqueeze(2) - dkp.unsqueeze(1)) ** 2).sum(3)
        w = torch.clamp(1 - torch.sqrt(sq) / ext, min=0.0).transpose(1, 2)
        nx = torch.cat([x, torch.zeros_like(x[:1])], 0)[ii]
        ref = torch.matmul(w, nx).reshape(nq, 15 * cin)
        ref.backward(gy)
        want = [x.grad.clone(), dkp.grad.clone()]
        x.grad = dkp.grad = None
        out = ops.KPConvDeformedFunction.apply(x, dkp, q, s_, inds, ext)
        out.backward(gy)
        assert _close(out, ref, 5e-5)
        assert _close(x.grad, want[0], 2e-4), float((x.grad - want[0]).abs().max())
        err = (dkp.grad - want[1]).abs()
        assert float(err.sum()) <= 1e-4 * float(want[1].abs().sum()), (float(err.sum()), float(want[1].abs().sum()))
        assert int((err > 1e-3 * max(1.0, float(want[1].abs().max()))).sum()) <= 3, int((err > 1e-3 * float(want[1].abs().max())).sum())


def test_randlanet_with_the_semantickitti_widths_matches_the_reference(golden_dir):
    """RandLANet with randlanet_semantickitti.yml's widths (stages of 16 / 64 / 128 / 256 channels: all four LDS classes of the fused
    attention kernels) in train mode on the MI355X against the REAL reference's training forward + backward
    (tests/golden/train_randlanet_wide.npz, oracle/gen_golden_train.py): logits, loss, 39 gradients -- the score Linears of all eight
    attentive poolings among them."""
    from oracle import randlanet_ref as R
    from oracle.gen_golden_train import RANDLA_WIDE_TRAIN_CFG, randla_wide_train_inputs
    from ml3d.torch.models import RandLANet
    g = np.load(os.path.join(golden_dir, "train_randlanet_wide.npz"))
    cfg = dict(RANDLA_WIDE_TRAIN_CFG)
    m = RandLANet(**cfg, device="cuda:0")
    m.load_state_dict(R.make_state_dict(cfg, 56))
    m.to("cuda:0")
    m.train()
    m.fc1[2].eval()
    pts, feats, labels = randla_wide_train_inputs()
    logits = m({"coords": [torch.from_numpy(pts).cuda()], "features": torch.from_numpy(feats).cuda()})
    assert logits.requires_grad and np.abs(logits.detach().cpu().numpy() - g["logits"]).max() <= 1e-4
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    loss, lab, _ = m.get_loss(loss_obj, logits, {"data": {"labels": torch.from_numpy(labels)}}, "cuda:0")
    assert int(lab.numel()) == int(g["n_valid"]) and abs(float(loss) - float(g["loss"])) <= 1e-5
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("grad:"):
            want, got = g[key], named[key[5:]].grad.detach().cpu().numpy()
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= max(2e-6, 1e-3 * float(np.abs(want).max())), (key, float(np.abs(got - want).max()), float(np.abs(want).max()))
            checked += 1
    assert checked == 39
    assert np.abs(m.encoder[3].pool2.mlp.batch_norm.running_var.cpu().numpy() - g["running_var:encoder.3.pool2.mlp"]).max() <= 1e-5


def test_offset_regulariser_matches_the_reference_formulation_on_torch_autograd():
    """``ops.OffsetRegulariserFunction`` on the hardware (wave shuffles inside 16-lane groups, the block reduction) against the
    reference's formulation of ``p2p_fitting_regularizer`` (kpconv.py:2167-2206) + ``min_d2`` (kpconv.py:1058-1074) on torch's
    autograd: both L1 terms, ``min_d2``, the gradient of the deformed kernel points."""
    from ml3d import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    l1 = torch.nn.L1Loss()
    for nq, ns, H, K in ((4000, 6000, 33, 15), (16, 25, 14, 15), (7, 30, 5, 6)):
        q = torch.rand((nq, 3), device="cuda", generator=g); s_ = torch.rand((ns, 3), device="cuda", generator=g)
        inds = torch.randint(0, ns + 3, (nq, H), device="cuda", generator=g).to(torch.int32)
        dkp = (torch.randn((1, K, 3), device="cuda", generator=g) * 0.3 + torch.randn((nq, K, 3), device="cuda", generator=g) * 0.1).requires_grad_(True)
        ext, rep_ext = 0.35, 1.2
        far = torch.cat([s_, torch.zeros_like(s_[:1]) + 1e6], 0)
        nb = far[inds.long().clamp(max=ns)] - q.unsqueeze(1)
        min_d2 = ((nb.unsqueeze(2) - dkp.unsqueeze(1)) ** 2).sum(3).min(1)[0]
        d2 = min_d2 / ext ** 2
        fitting = l1(d2, torch.zeros_like(d2))
        locs = dkp / ext
        repulsive = 0
        for i in range(K):
            others = torch.cat([locs[:, :i], locs[:, i + 1:]], 1).detach()
            dist = torch.sqrt(((others - locs[:, i:i + 1]) ** 2).sum(2))
            rep = (torch.clamp_max(dist - rep_ext, 0.0) ** 2).sum(1)
            repulsive = repulsive + l1(rep, torch.zeros_like(rep)) / K
        (2 * fitting + repulsive).backward()
        want = dkp.grad.clone(); dkp.grad = None
        terms, md2 = ops.OffsetRegulariserFunction.apply(dkp, q, s_, inds, ext, rep_ext)
        (2 * terms[0] + terms[1]).backward()
        assert abs(float(terms[0]) - float(fitting)) <= 2e-5 * max(1.0, abs(float(fitting)))
        assert abs(float(terms[1]) - float(repulsive)) <= 2e-5 * max(1.0, abs(float(repulsive)))
        assert _close(md2, min_d2.detach(), 1e-5)
        assert float((dkp.grad - want).abs().max()) <= 5e-5 * max(1e-3, float(want.abs().max())), (nq, float((dkp.grad - want).abs().max()), float(want.abs().max()))
