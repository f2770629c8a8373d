"""GPU parity: KPConv (rigid) — the GPU batcher (fixed-radius search + grid subsample) and the KPFCNN
forward on HIP kernels vs the CPU oracle and the reference-generated golden vectors (tests/golden/kpconv_*.npz,
produced by the REAL reference KPConvBatch + KPFCNN, oracle/gen_golden.py).
Tolerance: batch index matrices / pooled points bit-exact, logits max|d| <= 1e-4."""
import os

import numpy as np
import pytest
import torch

import synth_data
from oracle import kpconv_ref as K

pytestmark = pytest.mark.gpu
TOL = 1e-4
CFG = dict(K.TORONTO3D_CFG)


def _gpu_batch(spheres, np_seed=None, rotations="random"):
    from ml3d.torch.models.kpconv import KPConvBatch
    if np_seed is not None:
        np.random.seed(np_seed)
    pts = np.concatenate(spheres)
    return KPConvBatch(pts, [len(s) for s in spheres], CFG, rotations=rotations, device="cuda:0")


def _model(sd):
    from ml3d.torch.models.kpconv import KPFCNN
    m = KPFCNN(**CFG, device="cuda:0")
    m.load_state_dict(sd)
    return m.eval()


@pytest.mark.parametrize("name", ["kpconv_small", "kpconv_toronto3d"])
def test_batcher_and_forward_match_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    spheres = [synth_data.toronto3d_sphere(int(f), int(g["max_points"])) for f in g["frame_ids"]]
    batch = _gpu_batch(spheres, int(g["np_seed"]))
    # ---- batch: what the reference's KPConvBatch produced with the same np.random stream -----------------
    assert np.array_equal(batch.rotations[0], g["rot0"])
    for l in range(CFG["num_layers"]):
        nb = batch.neighbors[l].cpu().numpy().astype(np.int64)
        assert list(nb.shape) == list(g["nbr_shape%d" % l])
        assert np.int64((nb * (np.arange(nb.shape[1]) + 1)).sum()) == g["nbr_checksum%d" % l]
        assert np.array_equal(batch.lengths[l].numpy(), g["lengths"][l])
        assert np.array_equal(batch.points[l].cpu().numpy().astype(np.float64).sum(0), g["points_sum%d" % l])
    assert np.array_equal(batch.neighbors[0][:64].cpu().numpy(), g["nbr0_head"])
    assert np.array_equal(batch.pools[0][:64].cpu().numpy(), g["pool0_head"])
    assert np.array_equal(batch.upsamples[0][:64].cpu().numpy(), g["up0_head"])
    # ---- forward -----------------------------------------------------------------------------------------
    sd = K.make_state_dict(CFG, int(g["weights_seed"]))
    out = _model(sd)(batch)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == g["logits"].shape
    assert np.abs(out - g["logits"]).max() <= TOL
    assert (out.argmax(1) == g["logits"].argmax(1)).mean() >= 0.9999


@pytest.mark.parametrize("rot", ["random", None])
def test_batch_bit_exact_vs_oracle_every_matrix(rot):
    spheres = [synth_data.toronto3d_sphere(31, 4000), synth_data.toronto3d_sphere(32, 2500)]
    np.random.seed(5)
    seg = K.segmentation_inputs(np.concatenate(spheres), [len(s) for s in spheres], CFG, rotations=rot)
    batch = _gpu_batch(spheres, 5, rot)
    for l in range(CFG["num_layers"]):
        assert np.array_equal(batch.points[l].cpu().numpy(), seg["points"][l]), l
        assert np.array_equal(batch.neighbors[l].cpu().numpy(), seg["neighbors"][l]), l
        assert np.array_equal(batch.pools[l].cpu().numpy(), seg["pools"][l]), l
        assert np.array_equal(batch.upsamples[l].cpu().numpy(), seg["upsamples"][l]), l


def test_forward_accepts_the_reference_cpu_int64_batch():
    """Drop-in: the reference's own KPConvBatch hands int64 CPU tensors to model(batch)."""
    sphere = synth_data.toronto3d_sphere(33, 3000)
    np.random.seed(9)
    seg = K.segmentation_inputs(sphere, [len(sphere)], CFG)
    sd = K.make_state_dict(CFG, 77)
    feats = torch.ones((len(sphere), 1))
    ref = K.forward(sd, CFG, K.to_torch_batch(seg), feats).numpy()

    class B:
        pass
    b = B()
    tb = K.to_torch_batch(seg)
    b.points, b.neighbors, b.pools, b.upsamples, b.features = tb["points"], tb["neighbors"], tb["pools"], tb["upsamples"], feats
    out = _model(sd)(b).cpu().numpy()
    assert np.abs(out - ref).max() <= TOL


def test_other_config_shapes_in_features_5_no_reduce_fc_gaussian():
    cfg = dict(CFG, in_features_dim=5, reduce_fc=False, first_features_dim=64, KP_extent=1.2, l_relu=0.1,
               architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb_strided", "resnetb",
                             "nearest_upsample", "unary", "nearest_upsample", "unary"], num_layers=3)
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    rng = np.random.default_rng(0)
    sphere = synth_data.toronto3d_sphere(34, 2500)
    feats = np.concatenate([np.ones((len(sphere), 1), np.float32), rng.random((len(sphere), 4), dtype=np.float32)], 1)
    np.random.seed(3)
    seg = K.segmentation_inputs(sphere, [len(sphere)], cfg)
    sd = K.make_state_dict(cfg, 5)
    ref = K.forward(sd, cfg, K.to_torch_batch(seg), torch.from_numpy(feats)).numpy()
    np.random.seed(3)
    batch = KPConvBatch(sphere, [len(sphere)], cfg, features=feats, device="cuda:0")
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    out = m(batch).cpu().numpy()
    assert np.abs(out - ref).max() <= TOL


def test_pipeline_build_under_forward_is_bit_identical_to_the_sequential_loop():
    """ml3d.engine.KPConvPipeline (what bench.py --workload kpconv times): the batch build of step i + 1 on one HIP stream under
    the forward of step i on another.  Five different batches, twice through the pipeline (buffers of the caching allocator get
    reused across streams) -- logits identical to model(KPConvBatch(..)) run one after the other."""
    from ml3d.engine import KPConvPipeline
    from ml3d.torch.models.kpconv import KPConvBatch
    sd = K.make_state_dict(CFG, 77)
    m = _model(sd)
    batches = [[synth_data.toronto3d_sphere(40 + 3 * i + j, 2500 + 400 * j + 150 * i) for j in range(3)] for i in range(5)]
    dev = torch.device("cuda:0")
    inputs = [(torch.from_numpy(np.concatenate(b)).to(dev), [len(s) for s in b]) for b in batches]
    np.random.seed(5)
    seq = [m(KPConvBatch(p, l, CFG, device=dev)).cpu().numpy() for p, l in inputs for _ in (0,)]
    for rep in range(4):
        np.random.seed(5)
        pipe = KPConvPipeline(m, CFG, dev, threaded=rep >= 2)     # forward enqueued by the caller / by the worker thread
        got = []
        for p, l in inputs:
            r = pipe.submit(p, l)
            if r is not None:
                got.append(r)
        got.append(pipe.flush())
        assert pipe.flush() is None
        outs = [r.wait().cpu().numpy() for r in got]
        assert len(outs) == len(seq)
        for a, b in zip(outs, seq):
            assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("builders,fwd_streams", [(2, 1), (3, 1), (2, 2), (1, 2)])
def test_pipeline_with_several_builds_in_flight_is_bit_identical_to_the_sequential_loop(builders, fwd_streams):
    """ml3d.engine.KPConvPipelineN (what bench.py --workload kpconv times since round 5): two / three one-call batch builds in
    flight on their own HIP streams and host threads, forwards in submission order -- logits identical to
    model(KPConvBatch(..)) run one after the other with the same seed (the grid orientations are drawn at submit time)."""
    from ml3d.engine import KPConvPipelineN
    from ml3d.torch.models.kpconv import KPConvBatch
    sd = K.make_state_dict(CFG, 77)
    m = _model(sd)
    batches = [[synth_data.toronto3d_sphere(40 + 3 * i + j, 2500 + 400 * j + 150 * i) for j in range(3)] for i in range(7)]
    dev = torch.device("cuda:0")
    inputs = [(torch.from_numpy(np.concatenate(b)).to(dev), [len(s) for s in b]) for b in batches]
    np.random.seed(5)
    seq = [m(KPConvBatch(p, l, CFG, device=dev)).cpu().numpy() for p, l in inputs]
    for rep in range(2):
        np.random.seed(5)
        pipe = KPConvPipelineN(m, CFG, dev, builders=builders, forward_streams=fwd_streams)     # (forwards alternate between streams)
        got = []
        for i, (p, l) in enumerate(inputs):
            r = pipe.submit(p, l)
            assert (r is None) == (i < builders)
            if r is not None:
                got.append(r)
        got += pipe.flush()
        assert pipe.flush() == []
        outs = [r.wait().cpu().numpy() for r in got]
        assert len(outs) == len(seq)
        for a, b in zip(outs, seq):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_bench_configuration_96_spheres_through_pipeline_n_matches_the_oracle_per_sphere():
    """The configuration bench.py --workload kpconv times (bench_models.run_kpconv): 96 input spheres (~960 000 points) per batch
    through ``KPConvPipelineN(builders=2, forward_streams=2)``, FOUR batches in a row so that two one-call builds are in flight
    on their own streams beside forwards that alternate between two compute streams (bf16x3 contractions co-running with the
    builds' LDS kernels).  Only at this size are deep split-K, the 32-bit offset epilogue (wf is 1.8 GB) and the three-launch
    scans selected.  Spheres {0, 47, 95} of the FIRST and the LAST batch are checked one by one against the oracle run on that
    sphere alone with the same grid rotation: every points / neighbours / pools / upsamples matrix exact (after removing the
    item's row offset; the batch's extra columns must be shadow entries), logits <= 1e-4.

    The oracle forward gets the item's matrices AT THE BATCH'S WIDTH: ``max_pool`` (kpconv.py:821-839) takes the maximum over
    all columns, shadow entries (zero features) included, so in the reference itself a row's pooled value depends on how many
    padding columns the longest row of the batch adds (max(negatives) vs max(negatives, 0)) -- batch items are independent
    only up to that padding."""
    from ml3d.engine import KPConvPipelineN
    B = 96
    spheres = [synth_data.toronto3d_sphere(i) for i in range(B)]          # the bench's spheres (rank 0)
    lens = [len(s) for s in spheres]
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(np.concatenate(spheres)).to(dev)
    sd = K.make_state_dict(CFG, 2024)
    m = _model(sd)
    np.random.seed(0)
    pipe = KPConvPipelineN(m, CFG, dev, builders=2, forward_streams=2)
    results = []
    for _ in range(4):
        r = pipe.submit(pts, lens)
        if r is not None:
            results.append(r)
    results += pipe.flush()
    assert len(results) == 4
    L = CFG["num_layers"]
    for res in (results[0], results[-1]):
        _check_batch_items_against_the_oracle(res, spheres, lens, sd, (0, 47, 95))


def _check_batch_items_against_the_oracle(res, spheres, lens, sd, items):
    logits = res.wait().cpu().numpy()
    torch.cuda.synchronize()
    batch = res.batch
    L = CFG["num_layers"]
    blens = [batch.lengths[l].numpy().astype(np.int64) for l in range(L)]
    offs = [np.concatenate([[0], np.cumsum(x)]) for x in blens]
    assert logits.shape == (sum(lens), CFG["num_classes"]) and np.isfinite(logits).all()

    def item_rows(mat, ref, rows, s_off, s_tot, s_len, where):
        """item rows of a batch matrix, global -> local indices (shadow -> local shadow); must extend ``ref`` by shadows only"""
        got = mat[rows[0]:rows[1]].cpu().numpy().astype(np.int64)
        loc = np.where(got == s_tot, s_len, got - s_off)
        w = ref.shape[1]
        assert loc.shape[1] >= w and np.array_equal(loc[:, :w], ref) and (loc[:, w:] == s_len).all(), where
        return loc

    for i in items:
        rots = [None if R is None else R[i:i + 1] for R in batch.rotations]      # this item's grid orientations
        seg = K.segmentation_inputs(spheres[i], [lens[i]], CFG, rotations=rots)
        wide = dict(seg, neighbors=[], pools=[], upsamples=[])
        for l in range(L):
            a, b = offs[l][i], offs[l][i + 1]
            assert np.array_equal(batch.points[l][a:b].cpu().numpy(), seg["points"][l]), (i, l)
            wide["neighbors"].append(item_rows(batch.neighbors[l], seg["neighbors"][l], (a, b), a, offs[l][-1], b - a, (i, l, "n")))
            if l + 1 < L:
                c, d = offs[l + 1][i], offs[l + 1][i + 1]
                wide["pools"].append(item_rows(batch.pools[l], seg["pools"][l], (c, d), a, offs[l][-1], b - a, (i, l, "p")))
                wide["upsamples"].append(item_rows(batch.upsamples[l], seg["upsamples"][l], (a, b), c, offs[l + 1][-1], d - c,
                                                   (i, l, "u")))
            else:
                wide["pools"].append(seg["pools"][l])
                wide["upsamples"].append(seg["upsamples"][l])
        ref = K.forward(sd, CFG, K.to_torch_batch(wide), torch.ones((lens[i], 1))).numpy()
        a, b = offs[0][i], offs[0][i + 1]
        assert np.abs(logits[a:b] - ref).max() <= TOL, (i, float(np.abs(logits[a:b] - ref).max()))


@pytest.mark.parametrize("cin,n,r", [(16, 2000, 0.3), (32, 4000, 0.3), (64, 2500, 0.35), (128, 1500, 0.3), (256, 900, 0.25),
                                     (24, 1200, 0.3)])
def test_kpconv_op_rows_wider_than_64_with_shadows_anywhere(cin, n, r):
    """``ops.kpconv_rigid`` on the hardware for every aggregation kernel: the MFMA form (cin in {16, 32, 64, 128, 256}: one wave
    per query, two index registers for rows of 65 .. 128 columns, ballot for the last real column, ds_bpermute for a group's
    column) and the packed-FMA form (cin = 24).  The columns of every row are shuffled so that shadow entries sit ANYWHERE;
    strided queries; a query with only shadow neighbours; bias + LeakyReLU.  <= 1e-4 against the oracle's restatement."""
    from ml3d import ops
    rng = np.random.default_rng(cin)
    s = synth_data.toronto3d_sphere(22, n)
    q = np.concatenate([K.batch_grid_subsampling(s, [len(s)], 0.1)[0], [[50, 50, 50]]]).astype(np.float32)
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], r)
    assert 40 < inds.shape[1] <= 128 and (inds[-1] == len(s)).all()
    inds = np.take_along_axis(inds, np.argsort(rng.random(inds.shape), axis=1), 1)
    x = rng.standard_normal((len(s), cin)).astype(np.float32)
    kp = K.synthetic_kernel_points(0.2)
    w = (rng.standard_normal((15, cin, 32)) * (0.5 / np.sqrt(cin))).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    d = "cuda:0"
    out = ops.kpconv_rigid(torch.from_numpy(q).to(d), torch.from_numpy(s).to(d), torch.from_numpy(inds).to(d),
                           torch.from_numpy(x).to(d), torch.from_numpy(kp).to(d),
                           torch.from_numpy(w.reshape(15 * cin, 32)).to(d), torch.from_numpy(b).to(d), 0.08, 1, 0.2, 1)
    torch.cuda.synchronize()
    ref = K.kpconv_rigid(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(inds).long(), torch.from_numpy(x),
                         torch.from_numpy(kp), torch.from_numpy(w), 0.08)
    ref = torch.nn.functional.leaky_relu(ref + torch.from_numpy(b), 0.2).numpy()
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    assert np.abs(got[-1] - np.where(b > 0, b, 0.2 * b)).max() <= 1e-6          # only shadows: act(bias)


def test_deformable_architecture_batch_and_forward_match_the_reference_golden(golden_dir):
    """The deformable blocks of kpconv_parislille3d.yml:28-32 on a three-layer architecture (synth_weights.
    KPCONV_DEFORM_SMALL_CFG): deformable KPConv 32 -> 32 at a full and a strided block, 64 -> 64 on the coarsest layer.  The
    batch (deform radius on the deformable layers' conv / pool searches, twice the POOL radius for the upsampling) against the
    REAL reference's KPConvBatch, the logits against the REAL reference's KPFCNN.forward (oracle/gen_golden.py), <= 1e-4."""
    from ml3d.torch.models.kpconv import KPConvBatch, KPFCNN
    cfg = dict(K.KPCONV_DEFORM_SMALL_CFG)
    g = np.load(os.path.join(golden_dir, "kpconv_deform_small.npz"))
    spheres = [synth_data.toronto3d_sphere(int(f), int(g["max_points"])) for f in g["frame_ids"]]
    np.random.seed(int(g["np_seed"]))
    batch = KPConvBatch(np.concatenate(spheres), [len(s) for s in spheres], cfg, device="cuda:0")
    assert np.array_equal(batch.rotations[0], g["rot0"])
    for l in range(cfg["num_layers"]):
        nb = batch.neighbors[l].cpu().numpy().astype(np.int64)
        assert list(nb.shape) == list(g["nbr_shape%d" % l]), l
        assert np.int64((nb * (np.arange(nb.shape[1]) + 1)).sum()) == g["nbr_checksum%d" % l], l
        assert np.array_equal(batch.lengths[l].numpy(), g["lengths"][l])
    assert np.array_equal(batch.upsamples[0][:64].cpu().numpy(), g["up0_head"])
    # every matrix against the oracle's restatement of the batcher
    np.random.seed(int(g["np_seed"]))
    seg = K.segmentation_inputs(np.concatenate(spheres), [len(s) for s in spheres], cfg)
    for l in range(cfg["num_layers"]):
        for name in ("neighbors", "pools", "upsamples"):
            assert np.array_equal(getattr(batch, name)[l].cpu().numpy(), seg[name][l]), (name, l)
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(K.make_state_dict(cfg, int(g["weights_seed"])))
    out = m.eval()(batch)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == g["logits"].shape and np.abs(out - g["logits"]).max() <= TOL
    assert (out.argmax(1) == g["logits"].argmax(1)).mean() >= 0.999


@pytest.mark.parametrize("cin,n,r,modulated", [(32, 4000, 0.3, False), (64, 2500, 0.55, False), (128, 1500, 0.3, True),
                                               (512, 700, 0.3, False)])
def test_kpconv_deformable_op_matches_the_reference_formulation(cin, n, r, modulated):
    """``ops.kpconv_deformable`` on the hardware (kpconv.py:1011-1159): inner convolution -> per-query kernel points (and
    modulations) in the MFMA aggregation; rows wider than 128 columns (0.55 m); cin = 512 as two channel slices.  Against the
    oracle's restatement of the reference branch INCLUDING its in-range pruning of the neighbour lists."""
    from ml3d import ops
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
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    out = ops.kpconv_deformable(g(q), g(s), g(inds), g(x), g(kp), g(w.reshape(15 * cin, 32)), g(b), 0.08,
                                g(ow.reshape(15 * cin, od)), g(ob), 1, 0.1, 1)
    torch.cuda.synchronize()
    t = torch.from_numpy
    ref = K.kpconv_deformable(t(q), t(s), t(inds).long(), t(x), t(kp), t(w), 0.08, t(ow), t(ob), modulated)
    ref = torch.nn.functional.leaky_relu(ref + t(b), 0.1).numpy()
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_parislille3d_architecture_runs_natively_and_matches_the_oracle():
    """kpconv_parislille3d.yml's model section (five deformable blocks, KPConv widths 128 / 128 / 256 / 256 / 512,
    deform_radius 6.0): GPU batch build against the oracle's restatement of the batcher (rows of 300+ columns on the deformable
    layers: the dense search's 128-entry stash overflows and falls back, the aggregation walks the rows 128 columns at a time)
    and the forward against the oracle's (pinned to the real reference on the small deformable architecture), <= 1e-4."""
    from ml3d.torch.models.kpconv import KPConvBatch, KPFCNN
    cfg = dict(K.PARISLILLE3D_CFG)
    spheres = [synth_data.toronto3d_sphere(8, 6000), synth_data.toronto3d_sphere(9, 2500)]
    pts, lens = np.concatenate(spheres), [len(s) for s in spheres]
    np.random.seed(3)
    seg = K.segmentation_inputs(pts, lens, cfg)
    np.random.seed(3)
    batch = KPConvBatch(pts, lens, cfg, device="cuda:0")
    assert max(seg["neighbors"][l].shape[1] for l in range(5)) > 128
    for l in range(cfg["num_layers"]):
        for name in ("neighbors", "pools", "upsamples"):
            assert np.array_equal(getattr(batch, name)[l].cpu().numpy(), seg[name][l]), (name, l)
    sd = K.make_state_dict(cfg, 21)
    m = KPFCNN(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    out = m.eval()(batch)
    torch.cuda.synchronize()
    ref = K.forward(sd, cfg, K.to_torch_batch(seg), torch.ones((len(pts), 1))).numpy()
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    assert (out.cpu().numpy().argmax(1) == ref.argmax(1)).mean() >= 0.999


def test_contractions_on_the_bf16x3_kernel_match_the_f32_kernel(monkeypatch):
    """ML3D_KP_GEMM=f32 (every GEMM on the f32 MFMA kernels) against the default (the [15 cin, cout] contractions of the convolutions
    with cin >= 64 on gemm_tile_bf3, split along K on the coarse levels): logits within 2e-5 of each other relative to their scale --
    the two paths differ by float rounding only -- and the same labels.  Also the op alone at one coarse-level shape."""
    from ml3d import ops
    sd = K.make_state_dict(CFG, 7)
    spheres = [synth_data.toronto3d_sphere(i, 6000) for i in range(4)]
    batch = _gpu_batch(spheres, 3)
    out = {}
    for path in ("f32", "bf16x3"):
        monkeypatch.setenv("ML3D_KP_GEMM", path)
        m = _model(sd)
        P = m.packed_params(m.device)
        wide = [p['conv'] for p in P['enc'] if p['conv']['w'].shape[0] >= 15 * 64]
        assert wide and all((c['packed'] is not None) == (path == "bf16x3") for c in wide)
        assert all(p['conv']['packed'] is None for p in P['enc'] if p['conv']['w'].shape[0] < 15 * 64)
        out[path] = m(batch).float()
    scale = float(out["f32"].abs().max())
    assert float((out["f32"] - out["bf16x3"]).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((out["f32"].argmax(1) == out["bf16x3"].argmax(1)).float().mean()) >= 0.9999
    # the op: 128 -> 128 channels on ~2 300 queries (K = 1 920: several K slices + gemm_reduce)
    rng = np.random.default_rng(5)
    s = synth_data.toronto3d_sphere(30, 9000)
    q = K.batch_grid_subsampling(s, [len(s)], 0.08)[0].astype(np.float32)
    inds = K.batch_neighbors(q, s, [len(q)], [len(s)], 0.12)
    dev = torch.device("cuda:0")
    tq, ts, ti = torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev), torch.from_numpy(inds).to(dev).int()
    x = torch.from_numpy(rng.standard_normal((len(s), 128)).astype(np.float32)).to(dev)
    kp = torch.from_numpy(K.synthetic_kernel_points(0.12)).to(dev)
    w = torch.from_numpy((rng.standard_normal((15 * 128, 128)) * 0.03).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(128).astype(np.float32)).to(dev)
    o32 = ops.kpconv_rigid(tq, ts, ti, x, kp, w, b, 0.06)
    obf = ops.kpconv_rigid(tq, ts, ti, x, kp, w, b, 0.06, packed=ops.pack_bf16x3(w))
    assert float((o32 - obf).abs().max()) <= 2e-5 * max(1.0, float(o32.abs().max()))
