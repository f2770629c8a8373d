"""GPU parity: RandLA-Net inference (GPU neighbour pyramid + fused HIP forward) vs the CPU oracle and
the reference-generated golden vectors.  Tolerance: indices exact, logits max|d| <= 1e-4 (north_star)."""
import os

import numpy as np
import pytest
import torch

import synth_data
from oracle import ops as oops
from oracle import randlanet_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-4

KITTI = dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
             dim_features=8, dim_output=[16, 64, 128, 256])
SMALL = dict(num_neighbors=16, num_layers=3, num_classes=8, sub_sampling_ratio=[4, 4, 2], in_channels=6,
             dim_features=8, dim_output=[16, 32, 64])
FIVE = dict(num_neighbors=16, num_layers=5, num_classes=13, sub_sampling_ratio=[4, 4, 4, 4, 2], in_channels=6,
            dim_features=8, dim_output=[16, 64, 128, 256, 512])


# level sizes 1100 / 275 / 68 / 17 per cloud: attention tiles (16, 2 and 4 points) straddle cloud boundaries
RAGGED = dict(num_neighbors=16, num_layers=3, num_classes=7, sub_sampling_ratio=[4, 4, 4], in_channels=3,
              dim_features=8, dim_output=[16, 64, 128])


def _model(cfg, sd):
    from ml3d.torch.models.randlanet import RandLANet
    assert torch.cuda.is_available()
    m = RandLANet(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    return m.eval()


def _run(cfg, sd, pts, feats):
    m = _model(cfg, sd)
    d = torch.device("cuda:0")
    out = m({"coords": [torch.from_numpy(pts).to(d)], "features": torch.from_numpy(feats).to(d)})
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cfg,B,N,seed", [(KITTI, 2, 4096, 1), (SMALL, 3, 1030, 2), (FIVE, 1, 8192, 3), (RAGGED, 3, 1100, 4)])
def test_forward_matches_oracle(cfg, B, N, seed):
    rng = np.random.default_rng(seed)
    pts = np.stack([synth_data.semantickitti_patch(50 + seed * 10 + b, N) for b in range(B)])
    feats = pts.copy() if cfg["in_channels"] == 3 else np.concatenate(
        [pts, rng.random((B, N, cfg["in_channels"] - 3), dtype=np.float32)], 2)
    sd = R.make_state_dict(cfg, 40 + seed)
    ref = R.forward(sd, cfg, R.build_inputs(pts, feats, cfg, oops.knn_search)).numpy()
    out = _run(cfg, sd, pts, feats)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= TOL


def test_reference_golden_small(golden_dir):
    g = np.load(os.path.join(golden_dir, "randlanet_small.npz"))
    sd = R.make_state_dict(SMALL, int(g["weights_seed"]))
    out = _run(SMALL, sd, g["points"], g["features"])
    assert np.abs(out - g["logits"]).max() <= TOL


def test_reference_golden_kitti4096(golden_dir):
    g = np.load(os.path.join(golden_dir, "randlanet_kitti4096.npz"))
    sd = R.make_state_dict(KITTI, int(g["weights_seed"]))
    out = _run(KITTI, sd, g["points"], g["points"].copy())
    assert np.abs(out - g["logits"]).max() <= TOL
    assert (out.argmax(-1) == g["logits"].argmax(-1)).mean() >= 0.9999


def test_reference_golden_full_frame_45056(golden_dir):
    """BASELINE size: one 45056-point frame against the reference's logits (every 64th point stored)
    and its arg-max labels for every point (the quantity mIoU is computed from)."""
    g = np.load(os.path.join(golden_dir, "randlanet_kitti45056.npz"))
    pts = synth_data.semantickitti_patch(int(g["frame_id"]), 45056)[None]
    assert abs(pts.astype(np.float64).sum() - float(g["points_sum"])) < 1e-6
    sd = R.make_state_dict(KITTI, int(g["weights_seed"]))
    from ml3d import ops
    t = torch.from_numpy(pts).cuda()
    nbr, _ = ops.randla_knn_pyramid(t, KITTI["sub_sampling_ratio"], 16)
    nb0 = nbr[0].cpu().numpy().astype(np.int64)
    assert int((nb0 * (np.arange(16) + 1)).sum()) == int(g["nbr0_checksum"])
    assert np.array_equal(nb0.sum(-1)[0, ::16], g["nbr0_rowsum"])
    out = _run(KITTI, sd, pts, pts.copy())
    assert np.abs(out[:, ::64] - g["logits_every64"]).max() <= TOL
    agree = (out.argmax(-1).astype(np.int8) == g["argmax"]).mean()
    assert agree >= 0.9999, agree


def test_accepts_reference_style_precomputed_indices():
    """The dict RandLANet.transform produces (int64 index lists) is consumed unchanged."""
    B, N = 1, 4096
    pts = synth_data.semantickitti_patch(9, N)[None]
    sd = R.make_state_dict(KITTI, 77)
    inp = R.build_inputs(pts, pts.copy(), KITTI, oops.knn_search)
    ref = R.forward(sd, KITTI, inp).numpy()
    m = _model(KITTI, sd)
    out = m(inp)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL


def test_engine_batch_is_frame_independent_and_deterministic():
    from ml3d.engine import RandLAInferenceEngine
    B, N = 4, 45056
    base = np.stack([synth_data.semantickitti_patch(200 + i, N) for i in range(2)])
    frames = np.concatenate([base, base], 0)
    sd = R.make_state_dict(KITTI, 5)
    eng = RandLAInferenceEngine(dict(KITTI, num_points=N), sd, B, N, "cuda:0")
    t = torch.from_numpy(frames).cuda()
    a = eng.step(t, t.clone()).clone()
    b = eng.step(t, t.clone()).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b)                       # run-to-run deterministic
    assert torch.equal(a[0], a[2]) and torch.equal(a[1], a[3])   # a frame's result does not depend on its batch slot
    assert torch.isfinite(a).all()


def test_pipelined_engine_matches_sequential_engine():
    """Two-stream ping-pong (pyramid of batch i+1 under the forward of batch i) returns the same scores."""
    from ml3d.engine import PipelinedRandLAEngine, RandLAInferenceEngine
    cfg = dict(KITTI, num_points=4096)
    sd = R.make_state_dict(cfg, 21)
    dev = torch.device("cuda:0")
    B, N = 3, 4096
    seq = RandLAInferenceEngine(cfg, sd, B, N, dev)
    pipe = PipelinedRandLAEngine(cfg, sd, B, N, dev)
    batches = [torch.from_numpy(np.stack([synth_data.semantickitti_patch(300 + 10 * j + b, N) for b in range(B)])).to(dev)
               for j in range(5)]
    want = [seq.step(p, p.clone()).clone() for p in batches]
    got = []
    for p in batches:
        out = pipe.submit(p, p.clone())
        torch.cuda.current_stream().wait_stream(pipe.compute)
        got.append(out.clone())          # the engine reuses its score buffer two submits later
    pipe.synchronize()
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_tile_order_engine_is_bit_identical():
    """tile_order=True: walking the attention tiles in the neighbour pyramid's brick order must not change a single logit
    (first hardware run: round 2, green)."""
    from ml3d.engine import RandLAInferenceEngine
    B, N = 3, 45056
    frames = np.stack([synth_data.semantickitti_patch(300 + i, N) for i in range(B)])
    sd = R.make_state_dict(KITTI, 7)
    t = torch.from_numpy(frames).cuda()
    plain = RandLAInferenceEngine(dict(KITTI, num_points=N), sd, B, N, "cuda:0", tile_order=False)
    a = plain.step(t, t.clone()).clone()
    ordered = RandLAInferenceEngine(dict(KITTI, num_points=N), sd, B, N, "cuda:0", tile_order=True)
    b = ordered.step(t, t.clone()).clone()
    torch.cuda.synchronize()
    two = RandLAInferenceEngine(dict(KITTI, num_points=N), sd, B, N, "cuda:0", tile_order=2)     # finest two levels only
    c = two.step(t, t.clone()).clone()
    torch.cuda.synchronize()
    assert two.order[2] is None and two.order[3] is None and torch.equal(a, c)
    for l in range(len(ordered.order)):
        o = ordered.order[l].cpu().numpy().reshape(B, -1)
        n_l = o.shape[1]
        for i in range(B):
            assert np.array_equal(np.sort(o[i]), np.arange(i * n_l, (i + 1) * n_l))
    assert torch.equal(a, b)


def test_bench_configuration_frame_stream_b128_matches_oracle():
    """The object bench.py times, at the bench configuration (bench.py: 128 frames per step since round 5): B = 128 frames of
    45056 points through RandLAFrameStream (pinned-host upload on the copy stream, pyramid on the search stream, forward on the
    compute stream, both ping-pong slots used).  Frames {0, 63, 64, 127} (first, both sides of the middle, last): neighbour /
    interpolation indices exact, logits <= 1e-4 vs the CPU oracle."""
    import bench
    from ml3d.engine import RandLAFrameStream
    cfg = dict(bench.CFG)
    B, N = 128, cfg["num_points"]
    frames = bench.synthetic_batch(0, B, N, 8)
    sd = R.make_state_dict(cfg, 11)
    dev = torch.device("cuda:0")
    stream = RandLAFrameStream(cfg, sd, B, N, dev, overlap=True)
    host = torch.from_numpy(frames).pin_memory()
    check = [0, 63, 64, 127]
    ref_in = {i: R.build_inputs(frames[i:i + 1], frames[i:i + 1].copy(), cfg, oops.knn_search) for i in check}
    ref = {i: R.forward(sd, cfg, ref_in[i]).numpy()[0] for i in check}
    outs = []
    for step in range(3):                      # slots 0, 1, 0: the third submit reuses the first slot's buffers
        sc = stream.submit(host)
        torch.cuda.current_stream().wait_stream(stream.compute_stream)
        outs.append(sc[check].clone())
    stream.synchronize()
    torch.cuda.synchronize()
    for step, got in enumerate(outs):
        for j, i in enumerate(check):
            assert np.abs(got[j].cpu().numpy() - ref[i]).max() <= TOL, "step %d frame %d" % (step, i)
    for slot, eng in enumerate(stream.engine.eng):
        for l in range(cfg["num_layers"]):
            nb = eng.nbr[l][check].cpu().long()
            it = eng.itp[l][check].cpu().long()
            for j, i in enumerate(check):
                assert torch.equal(nb[j], ref_in[i]["neighbor_indices"][l][0]), "slot %d layer %d frame %d" % (slot, l, i)
                assert torch.equal(it[j], ref_in[i]["interp_idx"][l][0]), "slot %d layer %d frame %d" % (slot, l, i)
