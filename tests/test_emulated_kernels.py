"""CPU: run the SAME .hip sources the GPU runs through the host emulator (tests/hipemu) and compare
with the oracle: neighbour indices bit-exact, logits within 1e-4.  This is a logic check of the
kernels; the authoritative parity tests are the `-m gpu` ones."""
import os

import numpy as np
import pytest

import emu
import synth_data
from oracle import ops as oops
from oracle import randlanet_ref as R

pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")


@pytest.mark.parametrize("n,kind", [(3000, "vol"), (5000, "surf"), (100, "vol"), (17, "vol"), (2000, "dup"), (1, "vol")])
def test_knn_self_query(n, kind):
    rng = np.random.default_rng(n)
    if kind == "vol":
        p = rng.random((n, 3), dtype=np.float32) * 10
    elif kind == "surf":
        p = rng.random((n, 3), dtype=np.float32) * np.array([30, 30, 0.05], np.float32)
    else:
        p = np.repeat(rng.random((n // 4, 3), dtype=np.float32), 4, 0)
    idx, d2 = emu.knn(p, [0, n], k=16)
    ref, rd = oops.knn_search(p, p, 16, return_distances=True)
    kk = ref.shape[1]
    assert np.array_equal(idx[:, :kk], ref) and np.array_equal(d2[:, :kk], rd)
    assert (idx[:, kk:] == -1).all()


def test_knn_external_queries_and_other_k():
    rng = np.random.default_rng(5)
    p = rng.random((4000, 3), dtype=np.float32) * 10
    q = rng.random((777, 3), dtype=np.float32) * 14 - 2        # some queries outside the support bbox
    for k in (1, 5, 8, 20, 33):
        idx, _ = emu.knn(p, [0, 4000], q, [0, 777], k=k)
        assert np.array_equal(idx, oops.knn_search(p, q, k))


def test_knn_batched_row_splits_with_empty_item():
    p = np.random.default_rng(6).random((4000, 3), dtype=np.float32)
    ps = [0, 1000, 1000, 2500, 4000]
    idx, _ = emu.knn(p, ps, k=16)
    ref, _ = oops.knn_search_batched(p, ps, p, ps, 16)
    assert np.array_equal(idx, ref)
    loc, _ = emu.knn(p, ps, k=1, local=True)
    assert np.array_equal(loc[:, 0], np.concatenate([np.arange(1000), np.arange(1500), np.arange(1500)]))


def test_degenerate_clouds():
    same = np.ones((50, 3), np.float32) * 3.5                  # zero extent
    idx, _ = emu.knn(same, [0, 50], k=4)
    assert np.array_equal(idx, oops.knn_search(same, same, 4))
    line = np.zeros((500, 3), np.float32); line[:, 0] = np.linspace(0, 1, 500)
    idx, _ = emu.knn(line, [0, 500], k=16)
    assert np.array_equal(idx, oops.knn_search(line, line, 16))
    far = np.random.default_rng(1).random((2000, 3), dtype=np.float32); far[0] = [1e4, -1e4, 5e3]   # outlier
    idx, _ = emu.knn(far, [0, 2000], k=16)
    assert np.array_equal(idx, oops.knn_search(far, far, 16))


def test_randla_pyramid_on_lidar_patches():
    pts = np.stack([synth_data.semantickitti_patch(i, 4096) for i in range(2)])
    nbr, itp = emu.pyramid(pts, [4, 4, 4, 4])
    for b in range(2):
        pc = pts[b]
        for l in range(4):
            assert np.array_equal(nbr[l][b], oops.knn_search(pc, pc, 16))
            sub = pc[:pc.shape[0] // 4]
            assert np.array_equal(itp[l][b], oops.knn_search(sub, pc, 1))
            pc = sub


CFGS = [
    dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
         dim_features=8, dim_output=[16, 64, 128, 256]),
    dict(num_neighbors=16, num_layers=2, num_classes=5, sub_sampling_ratio=[4, 2], in_channels=6,
         dim_features=8, dim_output=[8, 32]),
    # level sizes 1100 / 275 / 68 / 17 per cloud: tiles of the attention kernels (16, 2 and 4 points) straddle clouds
    # (in_channels 6 -- xyz + colours, the S3DIS / Semantic3D configs -- through the fused fc0 + mlp1 head kernel)
    dict(num_neighbors=16, num_layers=3, num_classes=7, sub_sampling_ratio=[4, 4, 4], in_channels=6,
         dim_features=8, dim_output=[16, 64, 128]),
    # 16 features into the 16-wide first layer: NOT the shape of the fused head / attention epilogues (8 features, every reference
    # config) -> fc0, mlp1, pool1.mlp and the pool2 | mlp2 | shortcut chain as separate launches around lfa_attn_mfma16
    dict(num_neighbors=16, num_layers=2, num_classes=6, sub_sampling_ratio=[4, 4], in_channels=4,
         dim_features=16, dim_output=[16, 64]),
]


@pytest.mark.parametrize("ci,B,N", [(0, 2, 1024), (1, 3, 515), (2, 3, 1100), (3, 2, 600)])
def test_randla_forward_matches_oracle(ci, B, N):
    cfg = CFGS[ci]
    rng = np.random.default_rng(3)
    pts = synth_data.uniform_cloud(3, B * N).reshape(B, N, 3)
    feats = pts.copy() if cfg["in_channels"] == 3 else np.concatenate(
        [pts, rng.random((B, N, cfg["in_channels"] - 3), dtype=np.float32)], 2)
    sd = R.make_state_dict(cfg, 11)
    inp = R.build_inputs(pts, feats, cfg, oops.knn_search)
    ref = R.forward(sd, cfg, inp).numpy()
    nbr = [np.ascontiguousarray(x.numpy().astype(np.int32)) for x in inp["neighbor_indices"]]
    itp = [np.ascontiguousarray(x.numpy().astype(np.int32)) for x in inp["interp_idx"]]
    rc, out = emu.randla_forward(cfg, sd, pts, feats, nbr, itp)
    assert rc == 0
    assert np.abs(out - ref).max() <= 1e-4


_VARIANT_CODE = r'''
import sys
import numpy as np
import emu, synth_data
from oracle import ops as oops
from oracle import randlanet_ref as R
import test_emulated_kernels as T
ci, B, N, seed_pts, seed_w = (int(v) for v in sys.argv[1:6])
cfg = T.CFGS[ci]
pts = synth_data.uniform_cloud(seed_pts, B * N).reshape(B, N, 3)
feats = pts.copy() if cfg["in_channels"] == 3 else np.concatenate(
    [pts, np.random.default_rng(seed_pts).random((B, N, cfg["in_channels"] - 3), dtype=np.float32)], 2)
sd = R.make_state_dict(cfg, seed_w)
inp = R.build_inputs(pts, feats, cfg, oops.knn_search)
ref = R.forward(sd, cfg, inp).numpy()
nbr = [np.ascontiguousarray(x.numpy().astype(np.int32)) for x in inp["neighbor_indices"]]
itp = [np.ascontiguousarray(x.numpy().astype(np.int32)) for x in inp["interp_idx"]]
rc, out = emu.randla_forward(cfg, sd, pts, feats, nbr, itp)
assert rc == 0, rc
err = float(np.abs(out - ref).max())
assert err <= 1e-4, err
print("VARIANT_OK", err)
'''


def _run_variant(knobs, ci, B, N, seed_pts, seed_w):
    """The library reads its ML3D_* A/B switches once per process (randla.hip: knobs()), so every setting gets its own
    interpreter."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    env = dict(os.environ, **knobs)
    env["PYTHONPATH"] = os.pathsep.join([here, root, os.path.join(root, "open3d-ml_amd")])
    r = subprocess.run([sys.executable, "-c", _VARIANT_CODE, str(ci), str(B), str(N), str(seed_pts), str(seed_w)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("ci,B,N", [(0, 2, 1024), (2, 3, 1100)])
def test_randla_forward_fused_linear_chains(ci, B, N):
    """pool2.mlp + (mlp2 | shortcut) as one 2-layer chain launch, decoder-last + fc1 as one 4-layer chain: these per-wave kernels
    take levels of >= 64 k rows; ML3D_RANDLA_FUSE_ROWS=1 (the library's one test hook, read once per process) lets the small
    levels the emulator can afford reach them.  The default (per-layer launches at this size) runs in every other test here; the
    generic VALU kernels (lfa_stage, linear_act: the fallback for widths without an MFMA kernel) in the dim_output [8, 32] config."""
    _run_variant({"ML3D_RANDLA_FUSE_ROWS": "1"}, ci, B, N, 4, 12)


def test_tile_order_is_a_cloud_major_spatial_permutation():
    pts = np.stack([synth_data.semantickitti_patch(i, 2048) for i in range(3)])
    nbr, itp, order = emu.pyramid_ordered(pts, [4, 4])
    n = [2048, 512]
    for l in range(2):
        o = order[l].reshape(3, n[l])
        for b in range(3):
            assert np.array_equal(np.sort(o[b]), np.arange(b * n[l], (b + 1) * n[l]))      # a permutation inside each cloud
        # spatial: consecutive points of the order are much closer than consecutive points of the (random) row order
        p = pts[0, :n[l]]
        d_ord = np.linalg.norm(np.diff(p[o[0]], axis=0), axis=1).mean()
        d_row = np.linalg.norm(np.diff(p, axis=0), axis=1).mean()
        assert d_ord < 0.5 * d_row
    a, b = emu.pyramid(pts, [4, 4])
    assert all(np.array_equal(x, y) for x, y in zip(nbr + itp, a + b))                   # the searches are unaffected


@pytest.mark.parametrize("ci,B,N,kind", [(0, 2, 1024, "grid"), (2, 3, 1100, "grid"), (2, 3, 1100, "reversed"),
                                         (0, 2, 1024, "random")])
def test_randla_forward_with_a_tile_order_is_bit_identical(ci, B, N, kind):
    """The tile order only changes which points share a tile: per-point arithmetic, hence every logit, is unchanged --
    for the grid's order and for ANY cloud-major permutation (incl. tiles straddling clouds, cfg 2)."""
    cfg = CFGS[ci]
    pts = synth_data.uniform_cloud(21, B * N).reshape(B, N, 3)
    sd = R.make_state_dict(cfg, 17)
    ratios = cfg["sub_sampling_ratio"]
    nbr, itp, order = emu.pyramid_ordered(pts, ratios)
    if kind != "grid":
        rng = np.random.default_rng(4)
        n = N
        order = []
        for r in ratios:
            o = np.arange(B * n, dtype=np.int32).reshape(B, n)
            o = o[:, ::-1] if kind == "reversed" else np.stack([rng.permutation(row) for row in o])
            order.append(np.ascontiguousarray(o.reshape(-1)))
            n //= r
    feats = pts.copy() if cfg["in_channels"] == 3 else np.concatenate(
        [pts, np.random.default_rng(2).random((B, N, cfg["in_channels"] - 3), dtype=np.float32)], 2)
    rc0, base = emu.randla_forward(cfg, sd, pts, feats, nbr, itp)
    rc1, out = emu.randla_forward(cfg, sd, pts, feats, nbr, itp, order=order)
    assert rc0 == 0 and rc1 == 0
    assert np.array_equal(out, base)


def test_randla_forward_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "randlanet_small.npz"))
    cfg = dict(num_neighbors=16, num_layers=3, num_classes=8, sub_sampling_ratio=[4, 4, 2], in_channels=6,
               dim_features=8, dim_output=[16, 32, 64])
    sd = R.make_state_dict(cfg, int(g["weights_seed"]))
    nbr, itp = emu.pyramid(g["points"], cfg["sub_sampling_ratio"])
    assert np.array_equal(nbr[0], g["nbr0"]) and np.array_equal(itp[0], g["interp0"])
    rc, out = emu.randla_forward(cfg, sd, g["points"], g["features"], nbr, itp)
    assert rc == 0 and np.abs(out - g["logits"]).max() <= 1e-4


def test_forward_rejects_levels_with_fewer_than_16_points():
    cfg = dict(CFGS[0])
    sd = R.make_state_dict(cfg, 1)
    pts = synth_data.uniform_cloud(1, 512).reshape(1, 512, 3)         # level 3 would have 8 < 16 points
    nbr = [np.zeros((1, 512 // 4 ** l, 16), np.int32) for l in range(4)]
    itp = [np.zeros((1, 512 // 4 ** l, 1), np.int32) for l in range(4)]
    rc, _ = emu.randla_forward(cfg, sd, pts, pts, nbr, itp)
    assert rc == -4
