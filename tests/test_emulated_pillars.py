"""CPU: the PointPillars building blocks (pillars.hip + gemm.hip) through the host emulator vs the oracle's
PyTorch restatement of the reference ops (oracle/pointpillars_ref.py) — float tolerance 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emu
import synth_data
from oracle import pointpillars_ref as P

pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")
TOL = 1e-4


def _fold_pfn(sd, i, eps=1e-3):
    p = "voxel_encoder.pfn_layers.%d" % i
    s = (sd[p + ".norm.weight"].double() / torch.sqrt(sd[p + ".norm.running_var"].double() + eps))
    t = sd[p + ".norm.bias"].double() - sd[p + ".norm.running_mean"].double() * s
    wt = (sd[p + ".linear.weight"].double() * s[:, None]).t().contiguous()
    return wt.float().numpy(), t.float().numpy()


@pytest.mark.parametrize("cfg_name,frames", [("SMALL_CFG", [5, 6]), ("SMALL_ONE", [7])])
def test_pillar_features_fused_voxel_gather_pfn_scatter(cfg_name, frames):
    cfg = dict(P.SMALL_CFG)
    if cfg_name == "SMALL_ONE":       # single PFN layer, 4-channel points (the KITTI shape family)
        cfg["voxel_encoder"] = dict(in_channels=4, feat_channels=[64], voxel_size=[0.4, 0.4, 4])
    sd = P.make_state_dict(cfg, 9)
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(f), cfg) for f in frames]
    (_, _, _), aux = P.forward(sd, cfg, [torch.from_numpy(c) for c in clouds])
    ref = P.scatter(cfg, aux["pillar_features"], aux["coors"], len(clouds)).permute(0, 2, 3, 1).numpy()   # NHWC
    pts = np.concatenate(clouds)
    rs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])])
    vz, pcr = cfg["voxelize"], cfg["point_cloud_range"]
    vox = emu.voxelize(pts, rs, vz["voxel_size"], pcr[:3], pcr[3:], vz["max_num_points"], vz["max_voxels"][1])
    nl = len(cfg["voxel_encoder"]["feat_channels"])
    layers = [_fold_pfn(sd, i) for i in range(nl)]
    vx, vy = vz["voxel_size"][:2]
    ny, nx = cfg["scatter"]["output_shape"]
    rc, canvas = emu.pillar_features(pts, vox, cfg["voxel_encoder"]["in_channels"], vz["max_num_points"], vx, vy,
                                     vx / 2 + pcr[0], vy / 2 + pcr[1], nx, ny, layers, len(clouds))
    assert rc == 0
    assert np.abs(canvas - ref).max() <= TOL
    assert (canvas != 0).any(-1).sum() == len(aux["coors"])          # exactly the in-bounds pillars were written


@pytest.mark.parametrize("cin,cout,stride,hw", [(64, 64, 2, (20, 28)), (32, 48, 1, (9, 13)), (128, 64, 2, (16, 16))])
def test_conv3x3_bn_relu_matches_torch(cin, cout, stride, hw):
    rng = np.random.default_rng(cin + cout)
    x = rng.standard_normal((2, cin) + hw).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * (1.0 / np.sqrt(9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=1))
    wk = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(9 * cin, cout))          # [(ky,kx,ci), co]
    rc, out = emu.conv2d_nhwc(x.transpose(0, 2, 3, 1), wk, b, stride, 1)
    assert rc == 0 and np.abs(out - ref.permute(0, 2, 3, 1).numpy()).max() <= TOL


@pytest.mark.parametrize("stride", [1, 2, 4])
def test_deconv_pixel_shuffle_into_concat_slice(stride):
    rng = np.random.default_rng(stride)
    cin, cout = 64, 32
    x = rng.standard_normal((2, cin, 6, 5)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, stride, stride)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.relu(F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride))
    wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(cin, stride * stride * cout))   # [ci, (dy,dx,co)]
    big = np.full((2, 6 * stride, 5 * stride, 80), -1.0, np.float32)
    rc, out = emu.deconv2d_nhwc(x.transpose(0, 2, 3, 1), wk, b, stride, cout, out=big, ch_off=40)
    assert rc == 0
    assert np.abs(out[..., 40:72] - ref.permute(0, 2, 3, 1).numpy()).max() <= TOL
    assert (out[..., :40] == -1).all() and (out[..., 72:] == -1).all()


def test_nhwc_to_nchw_slices():
    x = np.random.default_rng(0).standard_normal((2, 7, 9, 72)).astype(np.float32)
    assert np.array_equal(emu.nhwc_to_nchw(x, 18, 42), x[..., 18:60].transpose(0, 3, 1, 2))


def _run_in_subprocess_with_big_gemm(code, **extra_env):
    """The register-blocked 128-row GEMM only takes problems that fill the chip; ML3D_GEMM_BIG_MIN_TILES (read once when
    the library is first used) lowers the bar, so the check runs in a fresh interpreter."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    env = dict(os.environ, ML3D_GEMM_BIG_MIN_TILES="1",
               PYTHONPATH=os.pathsep.join([here, root, os.path.join(root, "open3d-ml_amd")]), **extra_env)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "OK" in r.stdout
    return r.stderr


def test_register_blocked_gemm_kernel_conv_deconv_linear():
    """gemm_tile2 (128 x {64, 128} tiles, 2 x 2 / 2 x 1 MFMA blocks, tap-mask conv loader): 3x3 convs with stride 1 / 2
    at image borders, a ragged last row tile, the pixel-shuffle (transposed conv) store, and dense rows with a
    concatenated second operand + residual -- against torch."""
    _run_in_subprocess_with_big_gemm(r'''
import numpy as np, torch, torch.nn.functional as F
import emu
rng = np.random.default_rng(0)
for cin, cout, stride, hw in [(32, 64, 1, (13, 11)), (64, 128, 2, (12, 14)), (32, 192, 1, (9, 16)), (64, 64, 1, (10, 13))]:
    x = rng.standard_normal((2, cin) + hw).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=1))
    wk = w.transpose(2, 3, 1, 0).reshape(9 * cin, cout)
    rc, out = emu.conv2d_nhwc(x.transpose(0, 2, 3, 1), wk, b, stride, 1)
    assert rc == 0
    assert np.abs(out.transpose(0, 3, 1, 2) - ref.numpy()).max() < 2e-4, (cin, cout, stride)
# transposed conv, kernel == stride == 2: N = 4 * 32 = 128 columns, pixel-shuffle store into a wider map
cin, cout, s = 32, 32, 2
x = rng.standard_normal((1, cin, 12, 11)).astype(np.float32)
w = (rng.standard_normal((cin, cout, s, s)) * 0.1).astype(np.float32)
b = rng.standard_normal(cout).astype(np.float32)
ref = F.relu(F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s)).numpy()
wk = w.transpose(0, 2, 3, 1).reshape(cin, s * s * cout)
big = np.zeros((1, 24, 22, 72), np.float32)
rc, out = emu.deconv2d_nhwc(x.transpose(0, 2, 3, 1), wk, b, s, cout, out=big, ch_off=40)
assert rc == 0 and np.abs(out[..., 40:72].transpose(0, 3, 1, 2) - ref).max() < 2e-4
# dense rows: [a | a2] @ W + bias + residual, leaky relu; 300 rows (ragged last tile), 64 + 32 -> 64
a = rng.standard_normal((300, 64)).astype(np.float32); a2 = rng.standard_normal((300, 32)).astype(np.float32)
wt = (rng.standard_normal((96, 64)) * 0.1).astype(np.float32); bias = rng.standard_normal(64).astype(np.float32)
res = rng.standard_normal((300, 64)).astype(np.float32)
rc, out = emu.linear(a, wt, bias, a2=a2, residual=res, act=1, slope=0.1)
want = np.concatenate([a, a2], 1) @ wt + bias + res
want = np.where(want > 0, want, 0.1 * want)
assert rc == 0 and np.abs(out - want).max() < 2e-4
# a head-like GEMM: N = 72 is not a multiple of the 64 / 128 column tiles (the last tile is ragged)
a = rng.standard_normal((200, 64)).astype(np.float32)
wt = (rng.standard_normal((64, 72)) * 0.1).astype(np.float32); bias = rng.standard_normal(72).astype(np.float32)
rc, out = emu.linear(a, wt, bias)
assert rc == 0 and np.abs(out - (a @ wt + bias)).max() < 2e-4
print("OK")
''')


# ---- the same convolution on the bf16 matrix pipe (gemm_tile_bf3: three-way bf16 split of both operands) -----------------------
def _bf16_planes(packed, K, N):
    """decode ml3d_gemm_pack_bf16x3's layout [K / 32][3][Npad][32] bf16 -> three float64 [K, N] matrices"""
    npad = (N + 127) // 128 * 128
    u = np.frombuffer(packed.tobytes(), np.uint16)[: 3 * npad * K].reshape(K // 32, 3, npad, 32)
    f = (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return [f[:, p].transpose(0, 2, 1).reshape(K, npad)[:, :N] for p in range(3)], f[:, :, N:, :]


def test_bf16x3_weight_split_is_exact():
    """h + m + l == w for every float (normal range), each plane is bf16 and the padding columns are zero"""
    rng = np.random.default_rng(3)
    K, N = 96, 70
    w = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-20, 20, (K, N)))).astype(np.float32)
    w[0, :4] = [0.0, -0.0, 1.0, -3.0e38]
    rc, packed = emu.pack_bf16x3(w)
    assert rc == 0
    (h, m, l), pad = _bf16_planes(packed, K, N)
    assert np.array_equal(h + m + l, w.astype(np.float64))
    assert np.all(np.abs(m) <= np.abs(h) * 2.0 ** -8 + 1e-300) and np.all(np.abs(l) <= np.abs(h) * 2.0 ** -16 + 1e-300)
    assert not pad.any()


@pytest.mark.parametrize("cin,cout,stride,hw,act", [
    (32, 64, 1, (13, 11), 2),        # 64-column tiles, ragged last row tile
    (64, 128, 2, (12, 14), 2),       # 128-column tiles, M = 84 < one row tile
    (64, 192, 2, (12, 14), 0),       # two column tiles, the second half empty; no activation
    (32, 20, 1, (5, 5), 1),          # N < 64 and not a multiple of 4; leaky relu
    (96, 132, 2, (31, 9), 2),        # cin = 3 chunks per tap, N = 128 + 4
    (64, 64, 1, (16, 16), 2)])       # whole tiles only
def test_bf16x3_conv_matches_float64(cin, cout, stride, hw, act):
    """error against a float64 convolution: of the order of the f32 MFMA kernel's own (both far inside the 1e-4 bar)"""
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((2, cin) + hw).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=1)
    ref = {0: ref, 1: F.leaky_relu(ref, 0.0), 2: F.relu(ref)}[act].permute(0, 2, 3, 1).numpy()
    wk = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(9 * cin, cout))
    rc, out = emu.conv2d_nhwc_bf16x3(x.transpose(0, 2, 3, 1), wk, b, stride, 1, act=act)
    rc32, out32 = emu.conv2d_nhwc(x.transpose(0, 2, 3, 1), wk, b, stride, 1, act=act)
    assert rc == 0 and rc32 == 0
    e, e32 = np.abs(out - ref).max(), np.abs(out32 - ref).max()
    assert e <= 1e-5 and e <= 4 * e32 + 1e-6, (e, e32)


def test_bf16x3_conv_1x1_no_bias_into_a_concat_slice():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1, 9, 7, 64)).astype(np.float32)
    w = (rng.standard_normal((64, 48)) * 0.2).astype(np.float32)
    big = np.full((1, 9, 7, 100), 7.0, np.float32)
    rc, out = emu.conv2d_nhwc_bf16x3(x, w, None, 1, 0, act=0, kh=1, kw=1, out=big, ch_off=40)
    assert rc == 0
    ref = x.astype(np.float64).reshape(-1, 64) @ w.astype(np.float64)
    assert np.abs(out[..., 40:88].reshape(-1, 48) - ref).max() <= 1e-5
    assert (out[..., :40] == 7.0).all() and (out[..., 88:] == 7.0).all()


def test_bf16x3_rejects_what_it_cannot_run():
    L = emu.lib()
    assert L.ml3d_gemm_pack_bf16x3_bytes(40, 64) == 0 and L.ml3d_gemm_pack_bf16x3_bytes(64, 0) == 0
    assert L.ml3d_gemm_pack_bf16x3_bytes(64, 70) == 3 * 128 * 64 * 2
    w = np.zeros((40, 64), np.float32)
    buf = np.zeros(1 << 16, np.uint8)
    assert L.ml3d_gemm_pack_bf16x3(w.ctypes.data, 40, 64, buf.ctypes.data, buf.nbytes, None) == -4       # K % 32
    w = np.zeros((64, 64), np.float32)
    assert L.ml3d_gemm_pack_bf16x3(w.ctypes.data, 64, 64, buf.ctypes.data, 100, None) == -2              # packed buffer too small
    assert L.ml3d_gemm_pack_bf16x3(None, 64, 64, buf.ctypes.data, buf.nbytes, None) == -1
    x = np.zeros((1, 4, 4, 48), np.float32)
    out = np.zeros((1, 4, 4, 64), np.float32)
    rc = L.ml3d_conv2d_nhwc_bf16x3(x.ctypes.data, 1, 4, 4, 48, buf.ctypes.data, None, 3, 3, 1, 1, 2, 0.0, 64, out.ctypes.data, 64, None)
    assert rc == -4                                                                                       # cin % 32
    x = np.zeros((1, 4, 4, 64), np.float32)
    assert L.ml3d_conv2d_nhwc_bf16x3(x.ctypes.data, 1, 4, 4, 64, None, None, 3, 3, 1, 1, 2, 0.0, 64, out.ctypes.data, 64, None) == -1
    assert L.ml3d_conv2d_nhwc_bf16x3(x.ctypes.data, 1, 4, 4, 64, buf.ctypes.data, None, 3, 3, 1, 1, 2, 0.0, 64, out.ctypes.data, 32, None) == -1


@pytest.mark.parametrize("stride", [1, 2, 4])
def test_bf16x3_deconv_pixel_shuffle_into_concat_slice(stride):
    """SECONDFPN's transposed convolution (kernel == stride) as a dense-row GEMM on the bf16x3 path + pixel-shuffle store"""
    rng = np.random.default_rng(stride)
    cin, cout = 64, 32
    x = rng.standard_normal((2, cin, 6, 5)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, stride, stride)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.relu(F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride))
    wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(cin, stride * stride * cout))   # [ci, (dy,dx,co)]
    big = np.full((2, 6 * stride, 5 * stride, 80), -1.0, np.float32)
    rc, out = emu.deconv2d_nhwc_bf16x3(x.transpose(0, 2, 3, 1), wk, b, stride, cout, out=big, ch_off=40)
    assert rc == 0
    assert np.abs(out[..., 40:72] - ref.permute(0, 2, 3, 1).numpy()).max() <= 1e-5
    assert (out[..., :40] == -1).all() and (out[..., 72:] == -1).all()


@pytest.mark.parametrize("m,k,n,act", [(300, 384, 72, 0), (129, 32, 20, 2), (64, 96, 200, 1), (0, 64, 8, 0)])
def test_bf16x3_linear_matches_float64(m, k, n, act):
    """the head Linear's shape (K = 384, N = 72) and ragged ones; zero rows is a no-op"""
    rng = np.random.default_rng(m + k + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.1).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    rc, out = emu.linear_bf16x3(a, w, b, act=act)
    assert rc == 0 and out.shape == (m, n)
    ref = a.astype(np.float64) @ w.astype(np.float64) + b
    ref = {0: ref, 1: np.where(ref > 0, ref, 0.0), 2: np.maximum(ref, 0.0)}[act]
    if m:
        assert np.abs(out - ref).max() <= 1e-5
    # rows that are not float4-addressable / K not a multiple of 32 are refused, not mis-computed
    L = emu.lib()
    buf = np.zeros(1 << 16, np.uint8)
    a2 = np.zeros((4, 64), np.float32)
    o2 = np.zeros((4, 8), np.float32)
    call = lambda lda, k1, ldc: L.ml3d_linear_bf16x3(a2.ctypes.data, lda, k1, None, 0, 0, 4, buf.ctypes.data, None, None, 0, 8, 0, 0.0,
                                                     o2.ctypes.data, ldc, None, 0, None)
    assert call(66, 64, 8) == -4 and call(64, 48, 8) == -4 and call(64, 64, 4) == -1


@pytest.mark.parametrize("m,k1,k2,n,act,res", [(300, 64, 128, 256, 1, False), (129, 32, 64, 128, 2, True), (2000, 96, 0, 40, 0, True),
                                               (40, 512, 512, 64, 1, False)])
def test_bf16x3_linear_two_blocks_residual_split_k(m, k1, k2, n, act, res):
    """[a | a2] . W + bias + residual: KPFCNN's "unary2 + shortcut" GEMM over concatenated inputs; the last case (40 rows, K = 1024) is
    cut along K into 8 slices + gemm_reduce"""
    rng = np.random.default_rng(m + k1 + n)
    a = rng.standard_normal((m, k1)).astype(np.float32)
    a2 = rng.standard_normal((m, k2)).astype(np.float32) if k2 else None
    w = (rng.standard_normal((k1 + k2, n)) * 0.1).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32) if res else None
    rc, out = emu.linear_bf16x3(a, w, b, act=act, a2=a2, residual=r)
    assert rc == 0
    x = a if a2 is None else np.concatenate([a, a2], 1)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b + (r if res else 0.0)
    ref = {0: ref, 1: np.where(ref > 0, ref, 0.0), 2: np.maximum(ref, 0.0)}[act]
    assert np.abs(out - ref).max() <= 2e-5


@pytest.mark.parametrize("m,mc,k,n,act", [(700, 200, 256, 128, 1), (130, 40, 64, 40, 0), (60, 17, 1024, 64, 1)])
def test_bf16x3_linear_with_a_gathered_residual(m, mc, k, n, act):
    """ml3d_linear_bf16x3_gathered (round 6): act(a . W + bias + residual[g[m, 0]]) -- KPFCNN's decoder step split by linearity on the
    bf16 pipe: the gathered residual in the 128-row epilogue (shadow rows >= residual rows and negative rows add nothing), and, for
    the 60-row / K = 1024 case, in gemm_reduce behind the split-K slices."""
    rng = np.random.default_rng(m + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.1).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((mc, n)).astype(np.float32)
    g = rng.integers(-1, mc + 2, (m, 3)).astype(np.int32)             # first column: the residual row (mc, mc + 1, -1: none)
    rc, out = emu.linear_bf16x3(a, w, b, act=act, residual=r, residual_gather=g)
    assert rc == 0
    rr = np.where(((g[:, 0] >= 0) & (g[:, 0] < mc))[:, None], r[np.clip(g[:, 0], 0, mc - 1)], 0.0)
    ref = a.astype(np.float64) @ w.astype(np.float64) + b + rr
    ref = {0: ref, 1: np.where(ref > 0, ref, 0.0)}[act]
    assert np.abs(out - ref).max() <= 2e-5


def test_bf16x3_stride1_convolutions_through_both_kernels():
    """3 x 3 / stride 1 / pad 1 takes conv3x3s1_bf3 (the input window staged once per 16-channel chunk, border taps read a zero pixel);
    ML3D_CONV_WINDOW=0 (a test hook of the emulator build) sends the same problems through gemm_tile_bf3.  Shapes: image rows shorter
    and longer than a 128-pixel tile, a tile spanning two images of the batch, one-pixel-wide and one-pixel-high maps, N in one and two
    column tiles."""
    code = r'''
import numpy as np, torch, torch.nn.functional as F
import emu
rng = np.random.default_rng(1)
for cin, cout, hw, nb in [(32, 64, (13, 11), 2), (64, 128, (3, 140), 1), (32, 200, (9, 16), 3), (64, 40, (1, 37), 2), (32, 64, (45, 1), 2),
                          (96, 64, (16, 16), 1)]:
    x = rng.standard_normal((nb, cin) + hw).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=1, padding=1))
    wk = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(9 * cin, cout))
    rc, out = emu.conv2d_nhwc_bf16x3(x.transpose(0, 2, 3, 1), wk, b, 1, 1)
    assert rc == 0
    assert np.abs(out - ref.permute(0, 2, 3, 1).numpy()).max() <= 1e-5, (cin, cout, hw)
print("OK")
'''
    for window in ("1", "0"):
        _run_in_subprocess_with_big_gemm(code, ML3D_CONV_WINDOW=window)
