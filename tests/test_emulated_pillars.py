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
