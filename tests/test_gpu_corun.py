"""Co-run stress: every LDS-using kernel family of the library, run >= 100 times on one
HIP stream WHILE a second stream loops the bf16x3 kernels (`conv3x3s1_bf3` + `gemm_tile_bf3`: v_mfma_f32_32x32x16_bf16 with LDS
writes + barriers -- the co-runner under which `nmsb_mask`'s per-lane LDS lists returned wrong bits in round 5).  Every
repetition's output is compared BIT FOR BIT with the result of the same call on a quiet GPU.  A kernel that is not stable under
that co-runner fails here, in the driver's own `-m gpu` run.

Victims: `radius_gather` (LDS row sort, radius.hip), `rs_hist / rs_scatter` and (round 6) the fused `fs_pass` / `vox_group32` hand-off through voxelize and the batch grid subsample
(sort.hip / voxel.hip), `kp_agg_gemm32` (kpconv.hip), `lfa_attn_mfma16 / _wave / _pf` through the RandLA-Net forward
(randla.hip), `knn_query_multi` (knn.hip), `topk_*` (nms.hip), and the whole decode tail (`nmsb_*`)."""
import threading

import numpy as np
import pytest
import torch

import synth_data

pytestmark = pytest.mark.gpu

REPS = 300


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


class _Bf16x3CoRunner:
    """A host thread that keeps a side stream full of bf16x3 launches: SECOND's 3 x 3 64 -> 64 convolution on 8 sweeps' canvas
    size (conv3x3s1_bf3, 62.5 KB of LDS per workgroup) and a dense-row product (gemm_tile_bf3)."""

    def __init__(self):
        from ml3d import ops
        dev = _dev()
        g = torch.Generator().manual_seed(1)
        self.ops = ops
        self.x = torch.randn((4, 248, 216, 64), generator=g).to(dev)
        self.w = (torch.randn((9 * 64, 64), generator=g) * 0.05).to(dev)
        self.b = torch.zeros(64, device=dev)
        self.pk = ops.pack_bf16x3(self.w)
        assert self.pk is not None
        self.a = torch.randn((65536, 128), generator=g).to(dev)
        self.w2 = (torch.randn((128, 128), generator=g) * 0.05).to(dev)
        self.pk2 = ops.pack_bf16x3(self.w2)
        self.stream = torch.cuda.Stream()
        self.stop = False
        self.launches = 0
        self.error = None
        self.out = torch.empty((4, 248, 216, 64), device=dev)
        with torch.cuda.stream(self.stream):
            self.want_conv = ops.conv2d_nhwc(self.x, self.w, self.b, 3, 3, 1, 1, packed=self.pk).clone()
            self.want_lin = ops.linear_bf16x3(self.a, self.pk2, 128)
            assert self.want_lin is not None
            self.want_lin = self.want_lin.clone()
        self.stream.synchronize()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        try:
            with torch.cuda.device(_dev()), torch.cuda.stream(self.stream):
                while not self.stop:
                    for _ in range(6):
                        self.ops.conv2d_nhwc(self.x, self.w, self.b, 3, 3, 1, 1, packed=self.pk, out=self.out)
                        self.last_lin = self.ops.linear_bf16x3(self.a, self.pk2, 128)
                        self.launches += 2
                    self.stream.synchronize()
        except Exception as e:       # surfaced by __exit__
            self.error = e

    def __enter__(self):
        self.thread.start()
        while self.launches == 0 and self.error is None:
            pass
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.thread.join()
        self.stream.synchronize()
        if self.error is not None:
            raise self.error
        # the co-runner itself must have stayed correct beside the victims
        assert torch.equal(self.out, self.want_conv), "conv3x3s1_bf3 changed under the co-running victim"
        assert torch.equal(self.last_lin, self.want_lin), "gemm_tile_bf3 changed under the co-running victim"
        return False


def _same(a, b):
    if torch.is_tensor(a):
        return a.shape == b.shape and torch.equal(a, b)
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def _stress(call, reps=REPS, min_corun_launches=50):
    """``call()`` -> tensors; quiet result first, then ``reps`` calls beside the co-runner, each bit-compared."""
    quiet = call()
    torch.cuda.synchronize()
    quiet = torch.utils._pytree.tree_map(lambda t: t.clone() if torch.is_tensor(t) else t, quiet)
    bad = []
    with _Bf16x3CoRunner() as co:
        for i in range(reps):
            got = call()
            torch.cuda.current_stream().synchronize()
            if not _same(got, quiet):
                bad.append(i)
        launches = co.launches
    assert launches >= min_corun_launches, "the co-runner did not run beside the victim (%d launches)" % launches
    assert not bad, "%d of %d repetitions differ from the quiet run (first: %s)" % (len(bad), reps, bad[:8])


def test_radius_gather_under_bf16x3():
    from ml3d import ops
    dev = _dev()
    spheres = [synth_data.toronto3d_sphere(700 + i) for i in range(8)]
    lens = [len(s) for s in spheres]
    p = torch.from_numpy(np.concatenate(spheres)).to(dev)
    _stress(lambda: ops.radius_neighbors_dense(p, p, lens, lens, 0.2))


def test_radius_ragged_under_bf16x3():
    from ml3d import ops
    dev = _dev()
    p = torch.from_numpy(synth_data.toronto3d_sphere(711)).to(dev)

    def call():
        r = ops.fixed_radius_search(p, p, 0.3, return_distances=True)
        return r.neighbors_index, r.neighbors_row_splits, r.neighbors_distance
    _stress(call)


def test_sort_scatter_voxelize_and_subsample_under_bf16x3():
    from ml3d import ops
    dev = _dev()
    cloud = synth_data.kitti_sweep(3000) if hasattr(synth_data, "kitti_sweep") else None
    if cloud is None:
        rng = np.random.default_rng(5)
        cloud = np.concatenate([rng.uniform([0, -39.68, -3], [69.12, 39.68, 1], (60000, 3)), rng.random((60000, 1))], 1)
    pts = torch.from_numpy(np.ascontiguousarray(cloud[:, :3], dtype=np.float32)).to(dev)
    rs = torch.tensor([0, len(pts)], dtype=torch.int64)
    vs = torch.tensor([0.16, 0.16, 4.0])
    lo = torch.tensor([0, -39.68, -3.0])
    hi = torch.tensor([69.12, 39.68, 1.0])

    def vox():
        r = ops.voxelize(pts, rs, vs, lo, hi, 32, 40000)
        return r.voxel_coords, r.voxel_point_indices, r.voxel_point_row_splits, r.voxel_batch_splits
    _stress(vox)
    spheres = [synth_data.toronto3d_sphere(720 + i) for i in range(8)]
    lens = [len(s) for s in spheres]
    p = torch.from_numpy(np.concatenate(spheres)).to(dev)

    def sub():
        q, ql = ops.batch_grid_subsampling(p, lens, 0.16)[:2]
        return q, torch.as_tensor(ql)
    _stress(sub)


def test_kp_agg_gemm32_and_wide_kpconv_under_bf16x3():
    from ml3d import ops
    from oracle import kpconv_ref as K
    dev = _dev()
    rng = np.random.default_rng(32)
    s = synth_data.toronto3d_sphere(730)
    inds = K.batch_neighbors(s, s, [len(s)], [len(s)], 0.2)
    q = torch.from_numpy(s).to(dev)
    ti = torch.from_numpy(inds).to(dev)
    kp = torch.from_numpy(K.synthetic_kernel_points(0.2)).to(dev)
    for cin, cout in ((32, 32), (64, 64), (128, 128)):
        x = torch.from_numpy(rng.standard_normal((len(s), cin)).astype(np.float32)).to(dev)
        w = torch.from_numpy((rng.standard_normal((15 * cin, cout)) * (0.5 / np.sqrt(cin))).astype(np.float32)).to(dev)
        b = torch.zeros(cout, device=dev)
        _stress(lambda: ops.kpconv_rigid(q, q, ti, x, kp, w, b, 0.08, 1, 0.2, 1), reps=REPS if cin == 32 else 100)


def test_knn_pyramid_and_randla_attention_under_bf16x3():
    """`knn_query_multi` + every `lfa_attn_*` kernel (d = 16: mfma16, 64: wave, 128 / 256: pf) + the MFMA GEMMs of the forward."""
    from ml3d.engine import RandLAInferenceEngine
    from oracle import randlanet_ref as R
    cfg = dict(num_neighbors=16, num_layers=4, num_points=45056, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4],
               in_channels=3, dim_features=8, dim_output=[16, 64, 128, 256])
    B, N = 4, 45056
    frames = np.stack([synth_data.semantickitti_patch(740 + i, N) for i in range(B)])
    sd = R.make_state_dict(cfg, 3)
    eng = RandLAInferenceEngine(cfg, sd, B, N, "cuda:0")
    t = torch.from_numpy(frames).cuda()

    def call():
        sc = eng.step(t, t.clone())
        return [sc] + [x for x in eng.nbr] + [x for x in eng.itp]
    _stress(call)


def test_topk_100_of_321408_under_bf16x3():
    from ml3d import ops
    g = torch.Generator().manual_seed(9)
    vals = torch.randn((16, 321408), generator=g).to(_dev())
    _stress(lambda: ops.topk_rows(vals, 100, with_values=True), reps=300)


def test_topk_4096_of_70000_under_bf16x3():
    from ml3d import ops
    g = torch.Generator().manual_seed(10)
    vals2 = torch.randn((4, 70000), generator=g).to(_dev())
    _stress(lambda: ops.topk_rows(vals2, 4096, with_values=True), reps=100)


def test_rotated_nms_under_bf16x3():
    """rotated NMS on overlapping boxes (the kernel round 5's failure was found in)"""
    from ml3d import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    n = 2000
    ctr = torch.rand((n, 2), generator=g) * 40
    wh = torch.rand((n, 2), generator=g) * 3 + 1
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2, torch.rand((n, 1), generator=g) * 3.14], 1).to(dev)
    scores = torch.rand((n,), generator=g).to(dev)
    _stress(lambda: ops.nms(boxes, scores, 0.3), reps=1000)
