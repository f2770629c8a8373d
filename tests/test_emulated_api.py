"""CPU: host-side logic of the model classes that needs the library (vote update with padded patches, the preprocess cache
round trip) executed against the HOST EMULATION of the HIP sources.  Each case runs in its own interpreter because
tests/emu_runtime.py monkeypatches the package's device gates (test infrastructure; the product has no CPU mode)."""
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
import numpy as np, torch
import emu_runtime
emu_runtime.install("ml3d")
'''


def _run(body):
    emu.lib()
    r = subprocess.run([sys.executable, "-c", _PRELUDE % {"root": ROOT} + body], capture_output=True, text=True, timeout=900,
                       cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_update_probs_with_a_padded_patch_follows_numpy_fancy_assignment():
    """a cloud smaller than num_points makes the sampler pad the patch with repeated indices
    (semseg_spatially_regular.py:80-84); randlanet.py:457-462 then assigns with numpy semantics: every row reads its OLD
    value, the last occurrence's write wins."""
    _run(r'''
from ml3d.torch.models import RandLANet
cfg = dict(num_neighbors=16, num_layers=2, num_points=256, num_classes=5, sub_sampling_ratio=[4, 4], in_channels=3,
           dim_features=8, dim_output=[16, 32])
m = RandLANet(**cfg, device="cpu")
rng = np.random.default_rng(0)
n_cloud = 180
inds = np.concatenate([np.arange(n_cloud), rng.integers(0, n_cloud, 256 - n_cloud)])
rng.shuffle(inds)
logits = torch.from_numpy((rng.standard_normal((1, 256, 5)) * 3).astype(np.float32))
probs = rng.random((n_cloud, 5)).astype(np.float16)
out = m.update_probs({'data': {'point_inds': [inds]}}, logits, probs.copy())
ref = probs.copy()
p = torch.softmax(logits[0], -1).numpy()
ref[inds] = 0.95 * ref[inds] + (1 - 0.95) * p
d = np.abs(out.astype(np.float32) - ref.astype(np.float32))
assert out.dtype == np.float16 and d.max() <= 2 ** -10 and (d == 0).mean() > 0.98, (d.max(), (d == 0).mean())
# and the padded-patch sampler bumps a repeated point once (plain fancy +=)
m.possibility = np.zeros(n_cloud)
m.rng = np.random.default_rng(1)
class Tree:
    data = np.zeros((n_cloud, 3), np.float32)
pc = rng.random((n_cloud, 3)).astype(np.float32)
_, idxs, c = m._possibility_sampler(pc, None, None, Tree, 256)
dist = np.sum(np.square((pc[idxs] - c).astype(np.float32)), axis=1)
want = np.zeros(n_cloud); want[idxs] += np.square(1 - dist / np.max(dist))
assert len(idxs) == 256 and np.array_equal(m.possibility, want)
print("ok")
''')


def test_preprocess_result_survives_the_reference_dataloader_cache():
    """ml3d/utils/dataset_helper.py:65-69: the cache writes ``preprocess``'s dict with np.save (pickle) and re-reads it for
    every patch; the search structure must come back usable (device copy rebuilt lazily)."""
    _run(r'''
import tempfile, synth_data
from ml3d.torch.models import KPFCNN
import synth_weights
cfg = dict(synth_weights.TORONTO3D_CFG) if hasattr(synth_weights, "TORONTO3D_CFG") else None
m = KPFCNN(first_subsampling_dl=0.3, in_radius=2.0, num_classes=8, lbl_values=list(range(9)), in_features_dim=1,
           first_features_dim=32, device="cpu", sampler_index="sklearn")
data = synth_data.toronto3d_tile(3, half=2.0, density=0.05)
pre = m.preprocess(data, {'split': 'test'})
f = os.path.join(tempfile.mkdtemp(), "c.npy")
np.save(f, pre)
back = np.load(f, allow_pickle=True).item()
t0, t1 = pre['search_tree'], back['search_tree']
c = pre['point'][5:6]
assert np.array_equal(t0.query(c, k=40)[1], t1.query(c, k=40)[1])
assert np.array_equal(t0.query_radius(c, r=1.0)[0], t1.query_radius(c, r=1.0)[0])
assert np.array_equal(pre['proj_inds'], back['proj_inds']) and t1.data.dtype == np.float32
print("ok")
''')


def test_batched_get_bboxes_matches_the_oracle_loop_formulation():
    """``Anchor3DHead.get_bboxes`` (batched HIP decode + B x C NMS problems, one read-back) vs the oracle's restatement of the
    reference's per-sample / per-class loop (oracle/pointpillars_ref.get_bboxes_single, pinned to the real reference by
    tests/golden/pointpillars_small.npz): the golden's own head maps, then random maps with nms_pre = 300 candidates (five mask
    words per problem), a sample where nothing passes the score threshold, and a map smaller than nms_pre (no top-k)."""
    _run(r'''
from oracle import pointpillars_ref as P
from ml3d.torch.models.point_pillars import PointPillars
import copy
g = np.load(os.path.join(ROOT, "tests", "golden", "pointpillars_small.npz"))
cfg = P.SMALL_CFG
m = PointPillars(device="cpu", **cfg)
cls, reg, dr = (torch.from_numpy(g[k]) for k in ("cls", "reg", "dir"))
boxes, scores, labels = m.bbox_head.get_bboxes(cls, reg, dr)
for i in range(cls.shape[0]):
    assert np.array_equal(labels[i].numpy(), g["labels%d" % i]), i
    assert np.abs(scores[i].numpy() - g["scores%d" % i]).max() <= 1e-6
    assert np.abs(boxes[i].numpy() - g["boxes%d" % i]).max() <= 1e-4
    rb, rs, rl = P.get_bboxes_single(cfg, cls[i], reg[i], dr[i])
    assert np.array_equal(labels[i].numpy(), rl.numpy()) and np.abs(boxes[i].numpy() - rb.numpy()).max() <= 1e-4
# random maps, many survivors
for nms_pre, H, W, thr, bias in ((300, 12, 16, 0.3, 0.0), (64, 4, 4, 0.1, 0.0), (100, 10, 10, 0.5, -9.0)):
    c2 = copy.deepcopy(cfg)
    c2["head"] = dict(c2["head"], nms_pre=nms_pre, score_thr=thr)
    m2 = PointPillars(device="cpu", **c2)
    A, C = m2.bbox_head.num_anchors, len(c2["classes"])
    rng = np.random.default_rng(nms_pre)
    cls = torch.from_numpy((rng.standard_normal((3, A * C, H, W)) * 2 + bias).astype(np.float32))
    cls[2] -= 30.0 if bias == 0.0 else 0.0                     # sample 2: nothing above the threshold
    reg = torch.from_numpy((rng.standard_normal((3, A * 7, H, W)) * 0.3).astype(np.float32))
    dr = torch.from_numpy(rng.standard_normal((3, A * 2, H, W)).astype(np.float32))
    boxes, scores, labels = m2.bbox_head.get_bboxes(cls, reg, dr)
    for i in range(3):
        rb, rs, rl = P.get_bboxes_single(c2, cls[i], reg[i], dr[i])
        assert np.array_equal(labels[i].numpy(), rl.numpy()), (nms_pre, i, len(rl), len(labels[i]))
        if len(rl):
            assert np.abs(scores[i].numpy() - rs.numpy()).max() <= 1e-6
            assert np.abs(boxes[i].numpy() - rb.numpy()).max() <= 1e-4
    assert len(labels[2]) == 0 or bias != 0.0
    # the loop formulation of the product (one ml3d_nms call per class) agrees as well
    sb, ss, sl = m2.bbox_head.get_bboxes_single(cls[0], reg[0], dr[0])
    assert np.array_equal(sl.numpy(), labels[0].numpy()) and (len(sl) == 0 or np.abs(sb.numpy() - boxes[0].numpy()).max() <= 1e-6)
print("ok")
''')


def test_batched_get_bboxes_at_the_large_nms_pre_of_the_lyft_and_waymo_configs():
    """The batched decode at ``nms_pre`` = 1100 and 4096 (pointpillars_{argoverse,lyft,nuscenes}.yml: 1000, pointpillars_waymo.yml:
    4096): 18 and 64 mask words per problem, so ``nmsb_reduce`` walks many 64-row LDS stages and ``ml3d_topk_rows`` sorts 2048 / 4096
    keys -- against the oracle's loop formulation.  Hundreds of detections per sample: two scores one ulp apart (the kernel's
    sigmoid against torch's) may swap places, so the rows are compared in a canonical order."""
    _run(r'''
from oracle import pointpillars_ref as P
from ml3d.torch.models.point_pillars import PointPillars
import copy
cfg = P.SMALL_CFG

def canon(b, s, l):
    b, s, l = np.asarray(b), np.asarray(s), np.asarray(l)
    o = np.lexsort((np.round(b[:, 1], 3), np.round(b[:, 0], 3), -np.round(s, 5), l))
    return b[o], s[o], l[o]

for nms_pre, H, W, thr in ((1100, 24, 26, 0.3), (4096, 40, 56, 0.55)):
    c2 = copy.deepcopy(cfg)
    c2["head"] = dict(c2["head"], nms_pre=nms_pre, score_thr=thr)
    m2 = PointPillars(device="cpu", **c2)
    A, C = m2.bbox_head.num_anchors, len(c2["classes"])
    assert H * W * A > nms_pre
    rng = np.random.default_rng(nms_pre)
    cls = torch.from_numpy((rng.standard_normal((2, A * C, H, W)) * 2).astype(np.float32))
    reg = torch.from_numpy((rng.standard_normal((2, A * 7, H, W)) * 0.3).astype(np.float32))
    dr = torch.from_numpy(rng.standard_normal((2, A * 2, H, W)).astype(np.float32))
    boxes, scores, labels = m2.bbox_head.get_bboxes(cls, reg, dr)
    for i in range(2):
        rb, rs, rl = P.get_bboxes_single(c2, cls[i], reg[i], dr[i])
        assert len(rl) > 300 and len(rl) == len(labels[i]), (nms_pre, i, len(rl), len(labels[i]))
        gb, gs, gl = canon(boxes[i].numpy(), scores[i].numpy(), labels[i].numpy())
        ob, os_, ol = canon(rb.numpy(), rs.numpy(), rl.numpy())
        assert np.array_equal(gl, ol), (nms_pre, i)
        assert np.abs(gs - os_).max() <= 1e-6
        assert (np.abs(gb - ob) / np.maximum(1.0, np.abs(ob))).max() <= 1e-4
print("ok")
''')


def test_batched_nms_walk_at_the_stage_and_word_boundaries():
    """``nmsb_reduce`` stages 64 mask rows at a time and a mask word holds 64 candidates: problems with EXACTLY 63 / 64 / 65 / 127 /
    128 / 129 / 200 participating candidates (class 0: that many anchors above the score threshold, class 1: none), boxes drawn
    so that many overlap -- kept sets and order against the oracle's loop formulation."""
    _run(r'''
from oracle import pointpillars_ref as P
from ml3d.torch.models.point_pillars import PointPillars
import copy
cfg = P.SMALL_CFG
c2 = copy.deepcopy(cfg)
c2["head"] = dict(c2["head"], nms_pre=256, score_thr=0.5)
m2 = PointPillars(device="cpu", **c2)
A, C = m2.bbox_head.num_anchors, len(c2["classes"])
H, W = 10, 12
assert C == 2 and H * W * A > 256
for nv in (63, 64, 65, 127, 128, 129, 200):
    rng = np.random.default_rng(nv)
    cls = np.full((1, A * C, H, W), -8.0, np.float32)
    flat = cls.reshape(1, A, C, H * W)                               # channel = a * C + c
    pick = rng.choice(A * H * W, nv, replace=False)
    flat[0, pick // (H * W), 0, pick % (H * W)] = (0.5 + rng.random(nv) * 4).astype(np.float32)     # distinct scores above 0.5
    cls = torch.from_numpy(cls)
    reg = torch.from_numpy((rng.standard_normal((1, A * 7, H, W)) * 0.15).astype(np.float32))       # small deltas: neighbours overlap
    dr = torch.from_numpy(rng.standard_normal((1, A * 2, H, W)).astype(np.float32))
    boxes, scores, labels = m2.bbox_head.get_bboxes(cls, reg, dr)
    rb, rs, rl = P.get_bboxes_single(c2, cls[0], reg[0], dr[0])
    assert 0 < len(rl) < nv and (rl == 0).all()               # some survive, some are suppressed
    assert np.array_equal(labels[0].numpy(), rl.numpy()), (nv, len(rl), len(labels[0]))
    assert np.abs(scores[0].numpy() - rs.numpy()).max() <= 1e-6
    assert (np.abs(boxes[0].numpy() - rb.numpy()) / np.maximum(1.0, np.abs(rb.numpy()))).max() <= 1e-4
print("ok")
''')


def test_pointpillars_stream_returns_each_steps_detections_one_step_later():
    """``ml3d.engine.PointPillarsStream`` (what bench.py --workload pointpillars times): upload -> forward -> batched decode +
    NMS -> asynchronous copy back; ``submit`` hands out the previous step's lists, identical to ``get_bboxes`` of a plain
    forward of the same sweeps."""
    _run(r'''
import synth_data, synth_weights as W
from ml3d.engine import PointPillarsStream
from ml3d.torch.models.point_pillars import PointPillars
cfg = W.POINTPILLARS_SMALL_CFG
m = PointPillars(device="cpu", **cfg)
m.load_state_dict(W.pointpillars_state_dict(cfg, 4))
steps = [[torch.from_numpy(W.crop_for_cfg(synth_data.kitti_sweep(10 * s + i), cfg)) for i in range(2)] for s in range(3)]
want = [m.bbox_head.get_bboxes(*m(c)) for c in steps]          # the reference API path: NCHW head maps, then get_bboxes
for lanes in (1, 2, 3):                 # the step's sweeps dealt to 1 / 2 pipelines (3 lanes: clipped to the 2 sweeps of a step)
    pipe = PointPillarsStream(m, "cpu", lanes=lanes)
    got = [pipe.submit(c) for c in steps] + [pipe.flush()]
    assert got[0] is None and pipe.flush() is None
    for (wb, ws, wl), (gb, gs, gl) in zip(want, got[1:]):
        assert len(gb) == 2
        for i in range(2):
            assert torch.equal(wl[i], gl[i]) and torch.equal(wb[i], gb[i]) and torch.equal(ws[i], gs[i]) and len(gl[i]) > 0
print("ok")
''')


def test_one_pass_dense_radius_search_equals_the_two_phase_search_and_the_oracle():
    """``ops.radius_neighbors_dense`` = batch_neighbors (kpconv.py:2002-2034).  The one-traversal form (gather into a per-query
    stash, then expand) against the oracle and the two-phase form: batched supports != queries, an empty item, a query without
    neighbours, and a cluster of 200 coincident-ish points whose rows overflow the 128-entry stash (falls back, same matrix)."""
    _run(r'''
import synth_data
from oracle import kpconv_ref as K
from ml3d import ops
rng = np.random.default_rng(0)
a, b = synth_data.toronto3d_sphere(5, 1500), synth_data.toronto3d_sphere(6, 900)
sup = np.concatenate([a, b]).astype(np.float32)
qa = np.concatenate([K.batch_grid_subsampling(a, [len(a)], 0.3)[0], [[40, 40, 40]]]).astype(np.float32)     # last: no neighbours
qb = K.batch_grid_subsampling(b, [len(b)], 0.3)[0].astype(np.float32)
qry = np.concatenate([qa, qb])
for r in (0.25, 0.6):
    ref = K.batch_neighbors(qry, sup, [len(qa), 0, len(qb)], [len(a), 0, len(b)], r)
    got = ops.radius_neighbors_dense(torch.from_numpy(qry), torch.from_numpy(sup), [len(qa), 0, len(qb)], [len(a), 0, len(b)], r)
    assert got.dtype == torch.int32 and np.array_equal(got.numpy(), ref), r
    assert (got[len(qa) - 1] == len(sup)).all()
plan = ops.radius_plan_dense(torch.from_numpy(qry), torch.from_numpy(sup), [len(qa), 0, len(qb)], [len(a), 0, len(b)], 0.25)
assert type(plan).__name__ == "_DenseRadiusPlan" and plan.resolve().fallback is None
# overflow: 200 points within 1 cm -> rows of >= 200 neighbours
dense = np.concatenate([a[:300], a[7] + rng.normal(0, 0.003, (200, 3)).astype(np.float32)]).astype(np.float32)
ref = K.batch_neighbors(dense, dense, [len(dense)], [len(dense)], 0.25)
plan = ops.radius_plan_dense(torch.from_numpy(dense), torch.from_numpy(dense), [len(dense)], [len(dense)], 0.25)
got = ops.radius_fill_dense(plan, len(dense))
assert plan.fallback is not None and ref.shape[1] >= 200 and np.array_equal(got.numpy(), ref)
print("ok")
''')


def test_kpconv_batch_build_in_one_call_equals_the_per_layer_loop_and_the_oracle():
    """``ml3d_kpconv_batch_build`` (the whole 5-layer batch build enqueued from C++, one size read-back per layer) against the
    per-layer Python loop it replaces and against the oracle's ``segmentation_inputs`` (concat_batcher.py:186-305): every
    matrix, the pooled points, the per-item lengths and the consumed random draws identical -- rotated and axis-aligned grids,
    a first layer without convolution blocks, a batch item that pools to a handful of points; a row longer than the stash
    falls back to the per-layer path (same results), a short arena grows."""
    _run(r'''
import synth_data, synth_weights as W
from oracle import kpconv_ref as K
from ml3d import ops
from ml3d.ops import search
from ml3d.torch.models.kpconv import KPConvBatch
cfg = dict(W.TORONTO3D_CFG)
spheres = [synth_data.toronto3d_sphere(i, 2500 if i else 60, radius=2.0 if i else 1.2) for i in range(4)]
pts, lens = np.concatenate(spheres), [len(s) for s in spheres]
def same(a, b):
    assert len(a.points) == len(b.points)
    for l in range(len(a.points)):
        assert torch.equal(a.points[l], b.points[l]), l
        assert torch.equal(a.lengths[l], b.lengths[l]) and a.lengths[l].dtype == b.lengths[l].dtype, l
        for name in ("neighbors", "pools", "upsamples"):
            x, y = getattr(a, name)[l], getattr(b, name)[l]
            assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), (name, l, x.shape, y.shape)
for rot in ("random", None):
    np.random.seed(11); one = KPConvBatch(pts, lens, cfg, rotations=rot, device="cpu")
    after_one = np.random.rand()
    np.random.seed(11); loop = KPConvBatch(pts, lens, cfg, rotations=rot, device="cpu", one_call=False)
    after_loop = np.random.rand()
    assert getattr(one, "host_syncs", None) == cfg["num_layers"] and not hasattr(loop, "host_syncs")
    assert after_one == after_loop
    same(one, loop)
    if rot is not None:
        assert all(np.array_equal(a, b) for a, b in zip(one.rotations, loop.rotations))
        np.random.seed(11); seg = K.segmentation_inputs(pts, lens, cfg)
        for l in range(cfg["num_layers"]):
            for name in ("neighbors", "pools", "upsamples"):
                m = getattr(one, name)[l].numpy()
                if m.size or np.asarray(seg[name][l]).size:
                    assert np.array_equal(m, seg[name][l]), (name, l)
# an architecture whose first layer has no convolution blocks, three layers
cfg3 = dict(cfg, num_layers=3, architecture=["resnetb_strided", "resnetb", "resnetb_strided", "resnetb", "nearest_upsample", "unary"])
np.random.seed(3); one = KPConvBatch(pts, lens, cfg3, device="cpu")
np.random.seed(3); loop = KPConvBatch(pts, lens, cfg3, device="cpu", one_call=False)
assert one.host_syncs == 3 and one.neighbors[0].shape == (0, 1)
same(one, loop)
# a stash of 16 entries overflows on the first layer -> None (the caller's per-layer path); KPConvBatch hides that
r0 = cfg["first_subsampling_dl"] * cfg["conv_radius"]
t = torch.from_numpy(pts)
assert ops.kpconv_batch_build(t, lens, [r0, 2 * r0], [2 * r0 / 2.5, 0], [True, True], None, cap=16) is None
# a short arena: the call reports the bytes it needs and the wrapper grows the arena (same result)
search._ARENA_HINT.clear()
search._ARENA_HINT[(str(t.device), max(1, len(pts)).bit_length())] = 4096
np.random.seed(11); small = KPConvBatch(pts, lens, cfg, device="cpu")
np.random.seed(11); loop = KPConvBatch(pts, lens, cfg, device="cpu", one_call=False)
same(small, loop)
assert search._ARENA_HINT[(str(t.device), max(1, len(pts)).bit_length())] > 4096
print("ok")
''')


def test_kpconv_batch_build_in_one_call_edge_cases():
    """The one-call batch build on degenerate batches -- a single item, an EMPTY item between two spheres, one-point items,
    forty coincident points -- equals the per-layer loop matrix for matrix (rotated and axis-aligned pooling grids)."""
    _run(r'''
import synth_data, synth_weights as W
from ml3d.torch.models.kpconv import KPConvBatch
cfg = dict(W.TORONTO3D_CFG)
def same(a, b):
    assert len(a.points) == len(b.points)
    for l in range(len(a.points)):
        assert torch.equal(a.points[l], b.points[l]), ("points", l)
        assert torch.equal(a.lengths[l], b.lengths[l]), ("lengths", l, a.lengths[l], b.lengths[l])
        for name in ("neighbors", "pools", "upsamples"):
            x, y = getattr(a, name)[l], getattr(b, name)[l]
            assert x.shape == y.shape and torch.equal(x, y), (name, l, x.shape, y.shape)
cases = {
  "one item": [synth_data.toronto3d_sphere(3, 1500, radius=2.0)],
  "empty item in the middle": [synth_data.toronto3d_sphere(3, 800, radius=1.5), np.zeros((0, 3), np.float32), synth_data.toronto3d_sphere(4, 700, radius=1.5)],
  "one point items": [np.array([[0.1, 0.2, 0.3]], np.float32), synth_data.toronto3d_sphere(5, 600, radius=1.5), np.array([[5., 5., 5.]], np.float32)],
  "coincident points": [np.repeat(np.array([[1., 2., 3.]], np.float32), 40, 0), synth_data.toronto3d_sphere(6, 500, radius=1.2)],
}
for name, spheres in cases.items():
    pts, lens = np.concatenate(spheres).astype(np.float32), [len(s) for s in spheres]
    for rot in ("random", None):
        np.random.seed(2); one = KPConvBatch(pts, lens, cfg, rotations=rot, device="cpu")
        np.random.seed(2); loop = KPConvBatch(pts, lens, cfg, rotations=rot, device="cpu", one_call=False)
        same(one, loop)
        print(name, rot, "ok; one-call:", hasattr(one, "host_syncs"), [int(p.shape[0]) for p in one.points])
''')


def test_kpfcnn_with_deformable_blocks_matches_the_real_reference_golden():
    """``KPFCNN`` with ``resnetb_deformable`` / ``resnetb_deformable_strided`` blocks (kpconv_parislille3d.yml:28-32; here the
    three-layer KPCONV_DEFORM_SMALL_CFG): GPU-side batch build (deform radius on the deformable layers) and forward through the
    emulated library against the logits of the REAL reference's KPFCNN (tests/golden/kpconv_deform_small.npz), <= 1e-4; the
    state dict carries ``offset_conv.weights``, ``offset_conv.kernel_points`` and ``offset_bias`` like the reference's."""
    _run(r'''
import synth_data
from oracle import kpconv_ref as K
from ml3d.torch.models.kpconv import KPConvBatch, KPFCNN
cfg = dict(K.KPCONV_DEFORM_SMALL_CFG)
g = np.load(os.path.join(ROOT, "tests", "golden", "kpconv_deform_small.npz"))
spheres = [synth_data.toronto3d_sphere(int(f), int(g["max_points"])) for f in g["frame_ids"]]
np.random.seed(int(g["np_seed"]))
batch = KPConvBatch(np.concatenate(spheres), [len(s) for s in spheres], cfg, device="cpu")
for l in range(cfg["num_layers"]):
    nb = batch.neighbors[l].numpy().astype(np.int64)
    assert list(nb.shape) == list(g["nbr_shape%d" % l]) and np.int64((nb * (np.arange(nb.shape[1]) + 1)).sum()) == g["nbr_checksum%d" % l], l
m = KPFCNN(**cfg, device="cpu")
sd = K.make_state_dict(cfg, int(g["weights_seed"]))
assert set(m.state_dict().keys()) == set(sd.keys())
m.load_state_dict(sd)
out = m.eval()(batch).numpy()
assert out.shape == g["logits"].shape and np.abs(out - g["logits"]).max() <= 1e-4, np.abs(out - g["logits"]).max()
# a deformable block the kernels do not take is refused at construction (the open3d shim then falls back to the checkout)
bad = dict(cfg, KP_influence="gaussian")
try:
    KPFCNN(**bad, device="cpu")
    raise SystemExit("expected NotImplementedError")
except NotImplementedError:
    pass
print("ok")
''')


def test_kpfcnn_modulated_deformable_and_simple_deformable_blocks_match_the_oracle():
    """``modulated: true`` (offset_dim 60: 2 sigmoid modulations on the weighted features, kpconv.py:1017-1024, 1147-1149) and a
    ``simple_deformable`` block (kpconv.py:1321-1332) -- no reference config uses either, the reference code has both: native
    batch + forward on the emulated library against the oracle's restatement."""
    _run(r'''
import synth_data
from oracle import kpconv_ref as K
from ml3d.torch.models.kpconv import KPConvBatch, KPFCNN
cfg = dict(K.KPCONV_DEFORM_SMALL_CFG, modulated=True, num_layers=2,
           architecture=["simple", "resnetb_strided", "simple_deformable", "resnetb_deformable", "nearest_upsample", "unary"])
spheres = [synth_data.toronto3d_sphere(5, 1300), synth_data.toronto3d_sphere(6, 900)]
pts, lens = np.concatenate(spheres), [len(s) for s in spheres]
np.random.seed(4)
seg = K.segmentation_inputs(pts, lens, cfg)
np.random.seed(4)
batch = KPConvBatch(pts, lens, cfg, device="cpu")
for l in range(2):
    for name in ("neighbors", "pools", "upsamples"):
        assert np.array_equal(getattr(batch, name)[l].numpy(), seg[name][l]), (name, l)
sd = K.make_state_dict(cfg, 9)
assert sd["encoder_blocks.2.KPConv.offset_conv.weights"].shape[2] == 60 and sd["encoder_blocks.2.KPConv.offset_bias"].shape[0] == 60
m = KPFCNN(**cfg, device="cpu")
assert set(m.state_dict().keys()) == set(sd.keys())
m.load_state_dict(sd)
out = m.eval()(batch).numpy()
ref = K.forward(sd, cfg, K.to_torch_batch(seg), torch.ones((len(pts), 1))).numpy()
assert np.abs(out - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()
print("ok")
''')


def test_device_resident_patch_loop_reproduces_the_host_loop_patch_for_patch():
    """The model-class loop of ONE cloud with everything on the device (ml3d_nearest_to_center_dev, ml3d_patch_crop,
    ml3d_patch_recenter; only the data-free shuffle comes from the host) against the host loop that crops / recentres in
    numpy: same selected indices, bit-identical recentred coordinates and features (numpy's SEQUENTIAL float32 column means,
    its left-to-right float32 squared distances), identical possibilities and labels, patch after patch."""
    _run(r'''
import synth_data
from oracle import randlanet_ref as R
from ml3d.torch.models import RandLANet
for in_ch, aug in ((3, {"recenter": {"dim": [0, 1]}}), (6, {"recenter": {"dim": [0, 1, 2]}, "normalize": {"feat": {"method": "linear", "bias": 0, "scale": 255}}})):
    cfg = dict(num_neighbors=16, num_layers=2, num_points=640, num_classes=5, sub_sampling_ratio=[4, 4], in_channels=in_ch,
               dim_features=8, dim_output=[16, 32], grid_size=0.25, augment=aug)
    tile = synth_data.toronto3d_tile(3, half=3.0, density=0.08)
    data = dict(point=tile["point"], feat=tile["feat"] if in_ch == 6 else None, label=tile["label"])
    runs = []
    for device_loop in (False, True):
        m = RandLANet(**cfg, device="cpu", seed=9)
        m.load_state_dict(R.make_state_dict(cfg, 4))
        m.inference_begin(dict(data))
        assert m._dev_loop is not None and m.inference_data["point"].shape[0] > 640
        if not device_loop:
            m._dev_loop = None
        got = []
        for step in range(7):
            inp = m.inference_preprocess()["data"]
            res = m(inp)
            done = m.inference_end({"data": inp}, res)
            got.append({k: (np.asarray(v[0].cpu()) if isinstance(v, torch.Tensor) else [np.asarray(t[0].cpu()) for t in v])
                        for k, v in inp.items()} | {"logits": res.numpy().copy()})
        poss = m._dev_loop["possibility"].numpy() if device_loop else m.possibility
        runs.append((got, poss.copy(), m.test_probs.numpy().copy()))
    (a, pa, va), (b, pb, vb) = runs
    for step, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x["point_inds"], y["point_inds"]), step
        assert np.array_equal(x["coords"][0], y["coords"][0]) and np.array_equal(x["features"], y["features"]), step
        assert np.array_equal(x["labels"], y["labels"])
        for l in range(2):
            assert np.array_equal(x["neighbor_indices"][l], y["neighbor_indices"][l]) and np.array_equal(x["interp_idx"][l], y["interp_idx"][l])
        assert np.array_equal(x["logits"], y["logits"]), step
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
print("ok")
''')


def test_kpconv_autograd_function_matches_torch_autograd_of_the_oracle():
    """ops.KPConvFunction (HIP aggregation forward, hand-written HIP scatter backward + GEMMs) against torch's own autograd
    through the oracle's pure-torch KPConv (oracle/kpconv_ref.kpconv_rigid = kpconv.py:1005-1159): output, d/dx and d/dW, for the
    MFMA aggregation (cin 32), the generic one (cin 8), the first-layer one (cin 4), 96 channels (two lane chunks in the adjoint),
    strided queries and shadow columns anywhere."""
    _run(r'''
import synth_data
from oracle import kpconv_ref as K
from ml3d import ops
rng = np.random.default_rng(2)
s = synth_data.toronto3d_sphere(23, 700)
q = K.batch_grid_subsampling(s, [len(s)], 0.16)[0].astype(np.float32)
kp = K.synthetic_kernel_points(0.2)
for cin, cout, strided, infl in ((32, 16, False, 1), (8, 24, True, 1), (4, 32, False, 2), (96, 8, True, 1)):
    qq = q if strided else s
    inds = K.batch_neighbors(qq, s, [len(qq)], [len(s)], 0.2).astype(np.int32)
    inds = np.take_along_axis(inds, np.argsort(rng.random(inds.shape), axis=1), 1)        # shadows anywhere in a row
    x = torch.from_numpy(rng.standard_normal((len(s), cin)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((15, cin, cout)) * 0.2).astype(np.float32)).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((len(qq), cout)).astype(np.float32))
    tq, ts, ti, tk = torch.from_numpy(qq), torch.from_numpy(s), torch.from_numpy(inds), torch.from_numpy(kp)
    ref = K.kpconv_rigid(tq, ts, ti.long(), x, tk, w, 0.24, influence={1: "linear", 2: "gaussian"}[infl]) if "influence" in K.kpconv_rigid.__code__.co_varnames \
        else K.kpconv_rigid(tq, ts, ti.long(), x, tk, w, 0.24)
    if infl != 1 and "influence" not in K.kpconv_rigid.__code__.co_varnames:
        continue
    ref.backward(g)
    gx_ref, gw_ref = x.grad.clone(), w.grad.clone()
    x.grad = None; w.grad = None
    out = ops.KPConvFunction.apply(x, w, tq, ts, ti, tk, 0.24, infl)
    out.backward(g)
    sc = lambda t: max(1.0, float(t.abs().max()))
    assert (out - ref).abs().max() <= 1e-4 * sc(ref), (cin, float((out - ref).abs().max()))
    assert (x.grad - gx_ref).abs().max() <= 1e-4 * sc(gx_ref), (cin, float((x.grad - gx_ref).abs().max()))
    assert (w.grad - gw_ref).abs().max() <= 1e-4 * sc(gw_ref), (cin, float((w.grad - gw_ref).abs().max()))
print("ok")
''')


def test_attentive_pool_and_gather_max_functions_match_torch_autograd():
    """ops.AttentivePoolFunction against torch's autograd through the reference's formulation (randlanet.py:631-637:
    ``softmax(scores, dim=K)`` then ``sum(scores * x, dim=K)``), and ops.GatherMaxFunction against ``torch.max`` over the
    gathered neighbours (randlanet.py:318-327): outputs and every input gradient, for 8 / 16 / 100 channels, K = 16 and K = 5."""
    _run(r'''
from ml3d import ops
rng = np.random.default_rng(4)
for B, N, K, C in ((2, 70, 16, 16), (1, 33, 16, 100), (3, 9, 5, 8)):
    s = torch.from_numpy((rng.standard_normal((B, N, K, C)) * 3).astype(np.float32)).requires_grad_(True)
    x = torch.from_numpy(rng.standard_normal((B, N, K, C)).astype(np.float32)).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32))
    ref = (torch.softmax(s, dim=-2) * x).sum(-2)
    ref.backward(g)
    gs_ref, gx_ref = s.grad.clone(), x.grad.clone()
    s.grad = None; x.grad = None
    out = ops.AttentivePoolFunction.apply(s, x)
    out.backward(g)
    assert (out - ref).abs().max() <= 1e-5 and (s.grad - gs_ref).abs().max() <= 1e-5 and (x.grad - gx_ref).abs().max() <= 1e-5, (K, C)
for B, n_in, n_out, C in ((2, 64, 16, 32), (1, 40, 40, 8)):
    f = torch.from_numpy(rng.standard_normal((B, n_in, C)).astype(np.float32)).requires_grad_(True)
    idx = torch.from_numpy(rng.integers(0, n_in, (B, n_in, 16)).astype(np.int32))
    g = torch.from_numpy(rng.standard_normal((B, n_out, C)).astype(np.float32))
    ref = f[torch.arange(B)[:, None, None], idx[:, :n_out].long()].max(2)[0]
    ref.backward(g)
    gf_ref = f.grad.clone(); f.grad = None
    out = ops.GatherMaxFunction.apply(f, idx, n_out)
    out.backward(g)
    assert torch.equal(out, ref) and (f.grad - gf_ref).abs().max() <= 1e-5
# ADVICE r4: a pooled row whose 16 neighbour values are all -inf (or NaN) must send its gradient to one of ITS neighbours
# (the first listed, like torch.max's backward on ties), never to row 0 of batch item 0
f = torch.from_numpy(rng.standard_normal((2, 24, 4)).astype(np.float32))
idx = torch.from_numpy(rng.integers(8, 24, (2, 24, 16)).astype(np.int32))
f[1, 8:, 2] = float("-inf")
f = f.requires_grad_(True)
g = torch.ones((2, 6, 4))
ref = f[torch.arange(2)[:, None, None], idx[:, :6].long()].max(2)[0]
ref.backward(g)
gf_ref = f.grad.clone(); f.grad = None
out = ops.GatherMaxFunction.apply(f, idx, 6)
out.backward(g)
assert torch.equal(out, ref) and torch.equal(f.grad, gf_ref), (f.grad - gf_ref).abs().max()
assert f.grad[0, 0].abs().sum() == 0 and f.grad[1, :, 2].sum() == 6
print("ok")
''')


def test_pointpillars_training_forward_and_gradients_match_the_reference():
    """``PointPillars`` in ``.train()`` mode (voxelize on the library, decorations / PFN / scatter as batched torch expressions,
    SECOND / FPN / heads on the class's own torch modules) against ONE training forward + backward of the REAL reference model
    (tests/golden/train_pointpillars.npz, oracle/gen_golden_train.py: the sum of the three ``get_loss`` terms,
    object_detection.py:273-283): head maps, loss terms, parameter gradients, a BatchNorm running mean."""
    _run(r'''
from ml3d.torch.models import PointPillars
from oracle import pointpillars_ref as P
from oracle.gen_golden_train import pp_train_inputs, PP_LOSS_CFG
import synth_weights
g = np.load(os.path.join(ROOT, "tests", "golden", "train_pointpillars.npz"))
cfg = synth_weights.POINTPILLARS_SMALL_CFG
m = PointPillars(device="cpu", loss=PP_LOSS_CFG, **cfg)
m.load_state_dict(P.make_state_dict(cfg, 21))
m.train()
clouds, boxes, labels = pp_train_inputs()
assert [len(c) for c in clouds] == list(g["n_points"])
class In:
    point = [torch.from_numpy(c) for c in clouds]
    bboxes = boxes
In.labels = labels
maps = m(In)
for name, t in zip(("cls", "reg", "dir"), maps):
    want = g[name]
    assert t.requires_grad and np.abs(t.detach().numpy()[:, :, ::2, ::2] - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), name
terms = m.get_loss(maps, In)
got = np.array([float(terms["loss_cls"]), float(terms["loss_bbox"]), float(terms["loss_dir"])])
assert np.abs(got - g["loss"]).max() <= 1e-4 * np.abs(g["loss"]).max(), (got, g["loss"])
sum(terms.values()).backward()
named = dict(m.named_parameters())
checked = 0
for key in g.files:
    if key.startswith("grad:"):
        want, have = g[key], named[key[5:]].grad.numpy()
        assert np.abs(have - want).max() <= 1e-3 * float(np.abs(want).max()), (key, float(np.abs(have - want).max()), float(np.abs(want).max()))
        checked += 1
assert checked == 11
rm = dict(m.named_buffers())["backbone.blocks.0.1.running_mean"].numpy()
assert np.abs(rm - g["running_mean:backbone.blocks.0.1"]).max() <= 1e-5
# back to inference: the fused kernels see the weights as they are now
m.eval()
out = m(In)
assert not out[0].requires_grad and out[0].shape == maps[0].shape
print("ok")
''')


def test_deformable_kpfcnn_training_forward_regulariser_and_gradients_match_the_reference():
    """KPFCNN with three DEFORMABLE, modulated blocks in ``.train()`` mode against ONE training forward + backward of the REAL
    reference model (tests/golden/train_kpconv_deform.npz): logits, the cross entropy, the point-to-point offset regulariser of
    ``get_loss`` (kpconv.py:2167-2206), and the gradients -- those of the offset convolutions included, which only flow if the
    influences are differentiated with respect to the deformed kernel points."""
    _run(r'''
from ml3d.torch.dataloaders import kpconv_input_features
from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
from oracle import kpconv_ref as K
from oracle.gen_golden_train import DEFORM_TRAIN_CFG, deform_train_inputs
g = np.load(os.path.join(ROOT, "tests", "golden", "train_kpconv_deform.npz"))
cfg = dict(DEFORM_TRAIN_CFG)
m = KPFCNN(**cfg, device="cpu")
m.load_state_dict(K.make_state_dict(cfg, 78))
spheres, cols, labels = deform_train_inputs()
pts, columns = np.concatenate(spheres), np.concatenate(cols)
np.random.seed(32)
batch = KPConvBatch(pts, [len(s) for s in spheres], cfg, features=kpconv_input_features(pts, columns, cfg["in_features_dim"]).astype(np.float32),
                    device="cpu")
batch.labels = torch.from_numpy(np.concatenate(labels).astype(np.int64))
m.train()
logits = m(batch)
assert logits.requires_grad and np.abs(logits.detach().numpy() - g["logits"]).max() <= 1e-4 * max(1.0, float(np.abs(g["logits"]).max()))
L = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
loss, lab, scores = m.get_loss(L, logits, {"data": batch}, "cpu")
assert int(lab.numel()) == int(g["n_valid"])
assert abs(float(m.output_loss) - float(g["output_loss"])) <= 1e-5 and abs(float(m.reg_loss) - float(g["reg_loss"])) <= 1e-4 * float(g["reg_loss"])
assert abs(float(loss) - float(g["loss"])) <= 1e-4 * float(g["loss"])
loss.backward()
named = dict(m.named_parameters())
checked = 0
for key in g.files:
    if key.startswith("grad:"):
        want, have = g[key], named[key[5:]].grad.numpy()
        assert np.abs(have - want).max() <= 1e-3 * float(np.abs(want).max()), (key, float(np.abs(have - want).max()), float(np.abs(want).max()))
        checked += 1
assert checked == 11
print("ok")
''')
