"""CPU ORACLE for the KPConv (rigid) inference path — test infrastructure, NOT product code.

Restates, as plain functions over numpy / PyTorch-CPU tensors,
  * ``KPConvBatch.segmentation_inputs``  (ml3d/torch/dataloaders/concat_batcher.py:186-305)
  * ``batch_neighbors`` / ``batch_grid_subsampling`` (ml3d/torch/models/kpconv.py:2002-2164)
  * ``KPFCNN.__init__`` block walk and ``KPFCNN.forward`` (ml3d/torch/models/kpconv.py:131-291)
  * ``KPConv.forward`` rigid branch, ``UnaryBlock``, ``BatchNormBlock``, ``SimpleBlock``,
    ``ResnetBottleneckBlock``, ``NearestUpsampleBlock``, ``max_pool``, ``closest_pool``
    (kpconv.py:772-858, 1005-1159, 1213-1491)
on top of the oracle's neighbour / subsample ops (oracle/ops.py).

PINNED: ``oracle/gen_golden.py`` runs the REAL reference modules (KPFCNN + KPConvBatch imported from
/root/reference through oracle/ref_shim.py) on the same seeded spheres and weights, asserts this
restatement reproduces the batch (exact) and the logits (<= 1e-5), and stores the result in
tests/golden/kpconv_*.npz.  Deformable blocks are out of scope.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops as oops
from synth_weights import (TORONTO3D_CFG, KPCONV_DEFORM_SMALL_CFG, PARISLILLE3D_CFG, arch_plan, synthetic_kernel_points,  # noqa: F401  (input generation)
                           kpconv_state_dict as make_state_dict)



# ---------------------------------------------------------------------------------------------------
# batcher (concat_batcher.py:186-305)
# ---------------------------------------------------------------------------------------------------
def create_3d_rotations(axis, angle):
    """ml3d/datasets/utils/operations.py:10-41 (Rodrigues formula, same term order)."""
    t1 = np.cos(angle)
    t2 = 1 - t1
    t3 = axis[:, 0] * axis[:, 0]
    t6 = t2 * axis[:, 0]
    t7 = t6 * axis[:, 1]
    t8 = np.sin(angle)
    t9 = t8 * axis[:, 2]
    t11 = t6 * axis[:, 2]
    t12 = t8 * axis[:, 1]
    t15 = axis[:, 1] * axis[:, 1]
    t19 = t2 * axis[:, 1] * axis[:, 2]
    t20 = t8 * axis[:, 0]
    t24 = axis[:, 2] * axis[:, 2]
    R = np.stack([t1 + t2 * t3, t7 - t9, t11 + t12, t7 + t9, t1 + t2 * t15, t19 - t20, t11 - t12, t19 + t20,
                  t1 + t2 * t24], axis=1)
    return np.reshape(R, (-1, 3, 3))


def random_grid_rotations(B):
    """The np.random draws of batch_grid_subsampling (kpconv.py:2059-2080), in the same order."""
    theta = np.random.rand(B) * 2 * np.pi
    phi = (np.random.rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
    alpha = np.random.rand(B) * 2 * np.pi
    return create_3d_rotations(u.T, alpha).astype(np.float32)


def rotate_rows(points, lengths, R, transpose=False):
    """points[i0:i0+len] = sum(expand_dims(p, 2) * R[b], axis=1)  (kpconv.py:2086-2092 / 2105-2110)."""
    out = points.copy()
    i0 = 0
    for b, ln in enumerate(lengths):
        ln = int(ln)
        M = R[b].T if transpose else R[b]
        out[i0:i0 + ln, :] = np.sum(np.expand_dims(out[i0:i0 + ln, :], 2) * M, axis=1)
        i0 += ln
    return out


def batch_grid_subsampling(points, batches_len, sampleDl, R=None):
    """kpconv.py:2037-2111 (points only).  R = per-item rotation matrices or None (random_grid_orient=False)."""
    pts = points if R is None else rotate_rows(points, batches_len, R)
    s_points, s_len = oops.subsample_batch(pts, batches_len, sampleDl=sampleDl)
    if R is not None:
        s_points = rotate_rows(s_points, s_len, R, transpose=True)
    return s_points, s_len


def batch_neighbors(queries, supports, q_batches, s_batches, radius):
    """kpconv.py:2002-2034: dense int32 [Nq, max_nbrs], shadow index = Ns."""
    qs = np.concatenate([[0], np.cumsum(q_batches)]).astype(np.int64)
    ss = np.concatenate([[0], np.cumsum(s_batches)]).astype(np.int64)
    r = oops.fixed_radius_search(supports, queries, radius, ss, qs)
    splits = r.neighbors_row_splits
    max_nbrs = int((splits[1:] - splits[:-1]).max()) if len(splits) > 1 else 0
    dense = oops.ragged_to_dense(r.neighbors_index.reshape(-1, 1), splits, max_nbrs,
                                 np.array([supports.shape[0]], np.int32))
    return dense[:, :, 0]


def segmentation_inputs(stacked_points, stack_lengths, cfg, rotations="random"):
    """concat_batcher.py:186-305.  Layers with a deformable block search with ``deform_radius`` instead of ``conv_radius``
    (:219-252).  ``rotations``: "random" (draw from np.random like
    the reference), None (axis-aligned pooling grids) or a list with one [B,3,3] array per pooling layer.
    Returns dict(points, neighbors, pools, upsamples, lengths, rotations)."""
    r_normal = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    layer_blocks = []
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], lengths=[], rotations=[])
    stack_lengths = np.asarray(stack_lengths, np.int32)
    r_deform = lambda: r_normal * cfg.get("deform_radius", 6.0) / cfg["conv_radius"]
    for block in cfg["architecture"]:
        if not ("pool" in block or "strided" in block or "global" in block or "upsample" in block):
            layer_blocks.append(block)
            continue
        if layer_blocks:
            r = r_deform() if any("deformable" in b for b in layer_blocks) else r_normal
            conv_i = batch_neighbors(stacked_points, stacked_points, stack_lengths, stack_lengths, r)
        else:
            conv_i = np.zeros((0, 1), np.int32)
        if "pool" in block or "strided" in block:
            dl = 2 * r_normal / cfg["conv_radius"]
            li = len(out["points"])
            if isinstance(rotations, str):
                R = random_grid_rotations(len(stack_lengths))
            elif rotations is None:
                R = None
            else:
                R = rotations[li]
            pool_p, pool_b = batch_grid_subsampling(stacked_points, stack_lengths, dl, R)
            r_pool = r_deform() if "deformable" in block else r_normal
            pool_i = batch_neighbors(pool_p, stacked_points, pool_b, stack_lengths, r_pool)
            # (concat_batcher.py:262-263: `2 * r` with r as the pooling branch left it -- the deform radius after a deformable
            #  strided block)
            up_i = batch_neighbors(stacked_points, pool_p, stack_lengths, pool_b, 2 * r_pool)
            out["rotations"].append(R)
        else:
            pool_i = np.zeros((0, 1), np.int32)
            pool_p = np.zeros((0, 3), np.float32)
            pool_b = np.zeros((0,), np.int32)
            up_i = np.zeros((0, 1), np.int32)
        out["points"].append(stacked_points)
        out["neighbors"].append(conv_i.astype(np.int64))
        out["pools"].append(pool_i.astype(np.int64))
        out["upsamples"].append(up_i.astype(np.int64))
        out["lengths"].append(stack_lengths)
        stacked_points, stack_lengths = pool_p, pool_b
        r_normal *= 2
        layer_blocks = []
        if "global" in block or "upsample" in block:
            break
    return out


# ---------------------------------------------------------------------------------------------------
# forward (kpconv.py:270-291 and the block classes)
# ---------------------------------------------------------------------------------------------------
def _bn_block(sd, prefix, x, use_bn):
    """BatchNormBlock.forward (kpconv.py:1239-1250), eval mode."""
    if use_bn:
        return F.batch_norm(x, sd[prefix + ".batch_norm.running_mean"], sd[prefix + ".batch_norm.running_var"],
                            sd[prefix + ".batch_norm.weight"], sd[prefix + ".batch_norm.bias"], False, 0.0, 1e-5)
    return x + sd[prefix + ".bias"]


def _unary(sd, prefix, x, cfg, use_bn=True, relu=True):
    """UnaryBlock.forward (kpconv.py:1288-1293)."""
    x = x @ sd[prefix + ".mlp.weight"].t()
    x = _bn_block(sd, prefix + ".batch_norm", x, use_bn and cfg["use_batch_norm"])
    return F.leaky_relu(x, cfg.get("l_relu", 0.1)) if relu else x


def kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent):
    """KPConv.forward, non-deformable, KP_influence='linear', aggregation 'sum' (kpconv.py:1048-1159)."""
    s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1, :]) + 1e6), 0)
    neighbors = s_pad[neighb_inds, :] - q_pts.unsqueeze(1)
    differences = neighbors.unsqueeze(2) - kernel_points
    sq = torch.sum(differences ** 2, dim=3)
    w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0).transpose(1, 2)
    xp = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    nx = xp[neighb_inds]
    wf = torch.matmul(w, nx).permute(1, 0, 2)
    return torch.sum(torch.matmul(wf, weights), dim=0)


def kpconv_deformable(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, offset_weights, offset_bias,
                      modulated=False):
    """KPConv.forward, deformable branch, KP_influence='linear', aggregation 'sum' (kpconv.py:1011-1159): offsets from the
    inner rigid convolution, deformed kernel points per query, the in-range pruning of the neighbour lists (:1071-1103;
    it only drops neighbours whose influence is 0), weighted sum, optional modulations, kernel weights."""
    K = kernel_points.shape[0]
    off = kpconv_rigid(q_pts, s_pts, neighb_inds, x, kernel_points, offset_weights, extent) + offset_bias
    if modulated:
        unscaled = off[:, :3 * K].view(-1, K, 3)
        modulations = 2 * torch.sigmoid(off[:, 3 * K:])
    else:
        unscaled, modulations = off.view(-1, K, 3), None
    offsets = unscaled * extent
    s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1, :]) + 1e6), 0)
    neighbors = s_pad[neighb_inds, :] - q_pts.unsqueeze(1)
    deformed = (offsets + kernel_points).unsqueeze(1)                       # [n, 1, K, 3]
    sq = torch.sum((neighbors.unsqueeze(2) - deformed) ** 2, dim=3)         # [n, H, K]
    in_range = torch.any(sq < extent ** 2, dim=2).type(torch.int32)
    new_max = int(torch.max(torch.sum(in_range, dim=1)))
    row_bool, row_inds = torch.topk(in_range, new_max, dim=1)
    new_inds = neighb_inds.gather(1, row_inds)
    sq = sq.gather(1, row_inds.unsqueeze(2).expand(-1, -1, K))
    new_inds = new_inds * row_bool
    new_inds = new_inds - (row_bool.type(torch.int64) - 1) * int(s_pad.shape[0] - 1)
    w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0).transpose(1, 2)
    xp = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    wf = torch.matmul(w, xp[new_inds])
    if modulations is not None:
        wf = wf * modulations.unsqueeze(2)
    return torch.sum(torch.matmul(wf.permute(1, 0, 2), weights), dim=0)


def _conv(sd, p, b, q_pts, s_pts, inds, x, cfg):
    """The block's KPConv (rigid or deformable by the block name, kpconv.py:1321-1332)."""
    if "deform" in b["name"]:
        return kpconv_deformable(q_pts, s_pts, inds, x, sd[p + ".kernel_points"], sd[p + ".weights"], b["extent"],
                                 sd[p + ".offset_conv.weights"], sd[p + ".offset_bias"], cfg.get("modulated", False))
    return kpconv_rigid(q_pts, s_pts, inds, x, sd[p + ".kernel_points"], sd[p + ".weights"], b["extent"])


def max_pool(x, inds):
    """kpconv.py:841-858."""
    xp = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    return xp[inds].max(1)[0]


def closest_pool(x, inds):
    """kpconv.py:821-838."""
    xp = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    return xp[inds[:, 0]]


@torch.no_grad()
def forward(sd, cfg, batch, features):
    """KPFCNN.forward.  batch = dict(points, neighbors, pools, upsamples) of torch tensors per layer."""
    plan = arch_plan(cfg)
    lr = cfg.get("l_relu", 0.1)
    ubn = cfg["use_batch_norm"]
    x = features.clone()
    skip_x = []
    for i, b in enumerate(plan["encoder"]):
        if i in plan["encoder_skips"]:
            skip_x.append(x)
        p = "encoder_blocks.%d" % i
        L = b["layer"]
        strided = "strided" in b["name"]
        q_pts = batch["points"][L + 1] if strided else batch["points"][L]
        s_pts = batch["points"][L]
        inds = batch["pools"][L] if strided else batch["neighbors"][L]
        if "simple" in b["name"]:
            y = _conv(sd, p + ".KPConv", b, q_pts, s_pts, inds, x, cfg)
            x = F.leaky_relu(_bn_block(sd, p + ".batch_norm", y, ubn), lr)
        else:
            mid = b["out_dim"] // 4
            y = _unary(sd, p + ".unary1", x, cfg) if b["in_dim"] != mid else x
            y = _conv(sd, p + ".KPConv", b, q_pts, s_pts, inds, y, cfg)
            y = F.leaky_relu(_bn_block(sd, p + ".batch_norm_conv", y, ubn), lr)
            y = _unary(sd, p + ".unary2", y, cfg, relu=False)
            sc = max_pool(x, inds) if strided else x
            if b["in_dim"] != b["out_dim"]:
                sc = _unary(sd, p + ".unary_shortcut", sc, cfg, relu=False)
            x = F.leaky_relu(y + sc, lr)
    for i, b in enumerate(plan["decoder"]):
        if i in plan["decoder_concats"]:
            x = torch.cat([x, skip_x.pop()], dim=1)
        if "upsample" in b["name"]:
            x = closest_pool(x, batch["upsamples"][b["layer"] - 1])
        else:
            x = _unary(sd, "decoder_blocks.%d" % i, x, cfg)
    h0, h1 = plan["head"]
    x = _unary(sd, "head_mlp", x, cfg, use_bn=h0["bn"], relu=h0["relu"])
    x = _unary(sd, "head_softmax", x, cfg, use_bn=h1["bn"], relu=h1["relu"])
    return x


def to_torch_batch(seg):
    return dict(points=[torch.from_numpy(np.ascontiguousarray(p)) for p in seg["points"]],
                neighbors=[torch.from_numpy(a) for a in seg["neighbors"]],
                pools=[torch.from_numpy(a) for a in seg["pools"]],
                upsamples=[torch.from_numpy(a) for a in seg["upsamples"]])
