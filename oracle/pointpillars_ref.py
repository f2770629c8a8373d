"""CPU ORACLE for the PointPillars inference forward — test infrastructure, NOT product code.

Functional PyTorch-CPU restatement of
  * ``PointPillarsVoxelization.forward``  ml3d/torch/models/point_pillars.py:328-382
  * ``PillarFeatureNet.forward`` / ``PFNLayer.forward``          :512-555, 417-453
  * ``PointPillarsScatter.forward``                              :577-616
  * ``SECOND.forward`` / ``SECONDFPN.forward`` / ``Anchor3DHead.forward``   :619-841
  * ``PointPillars.voxelize`` / ``extract_feats`` / ``forward``  :102-138
on top of the oracle's voxelize / ragged_to_dense (oracle/ops.py).

PINNED: ``oracle/gen_golden.py`` runs the REAL reference PointPillars module (imported from /root/reference
through oracle/ref_shim.py) with the same weights and clouds, asserts this restatement agrees (<= 1e-5) and
writes tests/golden/pointpillars_*.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops as oops
from synth_weights import (POINTPILLARS_KITTI_CFG as KITTI_CFG, POINTPILLARS_SMALL_CFG as SMALL_CFG,  # noqa: F401
                           pointpillars_state_dict as make_state_dict, crop_for_cfg)

def voxelization(points_feats, cfg, training=False):
    """PointPillarsVoxelization.forward (point_pillars.py:328-382) for one sample [N, 3+C]."""
    vz = cfg["voxelize"]
    pcr = cfg["point_cloud_range"]
    voxel_size = torch.Tensor(vz["voxel_size"])
    rmin, rmax = torch.Tensor(pcr[:3]), torch.Tensor(pcr[3:])
    mv = vz["max_voxels"]
    max_voxels = (mv[0] if training else mv[1]) if isinstance(mv, (list, tuple)) else mv
    num_voxels = ((rmax - rmin) / voxel_size).type(torch.int32)
    pts = points_feats[:, :3].contiguous().numpy()
    ans = oops.voxelize(pts, np.array([0, pts.shape[0]], np.int64), voxel_size.numpy(), rmin.numpy(), rmax.numpy(),
                        vz["max_num_points"], max_voxels)
    feats = torch.cat([torch.zeros_like(points_feats[0:1, :]), points_feats])
    dense = oops.ragged_to_dense(ans.voxel_point_indices, ans.voxel_point_row_splits, vz["max_num_points"],
                                 np.int64(-1)) + 1
    out_voxels = feats[torch.from_numpy(dense)]
    out_coords = torch.from_numpy(ans.voxel_coords[:, [2, 1, 0]].copy())
    prs = torch.from_numpy(ans.voxel_point_row_splits)
    out_num = prs[1:] - prs[:-1]
    inb = torch.logical_and(out_coords[:, 2] < num_voxels[0], out_coords[:, 1] < num_voxels[1])
    return out_voxels[inb], out_coords[inb], out_num[inb]


def _bn(sd, prefix, x, eps=1e-3):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, eps)


def pillar_feature_net(sd, cfg, features, num_points, coors):
    """PillarFeatureNet.forward (point_pillars.py:512-555) + PFNLayer.forward (:417-453)."""
    ve = cfg["voxel_encoder"]
    pcr = cfg["point_cloud_range"]
    vx, vy = ve["voxel_size"][0], ve["voxel_size"][1]
    x_off, y_off = vx / 2 + pcr[0], vy / 2 + pcr[1]
    mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(features).view(-1, 1, 1)
    f_cluster = features[:, :, :3] - mean
    f_center = features[:, :, :2].clone()
    f_center[:, :, 0] = f_center[:, :, 0] - (coors[:, 3].type_as(features).unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = f_center[:, :, 1] - (coors[:, 2].type_as(features).unsqueeze(1) * vy + y_off)
    f = torch.cat([features, f_cluster, f_center], dim=-1)
    n = f.shape[1]
    mask = (num_points.view(-1, 1).int() > torch.arange(n, dtype=torch.int).view(1, -1)).unsqueeze(-1).type_as(f)
    f = f * mask
    nl = len(ve["feat_channels"])
    for i in range(nl):
        p = "voxel_encoder.pfn_layers.%d" % i
        x = F.linear(f, sd[p + ".linear.weight"])
        x = _bn(sd, p + ".norm", x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if i == nl - 1:
            f = x_max
        else:
            f = torch.cat([x, x_max.repeat(1, n, 1)], dim=2)
    return f.squeeze(dim=1)


def scatter(cfg, voxel_features, coors, batch_size):
    """PointPillarsScatter.forward (point_pillars.py:577-616)."""
    ny, nx = cfg["scatter"]["output_shape"]
    C = cfg["scatter"]["in_channels"]
    out = []
    for b in range(batch_size):
        canvas = torch.zeros(C, nx * ny, dtype=voxel_features.dtype)
        m = coors[:, 0] == b
        tc = coors[m, :]
        idx = (tc[:, 2] * nx + tc[:, 3]).long()
        canvas[:, idx] = voxel_features[m, :].t()
        out.append(canvas)
    return torch.stack(out, 0).view(batch_size, C, ny, nx)


def backbone_neck_head(sd, cfg, x):
    """SECOND.forward, SECONDFPN.forward, Anchor3DHead.forward (point_pillars.py:666-682, 739-755, 827-841)."""
    bb, nk = cfg["backbone"], cfg["neck"]
    outs = []
    for i, ln in enumerate(bb["layer_nums"]):
        x = F.relu(_bn(sd, "backbone.blocks.%d.1" % i,
                       F.conv2d(x, sd["backbone.blocks.%d.0.weight" % i], stride=bb["layer_strides"][i], padding=1)))
        for j in range(ln):
            x = F.relu(_bn(sd, "backbone.blocks.%d.%d" % (i, 4 + 3 * j),
                           F.conv2d(x, sd["backbone.blocks.%d.%d.weight" % (i, 3 + 3 * j)], padding=1)))
        outs.append(x)
    ups = []
    for i in range(len(nk["out_channels"])):
        s = nk["upsample_strides"][i]
        y = F.conv_transpose2d(outs[i], sd["neck.deblocks.%d.0.weight" % i], stride=s)
        ups.append(F.relu(_bn(sd, "neck.deblocks.%d.1" % i, y)))
    feat = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
    return tuple(F.conv2d(feat, sd["bbox_head.%s.weight" % n], sd["bbox_head.%s.bias" % n])
                 for n in ("conv_cls", "conv_reg", "conv_dir_cls"))


@torch.no_grad()
def forward(sd, cfg, points_list):
    """PointPillars.forward (point_pillars.py:102-138) for a list of [N_i, 3+C] clouds -> (cls, reg, dir)."""
    voxels, coors, nums = [], [], []
    for i, pts in enumerate(points_list):
        v, c, n = voxelization(pts, cfg)
        voxels.append(v)
        coors.append(F.pad(c, (1, 0), mode="constant", value=i))
        nums.append(n)
    voxels, coors, nums = torch.cat(voxels), torch.cat(coors), torch.cat(nums)
    vf = pillar_feature_net(sd, cfg, voxels, nums, coors)
    x = scatter(cfg, vf, coors, len(points_list))
    return backbone_neck_head(sd, cfg, x), dict(voxels=voxels, coors=coors, num_points=nums, pillar_features=vf)




# ---------------------------------------------------------------------------------------------------
# box decoding + NMS (Anchor3DHead.get_bboxes, point_pillars.py:945-1025; objdet_helper.py:53-350)
# ---------------------------------------------------------------------------------------------------
def grid_anchors(cfg, featmap_size):
    """Anchor3DRangeGenerator.grid_anchors (objdet_helper.py:164-244): [H*W*sizes*rots, 7], order (y, x, size, rot)."""
    hd = cfg["head"]
    H, W = featmap_size
    rots = torch.tensor(hd["rotations"], dtype=torch.float32)
    out = []
    ranges = hd["ranges"] if len(hd["ranges"]) == len(hd["sizes"]) else hd["ranges"] * len(hd["sizes"])
    for rng, size in zip(ranges, hd["sizes"]):
        r = torch.tensor(rng, dtype=torch.float32)
        zc = torch.linspace(r[2], r[5], 1)
        yc = torch.linspace(r[1], r[4], H)
        xc = torch.linspace(r[0], r[3], W)
        a = torch.zeros((1, H, W, 1, len(rots), 7))
        a[..., 0] = xc.view(1, 1, W, 1, 1)
        a[..., 1] = yc.view(1, H, 1, 1, 1)
        a[..., 2] = zc.view(1, 1, 1, 1, 1)
        a[..., 3:6] = torch.tensor(size, dtype=torch.float32)
        a[..., 6] = rots.view(1, 1, 1, 1, -1)
        out.append(a)
    return torch.cat(out, dim=-3).reshape(-1, 7)


def decode(anchors, deltas):
    """BBoxCoder.decode (objdet_helper.py:286-313)."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, wt, lt, ht, rt = torch.split(deltas, 1, dim=-1)
    za = za + ha / 2
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * ha + za
    lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
    return torch.cat([xg, yg, zg - hg / 2, wg, lg, hg, rt + ra], dim=-1)


def bev_xyxyr(boxes):
    """xywhr_to_xyxyr(box3d_to_bev(boxes)) (objdet_helper.py:69-100)."""
    b = boxes[:, [0, 1, 3, 4, 6]]
    out = torch.zeros_like(b)
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4] = b[:, 0] - hw, b[:, 1] - hh, b[:, 0] + hw, b[:, 1] + hh, b[:, 4]
    return out


@torch.no_grad()
def get_bboxes_single(cfg, cls_scores, bbox_preds, dir_preds, nms_fn=None):
    """Anchor3DHead.get_bboxes_single (point_pillars.py:965-1025) for one sample's [C, H, W] maps."""
    hd = cfg["head"]
    nc = len(cfg["classes"])
    nms_fn = nms_fn or (lambda b, s, t: torch.from_numpy(oops.nms(b.numpy(), s.numpy(), t)))
    anchors = grid_anchors(cfg, cls_scores.shape[-2:])
    dir_scores = torch.max(dir_preds.permute(1, 2, 0).reshape(-1, 2), dim=-1)[1]
    scores = cls_scores.permute(1, 2, 0).reshape(-1, nc).sigmoid()
    bbox_preds = bbox_preds.permute(1, 2, 0).reshape(-1, 7)
    if scores.shape[0] > hd["nms_pre"]:
        max_scores, _ = scores.max(dim=1)
        _, topk = max_scores.topk(hd["nms_pre"])      # torch's own top-k, as the reference calls it: EQUAL scores come out in an order torch
        # does not specify (it differs between its CPU and GPU kernels); oops.topk_rows fixes one for the primitive's tests, and the
        # goldens from the real reference that run through ties at the cut (pointpillars_argoverse) are reproduced only with this call
        anchors, bbox_preds, scores, dir_scores = anchors[topk], bbox_preds[topk], scores[topk], dir_scores[topk]
    bboxes = decode(anchors, bbox_preds)
    idxs = []
    for i in range(nc):
        m = scores[:, i] > hd["score_thr"]
        if not m.any():
            idxs.append(torch.tensor([], dtype=torch.long))
            continue
        orig = torch.arange(m.shape[0])[m]
        idxs.append(orig[nms_fn(bev_xyxyr(bboxes[m]), scores[m, i], 0.01)])
    labels = torch.cat([torch.full((len(idxs[i]),), i, dtype=torch.long) for i in range(nc)])
    sc = torch.cat([scores[idxs[i], i] for i in range(nc)])
    idx = torch.cat(idxs)
    bboxes, dir_scores = bboxes[idx], dir_scores[idx]
    if bboxes.shape[0] > 0:
        off = hd.get("dir_offset", 0)
        val = bboxes[..., 6] - off
        dir_rot = val - torch.floor(val / np.pi + 1) * np.pi
        bboxes[..., 6] = dir_rot + off + np.pi * dir_scores.to(bboxes.dtype)
    return bboxes, sc, labels
