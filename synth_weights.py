"""Seeded pseudo-trained weights and the benchmark configurations of the three models.

Bench / test INPUT generation only (like synth_data.py): deterministic state_dicts with the reference's
parameter names and shapes — there is no network for published checkpoints — plus the
inference-relevant keys of ml3d/configs/{randlanet_semantickitti,kpconv_toronto3d,pointpillars_kitti}.yml.
Neither the product package nor the oracle's algorithms live here.
"""
from collections import OrderedDict

import numpy as np
import torch

RANDLANET_SEMANTICKITTI_CFG = dict(   # ml3d/configs/randlanet_semantickitti.yml:17-33
    num_neighbors=16, num_layers=4, num_points=45056, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4],
    in_channels=3, dim_features=8, dim_output=[16, 64, 128, 256])


# ---------------------------------------------------------------------------------------------------
# RandLA-Net
# ---------------------------------------------------------------------------------------------------
def encoder_dims(cfg):
    """encoder_dim_list of randlanet.py:81-91."""
    out = []
    for i in range(cfg["num_layers"]):
        d = 2 * cfg["dim_output"][i]
        if i == 0:
            out.append(d)
        out.append(d)
    return out


def _conv(shapes, name, cin, cout, bn=True, transpose=False):
    shapes[name + ".conv.weight"] = (cin, cout, 1, 1) if transpose else (cout, cin, 1, 1)
    shapes[name + ".conv.bias"] = (cout,)
    if bn:
        _bn(shapes, name + ".batch_norm", cout)


def _bn(shapes, name, c):
    shapes[name + ".weight"] = (c,)
    shapes[name + ".bias"] = (c,)
    shapes[name + ".running_mean"] = (c,)
    shapes[name + ".running_var"] = (c,)
    shapes[name + ".num_batches_tracked"] = ()


def param_shapes(cfg):
    """state_dict keys/shapes of the reference RandLANet."""
    s = OrderedDict()
    s["fc0.weight"] = (cfg["dim_features"], cfg["in_channels"])
    s["fc0.bias"] = (cfg["dim_features"],)
    _bn(s, "bn0", cfg["dim_features"])
    d_in = cfg["dim_features"]
    for l in range(cfg["num_layers"]):
        d = cfg["dim_output"][l]
        p = "encoder.%d." % l
        _conv(s, p + "mlp1", d_in, d // 2)
        _conv(s, p + "lse1.mlp", 10, d // 2)
        s[p + "pool1.score_fn.0.weight"] = (d, d)
        s[p + "pool1.score_fn.0.bias"] = (d,)
        _conv(s, p + "pool1.mlp", d, d // 2)
        _conv(s, p + "lse2.mlp", d // 2, d // 2)
        s[p + "pool2.score_fn.0.weight"] = (d, d)
        s[p + "pool2.score_fn.0.bias"] = (d,)
        _conv(s, p + "pool2.mlp", d, d)
        _conv(s, p + "mlp2", d, 2 * d)
        _conv(s, p + "shortcut", d_in, 2 * d)
        d_in = 2 * d
    _conv(s, "mlp", d_in, d_in)
    ed = encoder_dims(cfg)
    prev = d_in
    for i in range(cfg["num_layers"]):
        skip = ed[-i - 2]
        _conv(s, "decoder.%d" % i, skip + prev, skip, transpose=True)
        prev = skip
    _conv(s, "fc1.0", prev, 64)
    _conv(s, "fc1.1", 64, 32)
    _conv(s, "fc1.3", 32, cfg["num_classes"], bn=False)
    return s


def randlanet_state_dict(cfg, seed):
    """Deterministic pseudo-trained weights (numpy Generator, independent of torch's RNG):
    He-style conv/linear weights, non-trivial BatchNorm affine + running statistics."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if name.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif ".batch_norm.weight" in name or name == "bn0.weight":
            v = rng.uniform(0.7, 1.3, shape)
        elif name.endswith(".bias"):
            v = rng.normal(0.0, 0.05, shape)
        else:
            if len(shape) == 4:
                fan_in = shape[0] if name.startswith("decoder.") else shape[1]
            else:
                fan_in = shape[1]
            v = rng.normal(0.0, 1.0, shape) * np.sqrt(1.2 / fan_in)
        sd[name] = torch.from_numpy(np.asarray(v, np.float32).reshape(shape))
    return sd



# ---------------------------------------------------------------------------------------------------
# KPConv
# ---------------------------------------------------------------------------------------------------
TORONTO3D_CFG = dict(   # ml3d/configs/kpconv_toronto3d.yml:23-82 (inference-relevant keys)
    KP_extent=1.0, KP_influence="linear", aggregation_mode="sum",
    architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb_strided", "resnetb",
                  "resnetb_strided", "resnetb", "resnetb_strided", "resnetb", "nearest_upsample", "unary",
                  "nearest_upsample", "unary", "nearest_upsample", "unary", "nearest_upsample", "unary"],
    batch_limit=10000, batch_norm_momentum=0.98, conv_radius=2.5, first_features_dim=128,
    first_subsampling_dl=0.08, fixed_kernel_points="center", in_features_dim=1, in_points_dim=3, in_radius=4.0,
    lbl_values=[0, 1, 2, 3, 4, 5, 6, 7, 8], ignored_label_inds=[0], max_in_points=10000, modulated=False,
    num_classes=8, num_kernel_points=15, num_layers=5, use_batch_norm=True, reduce_fc=True, l_relu=0.2)


PARISLILLE3D_CFG = dict(   # ml3d/configs/kpconv_parislille3d.yml:17-82 (inference-relevant keys): five deformable blocks
    KP_extent=1.0, KP_influence="linear", aggregation_mode="sum",
    architecture=["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb_strided", "resnetb_deformable",
                  "resnetb_deformable_strided", "resnetb_deformable", "resnetb_deformable_strided", "resnetb_deformable",
                  "nearest_upsample", "unary", "nearest_upsample", "unary", "nearest_upsample", "unary", "nearest_upsample",
                  "unary"],
    batch_limit=20000, batch_norm_momentum=0.98, conv_radius=2.5, deform_radius=6.0, first_features_dim=128,
    first_subsampling_dl=0.08, fixed_kernel_points="center", in_features_dim=1, in_points_dim=3, in_radius=4.0,
    lbl_values=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9], ignored_label_inds=[0], max_in_points=17000, modulated=False,
    num_classes=9, num_kernel_points=15, num_layers=5, use_batch_norm=True, reduce_fc=True, l_relu=0.2)

# a three-layer architecture with the deformable blocks of kpconv_parislille3d.yml:28-32 (deformable KPConv 32 -> 32 at a
# full and a strided block, 64 -> 64 on the coarsest layer); deform_radius below the YAML's 6.0 keeps the test batches small
KPCONV_DEFORM_SMALL_CFG = dict(
    KP_extent=1.2, KP_influence="linear", aggregation_mode="sum",
    architecture=["simple", "resnetb", "resnetb_strided", "resnetb_deformable", "resnetb_deformable_strided",
                  "resnetb_deformable", "nearest_upsample", "unary", "nearest_upsample", "unary"],
    batch_limit=10000, batch_norm_momentum=0.98, conv_radius=2.5, deform_radius=5.0, first_features_dim=64,
    first_subsampling_dl=0.08, fixed_kernel_points="center", in_features_dim=1, in_points_dim=3, in_radius=2.0,
    lbl_values=[0, 1, 2, 3, 4, 5], ignored_label_inds=[0], max_in_points=10000, modulated=False,
    num_classes=5, num_kernel_points=15, num_layers=3, use_batch_norm=True, reduce_fc=False, l_relu=0.1)


# ---------------------------------------------------------------------------------------------------
# architecture walk (kpconv.py:131-236) -> flat list of block descriptors
# ---------------------------------------------------------------------------------------------------
def arch_plan(cfg):
    """Mirrors the two loops of KPFCNN.__init__: returns (encoder, decoder, head) lists of dicts
    {name, kind, layer, in_dim, out_dim, radius, extent}, plus encoder_skips / decoder_concats."""
    arch = cfg["architecture"]
    layer, r = 0, cfg["first_subsampling_dl"] * cfg["conv_radius"]
    in_dim, out_dim = cfg["in_features_dim"], cfg["first_features_dim"]
    enc, skips, skip_dims = [], [], []
    for bi, block in enumerate(arch):
        if any(t in block for t in ("pool", "strided", "upsample", "global")):
            skips.append(bi)
            skip_dims.append(in_dim)
        if "upsample" in block:
            break
        enc.append(dict(name=block, layer=layer, in_dim=in_dim, out_dim=out_dim, radius=r,
                        extent=r * cfg["KP_extent"] / cfg["conv_radius"]))
        in_dim = out_dim // 2 if "simple" in block else out_dim
        if "pool" in block or "strided" in block:
            layer += 1
            r *= 2
            out_dim *= 2
    dec, concats = [], []
    start = next((i for i, b in enumerate(arch) if "upsample" in b), len(arch))
    for bi, block in enumerate(arch[start:]):
        if bi > 0 and "upsample" in arch[start + bi - 1]:
            in_dim += skip_dims[layer]
            concats.append(bi)
        dec.append(dict(name=block, layer=layer, in_dim=in_dim, out_dim=out_dim, radius=r,
                        extent=r * cfg["KP_extent"] / cfg["conv_radius"]))
        in_dim = out_dim
        if bi == 0 and cfg.get("reduce_fc", False):
            out_dim = out_dim // 2
        if "upsample" in block:
            layer -= 1
            r *= 0.5
            out_dim = out_dim // 2
    C = len(cfg["lbl_values"]) - len(cfg["ignored_label_inds"])
    if cfg.get("reduce_fc", False):
        head = [dict(in_dim=out_dim, out_dim=cfg["first_features_dim"] // 2, bn=True, relu=True),
                dict(in_dim=cfg["first_features_dim"] // 2, out_dim=C, bn=False, relu=False)]
    else:
        head = [dict(in_dim=out_dim, out_dim=cfg["first_features_dim"], bn=False, relu=True),
                dict(in_dim=cfg["first_features_dim"], out_dim=C, bn=False, relu=True)]
    return dict(encoder=enc, decoder=dec, head=head, encoder_skips=skips, decoder_concats=concats)


def synthetic_kernel_points(radius, K=15):
    """Deterministic stand-in for load_kernels (kpconv.py:1909-1999; the reference optimises a random
    disposition and caches it on disk): centre point + K-1 points on a Fibonacci sphere of 0.66 * radius."""
    pts = [[0.0, 0.0, 0.0]]
    n = K - 1
    for i in range(n):
        z = 1 - 2 * (i + 0.5) / n
        rr = np.sqrt(max(0.0, 1 - z * z))
        ph = i * np.pi * (3 - np.sqrt(5))
        pts.append([0.66 * rr * np.cos(ph), 0.66 * rr * np.sin(ph), 0.66 * z])
    return (np.asarray(pts) * radius).astype(np.float32)


def kpconv_state_dict(cfg, seed):
    """Pseudo-trained weights with the reference's state_dict keys and shapes."""
    g = torch.Generator().manual_seed(int(seed))
    plan = arch_plan(cfg)
    sd = {}
    K = cfg["num_kernel_points"]

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * scale

    def bn(prefix, c):
        if cfg["use_batch_norm"]:
            sd[prefix + ".batch_norm.weight"] = 1 + rnd(c, scale=0.3)
            sd[prefix + ".batch_norm.bias"] = rnd(c, scale=0.3)
            sd[prefix + ".batch_norm.running_mean"] = rnd(c, scale=0.2)
            sd[prefix + ".batch_norm.running_var"] = 0.5 + torch.rand(c, generator=g)
            sd[prefix + ".batch_norm.num_batches_tracked"] = torch.tensor(100)
        else:
            sd[prefix + ".bias"] = rnd(c, scale=0.3)

    def unary(prefix, cin, cout, use_bn=True):
        sd[prefix + ".mlp.weight"] = rnd(cout, cin, scale=(3.0 / cin) ** 0.5)
        if use_bn and cfg["use_batch_norm"]:
            bn(prefix + ".batch_norm", cout)
        else:
            sd[prefix + ".batch_norm.bias"] = rnd(cout, scale=0.3)

    def kpconv(prefix, cin, cout, radius, deformable=False):
        sd[prefix + ".weights"] = rnd(K, cin, cout, scale=(3.0 / (cin * 4.0)) ** 0.5)
        sd[prefix + ".kernel_points"] = torch.from_numpy(synthetic_kernel_points(radius, K))
        if deformable:
            # kpconv.py:948-965: an inner rigid KPConv cin -> 3 K (+ K modulations) and a bias; `kernel_points` IS the inner
            # convolution's parameter (kpconv.py:977-978: both names in the state dict).  Offsets of a few tenths of the extent.
            od = (4 if cfg.get("modulated", False) else 3) * K
            sd[prefix + ".offset_conv.weights"] = rnd(K, cin, od, scale=0.35 * (3.0 / (cin * 4.0)) ** 0.5)
            sd[prefix + ".offset_conv.kernel_points"] = sd[prefix + ".kernel_points"].clone()
            sd[prefix + ".offset_bias"] = rnd(od, scale=0.1)

    for i, b in enumerate(plan["encoder"]):
        p = "encoder_blocks.%d" % i
        if "simple" in b["name"]:
            kpconv(p + ".KPConv", b["in_dim"], b["out_dim"] // 2, b["radius"], "deform" in b["name"])
            bn(p + ".batch_norm", b["out_dim"] // 2)
        elif "resnetb" in b["name"]:
            mid = b["out_dim"] // 4
            if b["in_dim"] != mid:
                unary(p + ".unary1", b["in_dim"], mid)
            kpconv(p + ".KPConv", mid, mid, b["radius"], "deform" in b["name"])
            bn(p + ".batch_norm_conv", mid)
            unary(p + ".unary2", mid, b["out_dim"])
            if b["in_dim"] != b["out_dim"]:
                unary(p + ".unary_shortcut", b["in_dim"], b["out_dim"])
        else:
            raise NotImplementedError(b["name"])
    for i, b in enumerate(plan["decoder"]):
        if b["name"] == "unary":
            unary("decoder_blocks.%d" % i, b["in_dim"], b["out_dim"])
    h0, h1 = plan["head"]
    unary("head_mlp", h0["in_dim"], h0["out_dim"], use_bn=h0["bn"])
    unary("head_softmax", h1["in_dim"], h1["out_dim"], use_bn=h1["bn"])
    return sd



# ---------------------------------------------------------------------------------------------------
# PointPillars
# ---------------------------------------------------------------------------------------------------
POINTPILLARS_KITTI_CFG = dict(   # ml3d/configs/pointpillars_kitti.yml:7-66 (inference-relevant keys)
    point_cloud_range=[0, -39.68, -3, 69.12, 39.68, 1], classes=["Pedestrian", "Cyclist", "Car"],
    voxelize=dict(max_num_points=32, voxel_size=[0.16, 0.16, 4], max_voxels=[16000, 40000]),
    voxel_encoder=dict(in_channels=4, feat_channels=[64], voxel_size=[0.16, 0.16, 4]),
    scatter=dict(in_channels=64, output_shape=[496, 432]),
    backbone=dict(in_channels=64, out_channels=[64, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2]),
    neck=dict(in_channels=[64, 128, 256], out_channels=[128, 128, 128], upsample_strides=[1, 2, 4],
              use_conv_for_no_stride=False),
    head=dict(in_channels=384, feat_channels=384, nms_pre=100, score_thr=0.1,
              ranges=[[0, -39.68, -0.6, 70.4, 39.68, -0.6], [0, -39.68, -0.6, 70.4, 39.68, -0.6],
                      [0, -39.68, -1.78, 70.4, 39.68, -1.78]],
              sizes=[[0.6, 0.8, 1.73], [0.6, 1.76, 1.73], [1.6, 3.9, 1.56]], rotations=[0, 1.57],
              iou_thr=[[0.35, 0.5], [0.35, 0.5], [0.45, 0.6]]))

# a small two-PFN-layer, 3-channel-point variant (the argoverse / nuscenes shape family) on a 64 x 48 canvas
POINTPILLARS_SMALL_CFG = dict(
    point_cloud_range=[0, -9.6, -3, 25.6, 9.6, 1], classes=["Car", "Pedestrian"],
    voxelize=dict(max_num_points=20, voxel_size=[0.4, 0.4, 4], max_voxels=[3000, 3000]),
    voxel_encoder=dict(in_channels=3, feat_channels=[64, 64], voxel_size=[0.4, 0.4, 4]),
    scatter=dict(in_channels=64, output_shape=[48, 64]),
    backbone=dict(in_channels=64, out_channels=[32, 64, 96], layer_nums=[1, 2, 1], layer_strides=[1, 2, 2]),
    neck=dict(in_channels=[32, 64, 96], out_channels=[32, 32, 32], upsample_strides=[1, 2, 4],
              use_conv_for_no_stride=False),
    head=dict(in_channels=96, feat_channels=96, nms_pre=100, score_thr=0.1,
              ranges=[[0, -9.6, -0.6, 25.6, 9.6, -0.6], [0, -9.6, -1.78, 25.6, 9.6, -1.78]],
              sizes=[[0.6, 0.8, 1.73], [1.6, 3.9, 1.56]], rotations=[0, 1.57], iou_thr=[[0.35, 0.5], [0.45, 0.6]]))


def pointpillars_state_dict(cfg, seed):
    """Pseudo-trained weights with the reference's state_dict keys and shapes."""
    g = torch.Generator().manual_seed(int(seed))
    sd = {}

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * scale

    def bn(prefix, c):
        sd[prefix + ".weight"] = 1 + rnd(c, scale=0.3)
        sd[prefix + ".bias"] = rnd(c, scale=0.3)
        sd[prefix + ".running_mean"] = rnd(c, scale=0.2)
        sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(100)

    ve = cfg["voxel_encoder"]
    chans = [ve["in_channels"] + 5] + list(ve["feat_channels"])
    for i in range(len(chans) - 1):
        last = i == len(chans) - 2
        units = chans[i + 1] if last else chans[i + 1] // 2
        sd["voxel_encoder.pfn_layers.%d.linear.weight" % i] = rnd(units, chans[i], scale=(3.0 / chans[i]) ** 0.5)
        bn("voxel_encoder.pfn_layers.%d.norm" % i, units)
    bb = cfg["backbone"]
    cin = [bb["in_channels"]] + list(bb["out_channels"][:-1])
    for i, ln in enumerate(bb["layer_nums"]):
        co = bb["out_channels"][i]
        sd["backbone.blocks.%d.0.weight" % i] = rnd(co, cin[i], 3, 3, scale=(3.0 / (9 * cin[i])) ** 0.5 * 1.4)
        bn("backbone.blocks.%d.1" % i, co)
        for j in range(ln):
            sd["backbone.blocks.%d.%d.weight" % (i, 3 + 3 * j)] = rnd(co, co, 3, 3, scale=(3.0 / (9 * co)) ** 0.5 * 1.4)
            bn("backbone.blocks.%d.%d" % (i, 4 + 3 * j), co)
    nk = cfg["neck"]
    for i, co in enumerate(nk["out_channels"]):
        s = nk["upsample_strides"][i]
        ci = nk["in_channels"][i]
        sd["neck.deblocks.%d.0.weight" % i] = rnd(ci, co, s, s, scale=(3.0 / ci) ** 0.5)     # ConvTranspose2d [Cin,Cout,k,k]
        bn("neck.deblocks.%d.1" % i, co)
    hd = cfg["head"]
    na = len(hd["sizes"]) * len(hd["rotations"])
    nc = len(cfg["classes"])
    fc = hd["feat_channels"]
    for name, co in (("conv_cls", na * nc), ("conv_reg", na * 7), ("conv_dir_cls", na * 2)):
        sd["bbox_head.%s.weight" % name] = rnd(co, fc, 1, 1, scale=(3.0 / fc) ** 0.5)
        sd["bbox_head.%s.bias" % name] = rnd(co, scale=0.5)
    return sd


def crop_for_cfg(sweep, cfg):
    """Keep the points of a synthetic sweep that fall inside the config's range (what ObjectRangeFilter /
    the dataset crop hands to the model); float32 [N, 3 + C]."""
    r = cfg["point_cloud_range"]
    c = cfg["voxel_encoder"]["in_channels"]
    m = np.all((sweep[:, :3] >= np.array(r[:3], np.float32)) & (sweep[:, :3] <= np.array(r[3:], np.float32)), 1)
    return np.ascontiguousarray(sweep[m][:, :c], np.float32)
