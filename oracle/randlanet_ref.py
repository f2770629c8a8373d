"""CPU ORACLE for the RandLA-Net inference forward (test infrastructure, not product code).

A functional PyTorch-CPU restatement of ``RandLANet.forward`` and its sub-modules
(ml3d/torch/models/randlanet.py:241-350, 471-692 in the reference), driven by a plain
state_dict with the reference's parameter names.  It keeps the reference's op order
(1x1 Conv2d on (B, C, N, K) tensors, eval-mode BatchNorm2d with eps 1e-6, softmax over
the K axis, gather-based neighbour lookup) so that it agrees with the real reference
module to float rounding; ``oracle/gen_golden.py`` checks exactly that against the
module imported from /root/reference and stores golden vectors under tests/golden/.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from synth_weights import encoder_dims, param_shapes, randlanet_state_dict as make_state_dict  # noqa: F401  (input generation)

BN_EPS = 1e-6  # randlanet.py:77,499


# --------------------------------------------------------------------------------------------
# functional restatement
# --------------------------------------------------------------------------------------------

def _shared_mlp(sd, name, x, act_slope=None, bn=True, transpose=False):
    """SharedMLP.forward (randlanet.py:503-518): 1x1 conv -> BN(eval) -> activation."""
    w, b = sd[name + ".conv.weight"], sd[name + ".conv.bias"]
    x = F.conv_transpose2d(x, w, b) if transpose else F.conv2d(x, w, b)
    if bn:
        p = name + ".batch_norm."
        x = F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                         sd[p + "bias"], False, 0.01, BN_EPS)
    if act_slope is not None:
        x = F.leaky_relu(x, act_slope)
    return x


def _gather_neighbor(feat_bnd, idx):
    """LocalSpatialEncoding.gather_neighbor (randlanet.py:533-552): (B,N,d),(B,N,K)->(B,d,N,K)."""
    B, N, K = idx.shape
    d = feat_bnd.shape[2]
    ext_idx = idx.unsqueeze(1).expand(B, d, N, K)
    ext = feat_bnd.transpose(-2, -1).unsqueeze(-1).expand(B, d, N, K)
    return torch.gather(ext, 2, ext_idx)


def _lse(sd, name, coords, features, idx, relative_features=None):
    """LocalSpatialEncoding.forward (randlanet.py:554-605)."""
    B, N, K = idx.shape
    if relative_features is None:
        nb = _gather_neighbor(coords, idx)
        ext = coords.transpose(-2, -1).unsqueeze(-1).expand(B, 3, N, K)
        rel = ext - nb
        dist = torch.sqrt(torch.sum(torch.square(rel), dim=1, keepdim=True))
        relative_features = torch.cat([dist, rel, ext, nb], dim=1)
    relative_features = _shared_mlp(sd, name + ".mlp", relative_features, 0.2)
    nbf = _gather_neighbor(features.transpose(1, 2).squeeze(3), idx)
    return torch.cat([nbf, relative_features], dim=1), relative_features


def _att_pool(sd, name, x):
    """AttentivePooling.forward (randlanet.py:622-639)."""
    s = F.linear(x.permute(0, 2, 3, 1), sd[name + ".score_fn.0.weight"], sd[name + ".score_fn.0.bias"])
    s = torch.softmax(s, dim=-2).permute(0, 3, 1, 2)
    feats = torch.sum(s * x, dim=-1, keepdim=True)
    return _shared_mlp(sd, name + ".mlp", feats, 0.2)


def _lfa(sd, name, coords, feat, idx):
    """LocalFeatureAggregation.forward (randlanet.py:667-692)."""
    x = _shared_mlp(sd, name + ".mlp1", feat, 0.2)
    x, nbf = _lse(sd, name + ".lse1", coords, x, idx)
    x = _att_pool(sd, name + ".pool1", x)
    x, _ = _lse(sd, name + ".lse2", coords, x, idx, relative_features=nbf)
    x = _att_pool(sd, name + ".pool2", x)
    return F.leaky_relu(_shared_mlp(sd, name + ".mlp2", x) + _shared_mlp(sd, name + ".shortcut", feat), 0.01)


def _random_sample(feature, pool_idx):
    """RandLANet.random_sample (randlanet.py:300-327)."""
    feature = feature.squeeze(3)
    B, d = feature.shape[0], feature.shape[1]
    K = pool_idx.shape[2]
    flat = pool_idx.reshape(B, -1).unsqueeze(2).expand(B, -1, d)
    g = torch.gather(feature.transpose(1, 2), 1, flat).reshape(B, -1, K, d)
    return g.max(dim=2, keepdim=True)[0].permute(0, 3, 1, 2)


def _nearest_interpolation(feature, interp_idx):
    """RandLANet.nearest_interpolation (randlanet.py:329-350)."""
    feature = feature.squeeze(3)
    d = feature.shape[1]
    B, up = interp_idx.shape[0], interp_idx.shape[1]
    idx = interp_idx.reshape(B, up).unsqueeze(1).expand(B, d, -1)
    return torch.gather(feature, 2, idx).unsqueeze(3)


@torch.no_grad()
def forward(sd, cfg, inputs):
    """RandLANet.forward (randlanet.py:241-298) -> scores (B, N, num_classes)."""
    feat = inputs["features"]
    coords, nidx = inputs["coords"], inputs["neighbor_indices"]
    sub, interp = inputs["sub_idx"], inputs["interp_idx"]
    feat = F.linear(feat, sd["fc0.weight"], sd["fc0.bias"]).transpose(-2, -1).unsqueeze(-1)
    feat = F.batch_norm(feat, sd["bn0.running_mean"], sd["bn0.running_var"], sd["bn0.weight"],
                        sd["bn0.bias"], False, 0.01, BN_EPS)
    feat = F.leaky_relu(feat, 0.2)
    enc_list = []
    for i in range(cfg["num_layers"]):
        enc = _lfa(sd, "encoder.%d" % i, coords[i], feat, nidx[i])
        samp = _random_sample(enc, sub[i])
        if i == 0:
            enc_list.append(enc.clone())
        enc_list.append(samp.clone())
        feat = samp
    feat = _shared_mlp(sd, "mlp", feat, 0.2)
    for i in range(cfg["num_layers"]):
        up = _nearest_interpolation(feat, interp[-i - 1])
        feat = _shared_mlp(sd, "decoder.%d" % i, torch.cat([enc_list[-i - 2], up], dim=1), 0.2, transpose=True)
    x = _shared_mlp(sd, "fc1.0", feat, 0.2)
    x = _shared_mlp(sd, "fc1.1", x, 0.2)
    x = _shared_mlp(sd, "fc1.3", x, None, bn=False)  # Dropout(0.5) is identity in eval
    return x.squeeze(3).transpose(1, 2)


def build_inputs(points_bn3, features, cfg, knn):
    """The neighbour pyramid of RandLANet.transform (randlanet.py:213-236) for a batch.

    ``knn(support, query, k) -> int32 [Nq, k]`` is the oracle's knn_search."""
    B = points_bn3.shape[0]
    coords, nbrs, pools, ups = [], [], [], []
    pcs = [points_bn3[b] for b in range(B)]
    for i in range(cfg["num_layers"]):
        n_sub = pcs[0].shape[0] // cfg["sub_sampling_ratio"][i]
        nb = [knn(pc, pc, cfg["num_neighbors"]) for pc in pcs]
        subs = [pc[:n_sub] for pc in pcs]
        up = [knn(s, pc, 1) for s, pc in zip(subs, pcs)]
        coords.append(torch.from_numpy(np.stack(pcs)))
        nbrs.append(torch.from_numpy(np.stack(nb).astype(np.int64)))
        pools.append(torch.from_numpy(np.stack([x[:n_sub] for x in nb]).astype(np.int64)))
        ups.append(torch.from_numpy(np.stack(up).astype(np.int64)))
        pcs = subs
    return {"coords": coords, "neighbor_indices": nbrs, "sub_idx": pools, "interp_idx": ups,
            "features": torch.from_numpy(np.ascontiguousarray(features))}
