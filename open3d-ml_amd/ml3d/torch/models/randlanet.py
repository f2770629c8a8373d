"""RandLA-Net (inference) on MI355X — host-side mirror of the reference model class.

Same constructor arguments, parameter names and state_dict layout as
``ml3d/torch/models/randlanet.py:17-113,471-692`` of the reference, so
``ml3d/configs/randlanet_*.yml`` and published ``.pth`` checkpoints load unchanged.
The module tree below only OWNS the parameters; ``forward`` does not run PyTorch ops —
it folds BatchNorm into packed weights once (``_randla_pack``) and calls the hand-written
HIP kernels through the C ABI (``ml3d.ops.randla_knn_pyramid`` + ``randla_forward``).
There is no CPU execution path: on a non-GPU device ``forward`` raises.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _abi
from ... import ops
from . import _randla_pack
from ._datapath import GpuSearchTree, preprocess_segmentation   # noqa: F401


class SharedMLP(nn.Module):
    """Parameter holder for conv1x1 + BatchNorm2d(eps=1e-6) (reference randlanet.py:471-518)."""

    def __init__(self, in_channels, out_channels, transpose=False, bn=True, activation_fn=None):
        super().__init__()
        conv = nn.ConvTranspose2d if transpose else nn.Conv2d
        self.conv = conv(in_channels, out_channels, kernel_size=1)
        self.batch_norm = nn.BatchNorm2d(out_channels, eps=1e-6, momentum=0.01) if bn else None
        self.activation_fn = activation_fn


class LocalSpatialEncoding(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.mlp = SharedMLP(dim_in, dim_out, activation_fn=nn.LeakyReLU(0.2))


class AttentivePooling(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.score_fn = nn.Sequential(nn.Linear(in_channels, in_channels), nn.Softmax(dim=-2))
        self.mlp = SharedMLP(in_channels, out_channels, activation_fn=nn.LeakyReLU(0.2))


class LocalFeatureAggregation(nn.Module):
    """Parameter layout of the reference block (randlanet.py:642-665)."""

    def __init__(self, d_in, d_out):
        super().__init__()
        self.mlp1 = SharedMLP(d_in, d_out // 2, activation_fn=nn.LeakyReLU(0.2))
        self.lse1 = LocalSpatialEncoding(10, d_out // 2)
        self.pool1 = AttentivePooling(d_out, d_out // 2)
        self.lse2 = LocalSpatialEncoding(d_out // 2, d_out // 2)
        self.pool2 = AttentivePooling(d_out, d_out)
        self.mlp2 = SharedMLP(d_out, 2 * d_out)
        self.shortcut = SharedMLP(d_in, 2 * d_out)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _mark_prefix(t):
    """``sub_idx`` as ``transform`` builds it -- the prefix rows of the level's neighbour matrix (randlanet.py:222-223) -- carries
    this mark so that ``forward`` can skip the device-draining comparison for exactly these tensors and nothing else."""
    t._ml3d_prefix_of_neighbors = True
    return t


class RandLANet(nn.Module):

    def __init__(self, name='RandLANet', num_neighbors=16, num_layers=4, num_points=4096 * 11,
                 num_classes=19, ignored_label_inds=[0], sub_sampling_ratio=[4, 4, 4, 4], in_channels=3,
                 dim_features=8, dim_output=[16, 64, 128, 256], grid_size=0.06, batcher='DefaultBatcher',
                 ckpt_path=None, augment={}, device='cuda', **kwargs):
        super().__init__()
        self.cfg = _Cfg(name=name, num_neighbors=num_neighbors, num_layers=num_layers, num_points=num_points,
                        num_classes=num_classes, ignored_label_inds=list(ignored_label_inds),
                        sub_sampling_ratio=list(sub_sampling_ratio), in_channels=in_channels,
                        dim_features=dim_features, dim_output=list(dim_output), grid_size=grid_size,
                        batcher=batcher, ckpt_path=ckpt_path, augment=augment, **kwargs)
        self.device = torch.device(device) if isinstance(device, str) else device
        self.rng = np.random.default_rng(kwargs.get('seed', None))
        cfg = self.cfg
        self.fc0 = nn.Linear(cfg.in_channels, cfg.dim_features)
        self.bn0 = nn.BatchNorm2d(cfg.dim_features, eps=1e-6, momentum=0.01)
        widths, d = [], cfg.dim_features
        blocks = []
        for i in range(cfg.num_layers):
            blocks.append(LocalFeatureAggregation(d, cfg.dim_output[i]))
            d = 2 * cfg.dim_output[i]
            widths += [d, d] if i == 0 else [d]
        self.encoder = nn.ModuleList(blocks)
        self.mlp = SharedMLP(d, d, activation_fn=nn.LeakyReLU(0.2))
        dec = []
        for i in range(cfg.num_layers):
            skip = widths[-i - 2]
            dec.append(SharedMLP(skip + d, skip, transpose=True, activation_fn=nn.LeakyReLU(0.2)))
            d = skip
        self.decoder = nn.ModuleList(dec)
        self.fc1 = nn.Sequential(SharedMLP(d, 64, activation_fn=nn.LeakyReLU(0.2)),
                                 SharedMLP(64, 32, activation_fn=nn.LeakyReLU(0.2)), nn.Dropout(0.5),
                                 SharedMLP(32, cfg.num_classes, bn=False))
        self._packed = None       # (device, params tensor)
        self._engines = {}
        # HIP graphs around the device patch loop's per-patch sequence and the batch-of-one forward (see _transform_device,
        # _forward_graphed).  OFF by default: measured on the MI355X / ROCm 7.2 (profiles/r05_latency_graphs_ab.log) a replay of
        # the ~60-node patch graph + the ~35-node forward graph is no faster than launching the kernels (1.69-1.71 against
        # 1.64 ms per frame at batch 1: the chain is bound by the GPU's dependent-dispatch latency, not by the host) and shows
        # 90 ms outliers.  ML3D_RANDLA_GRAPHS=1 / ``model.use_graphs = True`` turn them on; results are bit-identical either way
        # (tests/test_gpu_api.py).
        self.use_graphs = os.environ.get("ML3D_RANDLA_GRAPHS", "0") == "1"
        self.eval()

    # ---- packed weights ---------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def packed_params(self, device):
        """BatchNorm-folded weights in ABI slot order on `device` (cached; call
        ``invalidate_packed()`` after changing parameters in place)."""
        if self._packed is None or self._packed[0] != device:
            desc = _abi.make_desc(self.cfg, 1, max(self.cfg.num_points, 1))
            off = _abi.randla_param_offsets(_abi.get(), desc)
            buf = _randla_pack.pack(self.state_dict(), self.cfg, off)
            self._packed = (device, torch.from_numpy(buf).to(device))
        return self._packed[1]

    def invalidate_packed(self):
        self._packed = None

    # ---- inference ----------------------------------------------------------------------------------
    def neighbor_pyramid(self, points):
        """GPU replacement of the 8 CPU ``knn_search`` calls of ``transform`` (randlanet.py:218-229)."""
        return ops.randla_knn_pyramid(points, self.cfg.sub_sampling_ratio, self.cfg.num_neighbors)

    def forward(self, inputs):
        """inputs: the dict ``transform``/the batcher produce (randlanet.py:231-239).  ``coords[0]``
        [B,N,3] and ``features`` [B,N,C] are required; ``neighbor_indices``/``interp_idx`` are used
        when present (int64 from the reference's CPU transform is accepted), otherwise the pyramid is
        searched on the GPU.  Returns scores [B, N, num_classes] like the reference."""
        if self.training:
            return self._forward_train(inputs)
        dev = self.device
        _abi.require_gpu(dev, "RandLANet.forward")
        fast = self._forward_graphed(inputs)
        if fast is not None:
            return fast
        coords = inputs['coords'][0] if isinstance(inputs['coords'], (list, tuple)) else inputs['coords']
        pts = coords.to(dev, torch.float32).contiguous()
        feat = inputs['features'].to(dev, torch.float32).contiguous()
        if 'neighbor_indices' in inputs and 'interp_idx' in inputs:
            nbr = [t.to(dev).to(torch.int32).contiguous() for t in inputs['neighbor_indices']]
            itp = [t.to(dev).to(torch.int32).contiguous() for t in inputs['interp_idx']]
            if 'sub_idx' in inputs:
                # random_sample (randlanet.py:300-327) pools through sub_idx; the kernels take it as the PREFIX of the
                # level's neighbour matrix, which is what transform builds (randlanet.py:222-223).  Anything else is refused.
                for l, (sub, nb) in enumerate(zip(inputs['sub_idx'], nbr)):
                    if getattr(sub, '_ml3d_prefix_of_neighbors', False) and sub.shape[-2] <= nb.shape[-2]:
                        continue      # marked by this class's own transform (prefix views of the device lists; the mark survives
                        #               this package's batcher only): comparing would drain the GPU.  Any other sub_idx is compared
                    sub = sub.to(dev)
                    if sub.shape[-2] > nb.shape[-2] or not torch.equal(sub.to(torch.int32), nb[..., :sub.shape[-2], :]):
                        raise RuntimeError("RandLANet.forward: sub_idx[%d] is not the prefix of neighbor_indices[%d]" % (l, l))
        else:
            nbr, itp = self.neighbor_pyramid(pts)
        B, N, _ = pts.shape
        desc = _abi.make_desc(self.cfg, B, N)
        return ops.randla_forward(desc, self.packed_params(dev), feat, pts, nbr, itp)

    def _forward_graphed(self, inputs):
        """Batch-of-one forward of a patch that ``_transform_device`` produced (recognised by its arena): the patch arena is copied
        into the static inputs of a captured graph of ``ops.randla_forward`` (one copy kernel), the graph replayed, the static
        scores cloned.  None: not such a patch / graphs off -> the eager path."""
        st = getattr(self, '_dev_loop', None)
        feats = inputs.get('features') if isinstance(inputs, dict) else None
        mark = getattr(feats, '_ml3d_arena', None)
        if st is None or mark is None or not getattr(self, 'use_graphs', True) or st.get('fwd_failed') or st.get('layout') is None:
            return None
        arena, lay_id = mark
        lay, nbytes = st['layout']
        if lay_id != id(lay) or arena.numel() != nbytes or feats.dim() != 3 or feats.shape[0] != 1:
            return None
        # every tensor the kernels will read must be the arena's own view (a caller may have swapped list entries)
        base, L = arena.data_ptr(), int(self.cfg.num_layers)
        coords = inputs['coords'][0] if isinstance(inputs['coords'], (list, tuple)) else inputs['coords']
        try:
            ok = feats.data_ptr() == base + lay['feats'][0] and coords.data_ptr() == base + lay['pts'][0] and \
                all(inputs['neighbor_indices'][l].data_ptr() == base + lay['nbr%d' % l][0] and
                    inputs['interp_idx'][l].data_ptr() == base + lay['itp%d' % l][0] for l in range(L))
        except Exception:
            ok = False
        if not ok:
            return None
        dev = self.device
        fg = st.get('fwd_graph')
        params = self.packed_params(dev)
        if fg is not None and fg['params'] is not params:
            fg = None                                            # the weights were repacked: capture again
        if fg is None:
            try:
                with torch.cuda.device(dev):
                    a_in = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    a_in.copy_(arena)
                    v = self._arena_views(a_in, lay)
                    k = int(self.cfg.num_points)
                    desc = _abi.make_desc(self.cfg, 1, k)
                    nbr = [v['nbr%d' % l] for l in range(L)]
                    itp = [v['itp%d' % l] for l in range(L)]
                    scores = torch.empty((1, k, int(self.cfg.num_classes)), dtype=torch.float32, device=dev)
                    run = lambda: ops.randla_forward(desc, params, v['feats'][None], v['pts'][None], nbr, itp, out=scores)
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        run()
                    torch.cuda.current_stream().wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        run()
                fg = st['fwd_graph'] = dict(graph=graph, arena=a_in, scores=scores, params=params, desc=desc)
            except Exception as e:
                st['fwd_failed'] = "%s: %s" % (type(e).__name__, e)
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass
                return None
        fg['arena'].copy_(arena)
        fg['graph'].replay()
        return fg['scores'].clone()

    # ---- training forward -----------------------------------------------------------------------------------
    def train(self, mode=True):
        """(the folded-BatchNorm parameter pack of the fused inference kernels is stale once training touches the weights)"""
        self._packed = None
        return super().train(mode)

    def _forward_train(self, inputs):
        """``RandLANet.forward`` in TRAINING mode (randlanet.py:241-298 with :533-692), differentiable, on hand-written HIP in BOTH
        passes (``ML3D_TRAIN_OPS=hip``, the default; csrc/train.hip): point-major ``[B, N, C]`` tensors, every 1x1 (transposed)
        convolution as ``ops.LinearFunction``, BatchNorm2d(eps 1e-6) on the batch statistics + LeakyReLU as
        ``ops.BatchNormActFunction`` (NOT folded: training updates the statistics), each attentive pooling -- neighbour gather,
        concat with the encoded relative positions, score Linear, softmax over K, weighted sum -- as ONE fused
        ``ops.AttentionStageFunction`` (no ``[B, N, K, d]`` tensor in either pass), ``random_sample`` through
        ``ops.GatherMaxFunction``, ``nearest_interpolation`` through ``ops.GatherRowsFunction``, the neighbour pyramid from the HIP
        search when the dict carries none.  torch carries the graph, the concatenations, the residual add and Dropout(0.5) of fc1
        (live, like in the reference).  ``ML3D_TRAIN_OPS=torch`` keeps rounds 3-4's formulation (Linear / BatchNorm / gathers on
        torch's autograd around ``ops.AttentivePoolFunction`` / ``ops.GatherMaxFunction``) as the A/B side."""
        import os
        import torch.nn.functional as F
        cfg, dev = self.cfg, self.device
        _abi.require_gpu(dev, "RandLANet.forward (training)")
        hip = os.environ.get("ML3D_TRAIN_OPS", "hip").strip().lower() != "torch"
        coords = inputs['coords'][0] if isinstance(inputs['coords'], (list, tuple)) else inputs['coords']
        pts = coords.to(dev, torch.float32).contiguous()
        feat = inputs['features'].to(dev, torch.float32)
        if 'neighbor_indices' in inputs and 'interp_idx' in inputs:
            nbr = [t.to(dev).long() for t in inputs['neighbor_indices']]
            itp = [t.to(dev).long() for t in inputs['interp_idx']]
        else:
            a, b = self.neighbor_pyramid(pts)
            nbr, itp = [t.long() for t in a], [t.long() for t in b]
        nbr32 = [t.to(torch.int32).contiguous() for t in nbr]
        B = pts.shape[0]
        rows = torch.arange(B, device=dev)

        def lin(x, w, b):
            return ops.LinearFunction.apply(x, w, b) if hip else F.linear(x, w, b)

        def bn_act(bn, y, slope):
            if hip:
                return ops.batch_norm_act(y, bn, slope)
            shp = y.shape
            y = F.batch_norm(y.reshape(-1, shp[-1]), bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum,
                             bn.eps).reshape(shp)
            return y if slope is None else F.leaky_relu(y, slope)

        def shared(m, x):           # SharedMLP.forward (randlanet.py:503-518) on [..., Cin] rows
            w = m.conv.weight[:, :, 0, 0]
            w = w.t() if isinstance(m.conv, nn.ConvTranspose2d) else w           # ConvTranspose2d stores [Cin, Cout]
            y = lin(x, w, m.conv.bias)
            act = m.activation_fn
            slope = act.negative_slope if isinstance(act, nn.LeakyReLU) else None
            if m.batch_norm is not None:
                y = bn_act(m.batch_norm, y, slope)
                return y if (slope is not None or act is None) else act(y)
            return act(y) if act is not None else y

        def gather(x, idx):         # x [B, N, C], idx [B, M, K] -> [B, M, K, C]
            return x[rows[:, None, None], idx]

        def relative(xyz, idx):     # the 10 relative-position channels of LocalSpatialEncoding.forward (randlanet.py:575-594): no
            with torch.no_grad():   # parameter and no trained input behind them
                nb = gather(xyz, idx)
                ctr = xyz[:, :, None, :].expand_as(nb)
                d = ctr - nb
                return torch.cat([torch.sqrt((d * d).sum(-1, keepdim=True)), d, ctr, nb], -1)

        def stage(pool, f, enc, l):  # LocalSpatialEncoding's gather + concat, then AttentivePooling.forward (randlanet.py:596-639)
            sc = pool.score_fn[0]
            if hip and ops.attention_stage_supported(nbr[l].shape[-1], f.shape[-1], enc.shape[-1], f.shape[0] * f.shape[1]):
                pooled = ops.AttentionStageFunction.apply(f, enc, nbr32[l], sc.weight, sc.bias)
            else:
                x = torch.cat([gather(f, nbr[l]), enc], -1)
                pooled = ops.AttentivePoolFunction.apply(lin(x, sc.weight, sc.bias), x)
            return shared(pool.mlp, pooled)

        y = bn_act(self.bn0, lin(feat, self.fc0.weight, self.fc0.bias), 0.2)
        skips, n = [], pts.shape[1]
        for i, blk in enumerate(self.encoder):
            xyz = pts[:, :n]
            f1 = shared(blk.mlp1, y)
            enc = shared(blk.lse1.mlp, relative(xyz, nbr[i]))
            p1 = stage(blk.pool1, f1, enc, i)
            enc2 = shared(blk.lse2.mlp, enc)
            p2 = stage(blk.pool2, p1, enc2, i)
            e = F.leaky_relu(shared(blk.mlp2, p2) + shared(blk.shortcut, y), 0.01)
            n_sub = n // cfg.sub_sampling_ratio[i]
            sub = ops.GatherMaxFunction.apply(e, nbr32[i], n_sub)
            if i == 0:
                skips.append(e)
            skips.append(sub)
            y, n = sub, n_sub
        y = shared(self.mlp, y)
        for i in range(cfg.num_layers):
            sel = itp[-i - 1][:, :, 0]                                                       # nearest_interpolation
            if hip:
                nc = y.shape[1]
                flat = (sel + rows[:, None] * nc).to(torch.int32).reshape(-1).contiguous()
                up = ops.GatherRowsFunction.apply(y.reshape(B * nc, y.shape[2]), flat).reshape(B, sel.shape[1], y.shape[2])
            else:
                up = y[rows[:, None], sel]
            y = shared(self.decoder[i], torch.cat([skips[-i - 2], up], -1))
        y = shared(self.fc1[0], y)
        y = shared(self.fc1[1], y)
        y = self.fc1[2](y)
        return shared(self.fc1[3], y)

    # ---- the reference's data path around forward (randlanet.py:115-239, 382-465), on the GPU ops ------------------
    def preprocess(self, data, attr):
        """Grid-subsample the raw cloud (barycentres, mean features, majority labels), build the search structure and,
        for test splits, project every raw point onto its nearest sub-cloud point (randlanet.py:115-154)."""
        return preprocess_segmentation(data, attr, self.cfg.grid_size, self.device)

    def _validation_augment(self, pc, feat):
        """The two augmentations the reference applies to EVERY split (randlanet.py:185-191;
        ml3d/datasets/augment/augmentation.py:16-60): recenter and normalize.  In place, like the reference."""
        aug = self.cfg.get('augment', {}) or {}
        if 'recenter' in aug and aug['recenter']:
            dim = aug['recenter'].get('dim', [0, 1, 2])
            pc[:, dim] = pc[:, dim] - pc.mean(0)[dim]
        if 'normalize' in aug and aug['normalize']:
            c = aug['normalize']
            if 'points' in c:
                if c['points'].get('method', 'linear') != 'linear':
                    raise ValueError(f"Unsupported method : {c['points'].get('method')}")
                pc -= pc.mean(0)
                pc /= (pc.max(0) - pc.min(0)).max()
            if 'feat' in c and feat is not None:
                cf = c['feat']
                if cf.get('method', 'linear') != 'linear':
                    raise ValueError(f"Unsupported method : {cf.get('method')}")
                bias, scale = cf.get('bias', 0), cf.get('scale', 1)
                feat -= bias
                feat /= scale

    def _training_augment(self, pc):
        """The augmentations the in-scope YAMLs add for the TRAINING split (randlanet.py:198-203 ->
        ml3d/datasets/augment/augmentation.py:68-146, 375-382), in the reference's order and with its draws from the model's
        generator (``default_rng(generator)`` hands the SAME generator back, so the pipeline's seed decides them): a yaw
        rotation by ``2 pi u``, an isotropic scale ``u (max_s - min_s) + min_s``, Gaussian jitter ``noise_std N(0, 1)`` --
        float32 like the reference's.  Anything else the reference's augmenter knows (dropout, flips, colour jitter) is refused."""
        aug = dict(self.cfg.get('augment', {}) or {})
        aug.pop('recenter', None)
        aug.pop('normalize', None)
        unknown = [k for k in aug if k not in ('rotate', 'scale', 'noise')]
        if unknown:
            raise NotImplementedError("RandLANet (MI355X build): training augmentation %s is not implemented" % unknown)
        rng = self.rng
        if 'rotate' in aug:
            method = (aug['rotate'] or {}).get('method', 'vertical')
            if method != 'vertical':
                raise NotImplementedError("RandLANet (MI355X build): rotate.method %r (only 'vertical')" % method)
            theta = rng.random() * 2 * np.pi
            c, s_ = np.cos(theta), np.sin(theta)
            pc = np.matmul(pc, np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=np.float32))
        if 'scale' in aug:
            c = aug['scale'] or {}
            lo, hi = c.get('min_s', 1.), c.get('max_s', 1.)
            factor = rng.random(pc.shape[1]) * (hi - lo) + lo if c.get('scale_anisotropic', False) else rng.random() * (hi - lo) + lo
            pc = pc * factor
        if 'noise' in aug:
            std = (aug['noise'] or {}).get('noise_std', 0.001)
            pc = pc + (rng.standard_normal((pc.shape[0], pc.shape[1])) * std).astype(np.float32)
        return pc

    def transform(self, data, attr, min_possibility_idx=None):
        """Patch crop by the pipeline's point sampler, recentring, feature assembly and the neighbour pyramid
        (randlanet.py:156-239).  The 8 ``knn_search`` calls of the reference run as ONE GPU pyramid call; the index lists
        come back as int32 DEVICE tensors (``forward`` consumes them in place; ``.cpu().long()`` gives the reference's
        arrays), everything else as numpy like the reference."""
        cfg = self.cfg
        training = attr['split'] in ['training', 'train']
        sampler = getattr(self, 'trans_point_sampler', None)
        if sampler is None:
            raise RuntimeError("RandLANet.transform: set model.trans_point_sampler (the pipeline takes it from the dataset "
                               "split's sampler, semantic_segmentation.py:156) or use inference_begin()")
        if not training and self._device_loop_serves(data, sampler):
            return self._transform_device()
        pc = data['point'].copy()
        label = data['label'].copy()
        feat = data['feat'].copy() if data['feat'] is not None else None
        tree = data['search_tree']
        pc, selected_idxs, center_point = sampler(pc=pc, feat=feat, label=label, search_tree=tree,
                                                  num_points=cfg.num_points)
        label = label[selected_idxs]
        if feat is not None:
            feat = feat[selected_idxs]
        self._validation_augment(pc, feat)
        if training:
            pc = self._training_augment(pc)
        feat = pc.copy() if feat is None else np.concatenate([pc, feat], axis=1)
        if cfg.in_channels != feat.shape[1]:
            raise RuntimeError("Wrong feature dimension, please update in_channels(3 + feature_dimension) in config")
        dpts = torch.from_numpy(np.ascontiguousarray(pc, dtype=np.float32)).to(self.device)
        nbr, itp = ops.randla_knn_pyramid(dpts[None], cfg.sub_sampling_ratio, cfg.num_neighbors)
        inputs = dict()
        coords, n = [], pc.shape[0]
        for i in range(cfg.num_layers):
            coords.append(pc[:n])
            n = n // cfg.sub_sampling_ratio[i]
        inputs['coords'] = coords
        inputs['neighbor_indices'] = [t[0] for t in nbr]
        inputs['sub_idx'] = [_mark_prefix(nbr[i][0, :pc.shape[0] // int(np.prod(cfg.sub_sampling_ratio[:i + 1]))])
                             for i in range(cfg.num_layers)]
        inputs['interp_idx'] = [t[0] for t in itp]
        inputs['features'] = feat
        inputs['point_inds'] = selected_idxs
        inputs['labels'] = label.astype(np.int64)
        return inputs

    def update_probs(self, inputs, results, test_probs):
        """Vote accumulation of ``semantic_segmentation.py:293-297`` / randlanet.py:441-465: batch items applied IN ORDER
        (a point that two patches of a batch share is smoothed twice), float16 accumulator, softmax + update on the GPU
        (``ml3d_vote_update``).  ``test_probs``: numpy float16 [num_points, classes] (returned updated, like the
        reference) or a float16 device tensor (updated in place and returned: no host round trip per patch)."""
        self.test_smooth = 0.95
        dev = self.device
        on_host = isinstance(test_probs, np.ndarray)
        tp = torch.from_numpy(np.ascontiguousarray(test_probs)).to(dev) if on_host else test_probs
        inds_all = inputs['data']['point_inds']
        for b in range(results.size()[0]):
            logits = torch.reshape(results[b], (-1, self.cfg.num_classes)).to(dev, torch.float32).contiguous()
            if isinstance(inds_all[b], torch.Tensor) and inds_all[b].is_cuda:
                # the device patch loop's indices: the k nearest of a cloud of >= k points, distinct by construction
                ops.vote_update(tp, inds_all[b].reshape(-1).to(torch.int32), logits, self.test_smooth)
                continue
            inds = np.asarray(inds_all[b].cpu() if isinstance(inds_all[b], torch.Tensor) else inds_all[b]).reshape(-1)
            if inds.size > tp.shape[0]:
                # a cloud smaller than num_points: the sampler pads the patch with REPEATED points
                # (semseg_spatially_regular.py:80-84).  numpy's ``test_probs[inds] = f(test_probs[inds])`` reads every row's
                # OLD value first and the LAST occurrence's write wins; the kernel updates rows in place, one wave per listed
                # row, so it gets each point once -- its last occurrence.
                _, first_in_reversed = np.unique(inds[::-1], return_index=True)
                keep = np.sort(inds.size - 1 - first_in_reversed)
                inds = inds[keep]
                logits = logits[torch.from_numpy(keep).to(dev)].contiguous()
            ops.vote_update(tp, torch.as_tensor(inds, dtype=torch.int32).to(dev), logits, self.test_smooth)
        return tp.cpu().numpy() if on_host else tp

    # ---- legacy single-cloud API (randlanet.py:382-439): the pipeline does not use it, but it is part of BaseModel ----
    def inference_begin(self, data):
        self.test_smooth = 0.95
        attr = {'split': 'test'}
        self.inference_ori_data = data
        self.inference_data = self.preprocess(data, attr)
        self.inference_proj_inds = self.inference_data['proj_inds']
        num_points = self.inference_data['search_tree'].data.shape[0]
        self.possibility = self.rng.random(num_points) * 1e-3
        self.test_probs = torch.zeros((num_points, self.cfg.num_classes), dtype=torch.float16, device=self.device)
        if getattr(self, 'trans_point_sampler', None) is None:
            self.trans_point_sampler = self._possibility_sampler
        self._dev_loop = self._device_loop_state(self.inference_data)

    # ---- the same loop with the cloud, the possibilities and every patch RESIDENT on the device -------------------------------
    # The host loop above touches the whole patch per step (copies of the cloud, a 45 056-index read-back, numpy crops, an
    # upload): ~3 ms of host work around ~0.5 ms of kernels.  With the model's own sampler nothing of that needs the host: the
    # centre is an argmin ON the device, the crop / distances / possibility bump / recentring are kernels that reproduce numpy's
    # arithmetic and ORDER (ml3d_patch_crop, ml3d_patch_recenter), and the only host input per patch is the shuffle of 0..k-1,
    # which depends on no data (``rng.permutation(idxs) == idxs[rng.permutation(k)]``: same draws) and is uploaded.  Patches are
    # identical to the host loop's, index for index (tests/test_gpu_api.py).
    def _device_loop_state(self, data):
        cfg = self.cfg
        aug = dict(cfg.get('augment', {}) or {})
        norm = dict(aug.get('normalize', None) or {})
        supported = set(aug) <= {'recenter', 'normalize', 'rotate', 'scale', 'noise'} and set(norm) <= {'feat'} and \
            (not norm or norm['feat'].get('method', 'linear') == 'linear')
        tree = data['search_tree']
        if not supported or not isinstance(tree, GpuSearchTree) or tree.data.shape[0] < cfg.num_points:
            return None
        dev = self.device
        feat = None if data['feat'] is None else torch.from_numpy(np.ascontiguousarray(data['feat'], dtype=np.float32)).to(dev)
        if cfg.in_channels != 3 + (0 if feat is None else feat.shape[1]):
            return None            # (the host path raises the reference's "Wrong feature dimension" error)
        rec = aug.get('recenter', None)
        nf = norm.get('feat', {}) if norm else {}
        return dict(points=tree._pts(), feat=feat, label=torch.from_numpy(np.ascontiguousarray(data['label'])).to(dev).long(),
                    possibility=torch.from_numpy(self.possibility).to(dev), data=data,
                    dims=tuple(rec.get('dim', [0, 1, 2])) if rec else (), bias=float(nf.get('bias', 0)), scale=float(nf.get('scale', 1)))

    def _device_loop_serves(self, data, sampler):
        st = getattr(self, '_dev_loop', None)
        return st is not None and data is st['data'] and sampler == self._possibility_sampler

    # ---- the patch arena: every per-patch tensor the loop hands to the batcher / forward, in ONE device buffer ------------------
    def _arena_layout(self):
        """name -> (byte offset, shape, dtype) of the per-patch tensors inside one uint8 arena, and its size: patch points, features,
        selected indices, labels, and per level the neighbour / interpolation index lists (the sub_idx lists are prefixes)."""
        cfg, st = self.cfg, self._dev_loop
        k, K = int(cfg.num_points), int(cfg.num_neighbors)
        c = 3 + (0 if st['feat'] is None else int(st['feat'].shape[1]))
        n = ops.pyramid_sizes(k, cfg.sub_sampling_ratio)
        items = [('pts', (k, 3), torch.float32), ('feats', (k, c), torch.float32), ('sel', (k,), torch.int32), ('labels', (k,), torch.int64)]
        for l in range(cfg.num_layers):
            items += [('nbr%d' % l, (1, n[l], K), torch.int32), ('itp%d' % l, (1, n[l], 1), torch.int32)]
        lay, off = {}, 0
        for name, shape, dt in items:
            nb = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            lay[name] = (off, shape, dt)
            off = (off + nb + 255) & ~255
        return lay, off

    @staticmethod
    def _arena_views(arena, lay):
        out = {}
        for name, (off, shape, dt) in lay.items():
            nb = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            out[name] = arena[off:off + nb].view(dt).view(*shape)
        return out

    def _patch_into(self, v, perm):
        """The device patch loop's step (centre argmin -> crop -> recentre -> neighbour pyramid -> labels) writing into the
        arena views ``v``; no host synchronisation, nothing allocated that outlives the call except through ``v``."""
        cfg, st = self.cfg, self._dev_loop
        k = int(cfg.num_points)
        center = torch.argmin(st['possibility']).reshape(1)
        ops.device_patch(st['points'], st['possibility'], center, perm, k, st['dims'], st['feat'], st['bias'], st['scale'],
                         out=(v['pts'], v['feats'], v['sel']))
        ops.randla_knn_pyramid(v['pts'][None], cfg.sub_sampling_ratio, cfg.num_neighbors,
                               out=([v['nbr%d' % l] for l in range(cfg.num_layers)], [v['itp%d' % l] for l in range(cfg.num_layers)]))
        torch.index_select(st['label'], 0, v['sel'].long(), out=v['labels'])      # (int64 labels cast once per cloud)

    def _inputs_from_arena(self, arena, lay):
        cfg = self.cfg
        k = int(cfg.num_points)
        v = self._arena_views(arena, lay)
        inputs, coords, n = dict(), [], k
        for i in range(cfg.num_layers):
            coords.append(v['pts'][:n])
            n = n // cfg.sub_sampling_ratio[i]
        nbr = [v['nbr%d' % l] for l in range(cfg.num_layers)]
        inputs['coords'] = coords
        inputs['neighbor_indices'] = [t[0] for t in nbr]
        inputs['sub_idx'] = [_mark_prefix(nbr[i][0, :k // int(np.prod(cfg.sub_sampling_ratio[:i + 1]))]) for i in range(cfg.num_layers)]
        inputs['interp_idx'] = [v['itp%d' % l][0] for l in range(cfg.num_layers)]
        inputs['features'] = v['feats']
        inputs['point_inds'] = v['sel']
        inputs['labels'] = v['labels']
        # the forward recognises a patch of this loop by its arena (``_stack`` of this package's batcher carries the mark through a
        # batch of one): one copy into the forward graph's static inputs instead of ~30 launches
        inputs['features']._ml3d_arena = (arena, id(lay))
        return inputs

    def _transform_device(self):
        """One patch of the device-resident loop.  With HIP graphs on (the default on a HIP device; ``self.use_graphs = False`` or
        a failed capture turn them off): the ~60 launches of the step are ONE graph replay that writes the patch into a static
        arena, and the caller gets a CLONE of that arena (one copy kernel) -- fresh tensors per patch, like the eager loop."""
        cfg, st, dev = self.cfg, self._dev_loop, self.device
        k = int(cfg.num_points)
        if st.get('layout') is None:
            st['layout'] = self._arena_layout()
        lay, nbytes = st['layout']
        perm_host = torch.from_numpy(self.rng.permutation(k).astype(np.int32))
        g = self._patch_graph(lay, nbytes) if getattr(self, 'use_graphs', True) else None
        if g is None:
            arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._patch_into(self._arena_views(arena, lay), perm_host.to(dev))
            return self._inputs_from_arena(arena, lay)
        # the shuffle goes through ONE pinned staging buffer: its previous upload has long finished, but the event makes that a
        # fact before the buffer is overwritten (an asynchronous copy out of pageable memory that is freed right after the call
        # faulted on the MI355X: "write access to a read-only page")
        g['uploaded'].synchronize()
        g['perm_pinned'].copy_(perm_host)
        g['perm'].copy_(g['perm_pinned'], non_blocking=True)
        g['uploaded'].record()
        g['graph'].replay()
        return self._inputs_from_arena(g['arena'].clone(), lay)

    def _patch_graph(self, lay, nbytes):
        st, dev = self._dev_loop, self.device
        if dev.type != 'cuda' or not hasattr(torch.cuda, 'CUDAGraph'):
            return None
        g = st.get('graph')
        if g is not None or st.get('graph_failed'):
            return g
        try:
            with torch.cuda.device(dev):
                arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                perm = torch.empty(int(self.cfg.num_points), dtype=torch.int32, device=dev)
                v = self._arena_views(arena, lay)
                # one eager pass on a side stream first (allocator pools, lazy module loading), WITHOUT touching the loop's state:
                # the possibilities are bumped by every crop, so they are saved and restored around warm-up and capture
                keep = st['possibility'].clone()
                perm.copy_(torch.arange(perm.numel(), dtype=torch.int32))
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._patch_into(v, perm)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._patch_into(v, perm)
                torch.cuda.current_stream().synchronize()
                st['possibility'].copy_(keep)
                uploaded = torch.cuda.Event()
                uploaded.record()
            g = st['graph'] = dict(graph=graph, arena=arena, perm=perm, uploaded=uploaded,
                                   perm_pinned=torch.empty(perm.numel(), dtype=torch.int32).pin_memory())
        except Exception as e:                                  # capture is an optimisation: the eager loop is always there
            st['graph_failed'] = "%s: %s" % (type(e).__name__, e)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            g = None
        return g

    def _min_possibility(self):
        st = getattr(self, '_dev_loop', None)
        return float(torch.min(st['possibility'])) if st is not None else float(np.min(self.possibility))

    def _possibility_sampler(self, pc, feat, label, search_tree, num_points):
        """Spatially regular patch sampler on ``self.possibility`` (the reference's
        ``SemSegSpatiallyRegularSampler._random_centered_gen``, semseg_spatially_regular.py:62-108, with this model's own
        seeded generator in place of the global ``random`` module)."""
        center_id = int(np.argmin(self.possibility))
        center_point = pc[center_id, :].reshape(1, -1)
        if pc.shape[0] < num_points:
            idxs = np.arange(pc.shape[0])
            idxs = np.concatenate([idxs, self.rng.choice(idxs, num_points - pc.shape[0])])
        else:
            idxs = search_tree.query(center_point, k=num_points)[1][0]
        idxs = self.rng.permutation(idxs)
        pc = pc[idxs]
        dists = np.sum(np.square((pc - center_point).astype(np.float32)), axis=1)
        delta = np.square(1 - dists / np.max(dists))
        self.possibility[idxs] += delta       # plain fancy +=, like the reference: a padded (repeated) point is bumped ONCE
        return pc, idxs, center_point

    def inference_preprocess(self):
        attr = {'split': 'test'}
        data = self.transform(self.inference_data, attr)
        batch = {k: ([t[None] if isinstance(t, torch.Tensor) else torch.as_tensor(t)[None] for t in v]
                     if isinstance(v, list) else torch.as_tensor(v)[None]) for k, v in data.items()}
        mark = getattr(data['features'], '_ml3d_arena', None)
        if mark is not None:
            batch['features']._ml3d_arena = mark      # (a batch of one device-loop patch: see _forward_graphed)
        self.inference_input = {'data': batch, 'attr': attr}
        return self.inference_input

    def inference_end(self, inputs, results):
        self.update_probs(inputs, results, self.test_probs)
        if self._min_possibility() > 0.5:
            probs = self.test_probs.cpu().numpy()
            pred_labels = np.argmax(probs, 1)[self.inference_proj_inds]
            self.inference_result = {'predict_labels': pred_labels, 'predict_scores': probs[self.inference_proj_inds]}
            return True
        return False

    def get_optimizer(self, cfg_pipeline):
        """randlanet.py:352-357."""
        optimizer = torch.optim.Adam(self.parameters(), **cfg_pipeline.optimizer)
        return optimizer, torch.optim.lr_scheduler.ExponentialLR(optimizer, cfg_pipeline.scheduler_gamma)

    def get_loss(self, Loss, results, inputs, device):
        """randlanet.py:359-380: class-weighted cross entropy (``Loss`` = the pipeline's ``SemSegLoss``) over the points whose
        label is not ignored.  Returns (loss, labels, scores)."""
        from ..modules import valid_scores_and_labels
        cfg = self.cfg
        scores, labels = valid_scores_and_labels(results, inputs['data']['labels'], cfg.num_classes, cfg.ignored_label_inds, device)
        return Loss.weighted_CrossEntropyLoss(scores, labels), labels, scores
