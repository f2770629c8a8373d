"""KPFCNN / KPConv (rigid and deformable blocks, inference) on MI355X — host-side mirror of the reference model.

Same constructor arguments, module/parameter names and state_dict layout as the reference
``ml3d/torch/models/kpconv.py:33-291``, so ``ml3d/configs/kpconv_*.yml`` and
published checkpoints load unchanged.  The module tree only OWNS parameters: ``forward`` folds eval-mode
BatchNorm into the weights once and runs the hand-written HIP kernels through the C ABI
(``ml3d.ops.kpconv_rigid`` / ``kpconv_deformable`` / ``linear`` / ``gather_pool``).  ``KPConvBatch`` builds the per-layer
points / neighbour / pool / upsample matrices of ``KPConvBatch.segmentation_inputs``
(ml3d/torch/dataloaders/concat_batcher.py:186-305) on the GPU (fixed-radius search + grid subsample).
Deformable blocks (kpconv_parislille3d.yml:28-32) run natively for ``KP_influence: linear`` (kpconv.py:1011-1159: inner
offset convolution, per-query kernel points, optional modulations); other deformable variants raise at construction.
There is no CPU execution path.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _abi
from ... import ops


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


# ---- parameter holders (names follow the reference classes) -----------------------------------------
class KPConv(nn.Module):
    """kpconv.py:891-1003: ``weights`` [K, cin, cout], ``kernel_points`` [K, 3] (not trained).  ``deformable``: the inner rigid
    convolution ``offset_conv`` (cin -> 3 K, + K when ``modulated``) and ``offset_bias``; ``kernel_points`` IS the inner
    convolution's parameter (kpconv.py:948-978: both names appear in the state dict)."""

    def __init__(self, kernel_size, in_channels, out_channels, KP_extent, radius, deformable=False, modulated=False):
        super().__init__()
        self.K, self.in_channels, self.out_channels = kernel_size, in_channels, out_channels
        self.KP_extent, self.radius = KP_extent, radius
        self.deformable, self.modulated = bool(deformable), bool(modulated)
        self.weights = nn.Parameter(torch.zeros((kernel_size, in_channels, out_channels), dtype=torch.float32))
        if self.deformable:
            self.offset_dim = (4 if self.modulated else 3) * kernel_size
            self.offset_conv = KPConv(kernel_size, in_channels, self.offset_dim, KP_extent, radius)
            self.offset_bias = nn.Parameter(torch.zeros(self.offset_dim, dtype=torch.float32))
            self.kernel_points = self.offset_conv.kernel_points
        else:
            self.offset_dim, self.offset_conv, self.offset_bias = None, None, None
            self.kernel_points = nn.Parameter(default_kernel_points(radius, kernel_size), requires_grad=False)
        nn.init.kaiming_uniform_(self.weights, a=5 ** 0.5)


class BatchNormBlock(nn.Module):
    """kpconv.py:1213-1254."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super().__init__()
        self.use_bn = use_bn
        if use_bn:
            self.batch_norm = nn.BatchNorm1d(in_dim, momentum=1 - bn_momentum)
        else:
            self.bias = nn.Parameter(torch.zeros(in_dim, dtype=torch.float32))


class UnaryBlock(nn.Module):
    """kpconv.py:1257-1300."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False, l_relu=0.1):
        super().__init__()
        self.in_dim, self.out_dim, self.no_relu, self.l_relu = in_dim, out_dim, no_relu, l_relu
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, use_bn, bn_momentum)


class SimpleBlock(nn.Module):
    """kpconv.py:1303-1358."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, cfg):
        super().__init__()
        self.block_name, self.layer_ind = block_name, layer_ind
        extent = radius * cfg.KP_extent / cfg.conv_radius
        self.KPConv = KPConv(cfg.num_kernel_points, in_dim, out_dim // 2, extent, radius,
                             deformable='deform' in block_name, modulated=cfg.get('modulated', False))
        self.batch_norm = BatchNormBlock(out_dim // 2, cfg.use_batch_norm, cfg.batch_norm_momentum)


class ResnetBottleneckBlock(nn.Module):
    """kpconv.py:1361-1461."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, cfg):
        super().__init__()
        self.block_name, self.layer_ind, self.in_dim, self.out_dim = block_name, layer_ind, in_dim, out_dim
        extent = radius * cfg.KP_extent / cfg.conv_radius
        bn, mom, lr = cfg.use_batch_norm, cfg.batch_norm_momentum, cfg.get('l_relu', 0.1)
        mid = out_dim // 4
        self.unary1 = UnaryBlock(in_dim, mid, bn, mom, l_relu=lr) if in_dim != mid else nn.Identity()
        self.KPConv = KPConv(cfg.num_kernel_points, mid, mid, extent, radius,
                             deformable='deform' in block_name, modulated=cfg.get('modulated', False))
        self.batch_norm_conv = BatchNormBlock(mid, bn, mom)
        self.unary2 = UnaryBlock(mid, out_dim, bn, mom, no_relu=True, l_relu=lr)
        self.unary_shortcut = UnaryBlock(in_dim, out_dim, bn, mom, no_relu=True, l_relu=lr) \
            if in_dim != out_dim else nn.Identity()


class NearestUpsampleBlock(nn.Module):
    """kpconv.py:1468-1481 (no parameters)."""

    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind


def default_kernel_points(radius, K=15):
    """Deterministic kernel disposition used for freshly constructed models: the centre plus K-1 points on a
    Fibonacci sphere of 0.66 * radius.  (The reference optimises a random disposition and caches it in the CWD,
    kpconv.py:1909-1999; checkpoints carry their own ``kernel_points``, which ``load_state_dict`` restores.)"""
    pts = [[0.0, 0.0, 0.0]]
    n = K - 1
    for i in range(n):
        z = 1 - 2 * (i + 0.5) / n
        rr = np.sqrt(max(0.0, 1 - z * z))
        ph = i * np.pi * (3 - np.sqrt(5))
        pts.append([0.66 * rr * np.cos(ph), 0.66 * rr * np.sin(ph), 0.66 * z])
    return torch.from_numpy((np.asarray(pts) * radius).astype(np.float32))


_DEFORMABLE_CIN = (16, 32, 64, 128, 256, 512)     # input widths ml3d_kpconv_deformable takes (the MFMA aggregation)


def _block_decider(block_name, radius, in_dim, out_dim, layer_ind, cfg):
    """kpconv.py:1171-1210: rigid and deformable blocks (the equivariant / invariant names of the reference's decider have
    no implementation there either)."""
    if 'equivariant' in block_name or 'invariant' in block_name:
        raise NotImplementedError("KPFCNN (MI355X build): block '%s' has no implementation" % block_name)
    if 'deformable' in block_name:
        cin = in_dim if 'simple' in block_name else out_dim // 4
        if cfg.KP_influence != 'linear' or cin not in _DEFORMABLE_CIN:
            raise NotImplementedError("KPFCNN (MI355X build): deformable block '%s' needs KP_influence='linear' and a KPConv "
                                      "input width in %s (got %s, %d).  With an Open3D-ML checkout (OPEN3D_ML_ROOT) the "
                                      "`open3d.ml.torch` registry falls back to the checkout's PyTorch KPFCNN for such "
                                      "configs" % (block_name, list(_DEFORMABLE_CIN), cfg.KP_influence, cin))
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, cfg.use_batch_norm, cfg.batch_norm_momentum, l_relu=cfg.get('l_relu', 0.1))
    if block_name in ('simple', 'simple_strided', 'simple_deformable', 'simple_deformable_strided'):
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, cfg)
    if block_name in ('resnetb', 'resnetb_strided', 'resnetb_deformable', 'resnetb_deformable_strided'):
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, cfg)
    if block_name == 'nearest_upsample':
        return NearestUpsampleBlock(layer_ind)
    raise NotImplementedError("KPFCNN (MI355X build): unsupported block '%s'" % block_name)


_INFLUENCE = {'constant': 0, 'linear': 1, 'gaussian': 2}


class KPFCNN(nn.Module):

    def __init__(self, name='KPFCNN', lbl_values=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19],
                 num_classes=19, ignored_label_inds=[0], ckpt_path=None, batcher='ConcatBatcher',
                 architecture=['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb',
                               'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb',
                               'nearest_upsample', 'unary', 'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
                               'nearest_upsample', 'unary'],
                 in_radius=4.0, max_in_points=100000, batch_num=8, batch_limit=30000, val_batch_num=8,
                 num_kernel_points=15, first_subsampling_dl=0.06, conv_radius=2.5, deform_radius=6.0, KP_extent=1.2,
                 KP_influence='linear', aggregation_mode='sum', first_features_dim=128, in_features_dim=2,
                 modulated=False, use_batch_norm=True, batch_norm_momentum=0.02, in_points_dim=3,
                 fixed_kernel_points='center', num_layers=5, l_relu=0.1, reduce_fc=False, device='cuda', **kwargs):
        super().__init__()
        cfg = _Cfg(name=name, lbl_values=list(lbl_values), num_classes=num_classes,
                   ignored_label_inds=list(ignored_label_inds), ckpt_path=ckpt_path, batcher=batcher,
                   architecture=list(architecture), in_radius=in_radius, max_in_points=max_in_points,
                   batch_num=batch_num, batch_limit=batch_limit, val_batch_num=val_batch_num,
                   num_kernel_points=num_kernel_points, first_subsampling_dl=first_subsampling_dl,
                   conv_radius=conv_radius, deform_radius=deform_radius, KP_extent=KP_extent, KP_influence=KP_influence,
                   aggregation_mode=aggregation_mode, first_features_dim=first_features_dim,
                   in_features_dim=in_features_dim, modulated=modulated, use_batch_norm=use_batch_norm,
                   batch_norm_momentum=batch_norm_momentum, in_points_dim=in_points_dim,
                   fixed_kernel_points=fixed_kernel_points, num_layers=num_layers, l_relu=l_relu, reduce_fc=reduce_fc,
                   **kwargs)
        self.cfg = cfg
        self.device = torch.device(device) if isinstance(device, str) else device
        if KP_influence not in _INFLUENCE or aggregation_mode != 'sum' or num_kernel_points != 15 or in_points_dim != 3:
            raise NotImplementedError("KPFCNN (MI355X build): KP_influence in %s, aggregation_mode='sum', 15 kernel "
                                      "points, 3-D points" % list(_INFLUENCE))
        # ---- encoder (kpconv.py:131-187) -----------------------------------------------------------------
        layer, r = 0, cfg.first_subsampling_dl * cfg.conv_radius
        in_dim, out_dim = cfg.in_features_dim, cfg.first_features_dim
        self.K = cfg.num_kernel_points
        self.C = len(cfg.lbl_values) - len(cfg.ignored_label_inds)
        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims, self.encoder_skips = [], []
        self.neighborhood_limits = []
        for block_i, block in enumerate(cfg.architecture):
            if any(t in block for t in ('pool', 'strided', 'upsample', 'global')):
                self.encoder_skips.append(block_i)
                self.encoder_skip_dims.append(in_dim)
            if 'upsample' in block:
                break
            if block == 'unary':
                # the reference's block_decider accepts it here; none of its configs has one and the fused encoder walk
                # below (KPConv block -> gather / GEMM epilogues) has no slot for it: refuse at construction, not at run time
                raise NotImplementedError("KPFCNN (MI355X build): a 'unary' block in the ENCODER is not supported "
                                          "(no reference config uses one)")
            self.encoder_blocks.append(_block_decider(block, r, in_dim, out_dim, layer, cfg))
            in_dim = out_dim // 2 if 'simple' in block else out_dim
            if 'pool' in block or 'strided' in block:
                layer += 1
                r *= 2
                out_dim *= 2
        # ---- decoder (kpconv.py:189-236) -----------------------------------------------------------------
        self.decoder_blocks = nn.ModuleList()
        self.decoder_concats = []
        start_i = next((i for i, b in enumerate(cfg.architecture) if 'upsample' in b), len(cfg.architecture))
        for block_i, block in enumerate(cfg.architecture[start_i:]):
            if block_i > 0 and 'upsample' in cfg.architecture[start_i + block_i - 1]:
                in_dim += self.encoder_skip_dims[layer]
                self.decoder_concats.append(block_i)
            self.decoder_blocks.append(_block_decider(block, r, in_dim, out_dim, layer, cfg))
            in_dim = out_dim
            if block_i == 0 and cfg.reduce_fc:
                out_dim = out_dim // 2
            if 'upsample' in block:
                layer -= 1
                r *= 0.5
                out_dim = out_dim // 2
        lr = cfg.get('l_relu', 0.1)
        if reduce_fc:
            self.head_mlp = UnaryBlock(out_dim, cfg.first_features_dim // 2, True, cfg.batch_norm_momentum, l_relu=lr)
            self.head_softmax = UnaryBlock(cfg.first_features_dim // 2, self.C, False, 1, no_relu=True, l_relu=lr)
        else:
            self.head_mlp = UnaryBlock(out_dim, cfg.first_features_dim, False, 0, l_relu=lr)
            self.head_softmax = UnaryBlock(cfg.first_features_dim, self.C, False, 0, l_relu=lr)
        self._packed = None
        self.eval()

    # ---- BatchNorm folding ---------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        """(the folded-BatchNorm parameter pack of the inference kernels is stale as soon as training touches the weights)"""
        self._packed = None
        return super().train(mode)

    def invalidate_packed(self):
        self._packed = None

    @staticmethod
    def _bn_affine(bnb, c):
        """BatchNormBlock (eval) as y = x * s + t, float64."""
        if bnb.use_bn:
            bn = bnb.batch_norm
            s = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
            t = bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * s
            return s, t
        return torch.ones(c, dtype=torch.float64), bnb.bias.detach().double().cpu()

    def _pack_unary(self, ub, dev):
        s, t = self._bn_affine(ub.batch_norm, ub.out_dim)
        wt = (ub.mlp.weight.detach().double().cpu() * s[:, None]).t().contiguous()      # [cin, cout]
        return dict(wt=wt.float().to(dev), b=t.float().to(dev), act=0 if ub.no_relu else 1, slope=ub.l_relu)

    def _pack_conv(self, conv, bnb, dev):
        s, t = self._bn_affine(bnb, conv.out_channels)
        w = conv.weights.detach().double().cpu() * s[None, None, :]
        d = dict(w=w.reshape(conv.K * conv.in_channels, conv.out_channels).float().contiguous().to(dev),
                 b=t.float().to(dev), kp=conv.kernel_points.detach().float().contiguous().to(dev),
                 extent=float(conv.KP_extent), ow=None, ob=None)
        if conv.deformable:        # the inner convolution: no batch norm, its bias is offset_bias (kpconv.py:1014-1016)
            d['ow'] = conv.offset_conv.weights.detach().float().reshape(conv.K * conv.in_channels, conv.offset_dim) \
                .contiguous().to(dev)
            d['ob'] = conv.offset_bias.detach().float().contiguous().to(dev)
        # the [15 cin, cout] contraction on the bf16 matrix pipe (ops.pack_bf16x3: exact three-way split, float32-equivalent);
        # cin >= 64: narrower convolutions run in the fused kernels, which keep the float matrix.  ML3D_KP_GEMM=f32 (A/B knob)
        # keeps the f32 MFMA kernel everywhere.  (The unary / shortcut Linears stay on gemm_tile: K = 96 .. 320 is three to ten chunks,
        # the 128-row kernel has nothing to pipeline there -- measured, forward 5.50 ms with only the contractions on bf16x3, 5.53 with
        # the dense Linears of the two finest levels as well, profiles/r05_kp_bf16x3_ab.log)
        d['packed'] = None
        if d['w'].is_cuda and conv.in_channels >= 64 and conv.in_channels % 32 == 0 and not conv.deformable and \
                os.environ.get("ML3D_KP_GEMM", "bf16x3") != "f32":
            d['packed'] = ops.pack_bf16x3(d['w'])
        return d

    def packed_params(self, dev):
        if self._packed is None or self._packed[0] != dev:
            P = dict(enc=[], dec=[])
            for blk in self.encoder_blocks:
                if isinstance(blk, SimpleBlock):
                    P['enc'].append(dict(conv=self._pack_conv(blk.KPConv, blk.batch_norm, dev)))
                else:
                    d = dict(conv=self._pack_conv(blk.KPConv, blk.batch_norm_conv, dev), u2=self._pack_unary(blk.unary2, dev))
                    d['u1'] = None if isinstance(blk.unary1, nn.Identity) else self._pack_unary(blk.unary1, dev)
                    d['sc'] = None if isinstance(blk.unary_shortcut, nn.Identity) else \
                        self._pack_unary(blk.unary_shortcut, dev)
                    # unary2 and the shortcut Linear are both activation-free and meet in one sum (kpconv.py:1451-1461):
                    # [y | shortcut_in] . [W_u2 ; W_sc] + (b_u2 + b_sc) is ONE GEMM over the concatenated K -- the shortcut's
                    # [N, out] output is neither written nor read back as a residual (656 MB per block on the full-resolution layer)
                    d['u2sc'] = None
                    if d['sc'] is not None and self._FUSE_SHORTCUT and d['sc']['act'] == 0 and d['u2']['act'] == 0:
                        d['u2sc'] = dict(wt=torch.cat([d['u2']['wt'], d['sc']['wt']], 0).contiguous(),
                                         b=(d['u2']['b'] + d['sc']['b']).contiguous(), act=0, slope=0.0)
                    P['enc'].append(d)
            for blk in self.decoder_blocks:
                P['dec'].append(self._pack_unary(blk, dev) if isinstance(blk, UnaryBlock) else None)
            P['head'] = [self._pack_unary(self.head_mlp, dev), self._pack_unary(self.head_softmax, dev)]
            self._packed = (dev, P)
        return self._packed[1]

    # ---- inference ------------------------------------------------------------------------------------------
    # Linears whose K (rows of the weight block) reaches this run on the bf16 matrix pipe (gemm_tile_bf3: three-way split, float32-
    # equivalent) -- the decoder's 384 .. 3072-deep steps and the deep unary / shortcut Linears are ~60 % of the forward's flops and
    # sat on the f32 pipe because their gathered residual had no bf16x3 epilogue until round 6.  ML3D_KP_LINEAR_B3: the K threshold
    # (0 = never: the A/B side), read when the module is imported.
    _LIN_B3_MINK = int(os.environ.get("ML3D_KP_LINEAR_B3", "256") or 0)

    @staticmethod
    def _packed_block(p, lo, hi):
        """bf16 planes of rows [lo, hi) of a packed Linear's weight (None: not eligible), built once per (block, slice)."""
        key = ('pk', lo, hi)
        if key not in p:
            p[key] = ops.pack_bf16x3(p['wt'][lo:hi].contiguous()) if (hi - lo) % 32 == 0 else None
        return p[key]

    @staticmethod
    def _unary(p, x, a2=None, gather=None, residual=None, act=None, slope=None):
        act = p['act'] if act is None else act
        slope = p['slope'] if slope is None else slope
        K = p['wt'].shape[0]
        if gather is None and KPFCNN._LIN_B3_MINK > 0 and K >= KPFCNN._LIN_B3_MINK and x.shape[1] % 32 == 0:
            pk = KPFCNN._packed_block(p, 0, K)
            if pk is not None:
                out = ops.linear_bf16x3(x, pk, p['wt'].shape[1], p['b'], act=act, slope=slope, a2=a2, residual=residual)
                if out is not None:
                    return out
        return ops.linear(x, p['wt'], p['b'], a2=a2, gather=gather, residual=residual, act=act, slope=slope)

    _SPLIT_DECODER = True        # decoder step split by linearity (+2 %, profiles/DESIGN_rounds_1_to_4.md §3.7); False = one gather + concat GEMM
    _FUSE_SHORTCUT = True        # unary2 + shortcut Linear as ONE GEMM over the concatenated K

    def _upsample_concat_unary(self, p, x, skip, up):
        """NearestUpsampleBlock + torch.cat + UnaryBlock of the decoder (kpconv.py:283-286, 821-838, 1468-1481):
        ``act(W . [x[up[:, 0]] ; skip] + b)``.  The Linear is split by linearity, ``(x W_x)[up[:, 0]] + skip W_skip``: the
        upsampled half is multiplied on the COARSE level (a quarter of the rows) and added back through the upsampling index
        in the fine GEMM's epilogue, whose A operand is then one dense row block (the streamlined K loop) instead of a
        gathered + concatenated one.  Same result up to summation order (tolerance 1e-4 on the logits)."""
        if not self._SPLIT_DECODER:
            return self._unary(p, x, a2=skip, gather=up)
        k1 = x.shape[1]
        K, n, mink = p['wt'].shape[0], p['wt'].shape[1], self._LIN_B3_MINK
        coarse = None
        if mink > 0 and k1 >= mink and (pk := self._packed_block(p, 0, k1)) is not None:
            coarse = ops.linear_bf16x3(x, pk, n)
        if coarse is None:
            coarse = ops.linear(x, p['wt'][:k1], None)
        if mink > 0 and K - k1 >= mink and (pk := self._packed_block(p, k1, K)) is not None:
            out = ops.linear_bf16x3(skip, pk, n, p['b'], act=p['act'], slope=p['slope'], residual=coarse, residual_gather=up)
            if out is not None:
                return out
        return ops.linear(skip, p['wt'][k1:], p['b'], residual=coarse, residual_gather=up, act=p['act'], slope=p['slope'])

    def forward(self, batch):
        """``batch``: object with ``points``, ``neighbors``, ``pools``, ``upsamples`` (lists per layer) and
        ``features`` — the reference's ``KPConvBatch`` (int64 CPU tensors are moved / narrowed to int32) or this
        module's GPU ``KPConvBatch``.  Returns logits [N0, num_classes - ignored] like ``KPFCNN.forward``
        (kpconv.py:270-291)."""
        if self.training:
            return self._forward_train(batch)
        dev = self.device
        _abi.require_gpu(dev, "KPFCNN.forward")
        P = self.packed_params(dev)
        lr = self.cfg.get('l_relu', 0.1)
        infl = _INFLUENCE[self.cfg.KP_influence]
        pts = [t.to(dev, torch.float32).contiguous() for t in batch.points]
        idx = lambda lst: [t.to(dev).to(torch.int32).contiguous() for t in lst]
        nbrs, pools, ups = idx(batch.neighbors), idx(batch.pools), idx(batch.upsamples)
        x = batch.features.to(dev, torch.float32).contiguous()
        skip_x = []
        for bi, (blk, p) in enumerate(zip(self.encoder_blocks, P['enc'])):
            if bi in self.encoder_skips:
                skip_x.append(x)
            L = blk.layer_ind
            strided = 'strided' in blk.block_name
            q_pts = pts[L + 1] if strided else pts[L]
            inds = pools[L] if strided else nbrs[L]
            c = p['conv']
            if c['ow'] is None:
                conv = lambda xin: ops.kpconv_rigid(q_pts, pts[L], inds, xin, c['kp'], c['w'], c['b'], c['extent'], 1, lr, infl,
                                                    packed=c['packed'])
            else:
                def conv(xin, m=blk.KPConv):
                    # the reference sets min_d2 / deformed_KP in EVERY forward, eval included (kpconv.py:1058,1074): its
                    # validation loss regularises the CURRENT batch.  The fused kernel does not materialise them, so the
                    # block keeps this call's inputs and `_offset_regulariser` derives the two tensors on demand
                    # (that pins the batch's features and neighbour matrices until the next forward or get_loss; a
                    #  caller that never asks for a loss -- pure inference -- sets ``model.retain_offset_geometry = False``)
                    m.min_d2 = m.deformed_KP = None
                    m._geom_inputs = (q_pts, pts[L], inds, xin, infl) if getattr(self, 'retain_offset_geometry', True) else None
                    return ops.kpconv_deformable(q_pts, pts[L], inds, xin, c['kp'], c['w'], c['b'], c['extent'], c['ow'],
                                                 c['ob'], 1, lr, infl)
            if isinstance(blk, SimpleBlock):
                x = conv(x)
            else:
                y = x if p['u1'] is None else self._unary(p['u1'], x)
                y = conv(y)
                sc = ops.gather_pool(x, inds, 'max') if strided else x
                if p['u2sc'] is not None:
                    x = self._unary(p['u2sc'], y, a2=sc, act=1, slope=lr)
                    continue
                if p['sc'] is not None:
                    sc = self._unary(p['sc'], sc)
                # unary2 (no relu) + shortcut, then LeakyReLU (kpconv.py:1451-1461): one GEMM epilogue
                x = self._unary(p['u2'], y, residual=sc, act=1, slope=lr)
        pending_up = None
        for bi, (blk, p) in enumerate(zip(self.decoder_blocks, P['dec'])):
            if isinstance(blk, NearestUpsampleBlock):
                pending_up = ups[blk.layer_ind - 1]           # fused into the next unary's A-operand gather
                continue
            if bi in self.decoder_concats:
                skip = skip_x.pop()
                if pending_up is not None:
                    x = self._upsample_concat_unary(p, x, skip, pending_up)
                    pending_up = None
                else:
                    x = self._unary(p, x, a2=skip)
            else:
                if pending_up is not None:
                    x = ops.gather_pool(x, pending_up, 'closest')
                    pending_up = None
                x = self._unary(p, x)
        if pending_up is not None:
            x = ops.gather_pool(x, pending_up, 'closest')
        x = self._unary(P['head'][0], x)
        return self._unary(P['head'][1], x)


    # ---- training forward: rigid architectures, differentiable ------------------------------------
    def _forward_train(self, batch):
        """``KPFCNN.forward`` in TRAINING mode (kpconv.py:270-291 with the blocks of :1343-1461): every KPConv is
        ``ops.KPConvFunction`` -- HIP aggregation forward, hand-written HIP scatter backward, the weight products on the
        library's MFMA GEMMs both ways -- and around it (``ML3D_TRAIN_OPS=hip``, the default; csrc/train.hip) hand-written HIP in
        both passes as well: bias-free Linears as ``ops.LinearFunction``, BatchNorm1d on the batch statistics (NOT folded: the
        inference kernels fold the running statistics, which training must update) + LeakyReLU as ``ops.BatchNormActFunction``,
        the max / closest pools as ``ops.GatherPoolFunction``; torch carries the graph, the concatenations and the residual
        add.  ``ML3D_TRAIN_OPS=torch`` keeps those modules on torch's autograd (rounds 3-4, the A/B side).  DEFORMABLE blocks (kpconv.py:1011-1159): the inner rigid convolution that
        produces the offsets is ``ops.KPConvFunction`` too; the deformed convolution itself -- whose influences depend on the
        trained offsets -- is written out in torch ([Nq, H, K] squared distances, linear influences, one matmul per block) so
        that autograd carries the gradient into the offsets; the block keeps ``min_d2`` / ``deformed_KP`` for the regulariser
        of ``get_loss``.  The reference's pruning of the neighbour rows (:1071-1103) is skipped: it only drops neighbours whose
        linear influence -- and its gradient -- is zero for every kernel point."""
        import torch.nn.functional as F
        dev = self.device
        _abi.require_gpu(dev, "KPFCNN.forward (training)")
        lr = self.cfg.get('l_relu', 0.1)
        infl = _INFLUENCE[self.cfg.KP_influence]
        pts = [t.to(dev, torch.float32).contiguous() for t in batch.points]
        idx = lambda lst: [t.to(dev).to(torch.int32).contiguous() for t in lst]
        nbrs, pools, ups = idx(batch.neighbors), idx(batch.pools), idx(batch.upsamples)

        hip = os.environ.get("ML3D_TRAIN_OPS", "hip").strip().lower() != "torch"

        def bn_act(blk, x, slope):  # BatchNormBlock.forward (kpconv.py:1238-1249: per-channel statistics over the N rows) + LeakyReLU
            if blk.use_bn and hip:
                return ops.batch_norm_act(x, blk.batch_norm, slope)
            x = blk.batch_norm(x) if blk.use_bn else x + blk.bias
            return x if slope is None else F.leaky_relu(x, slope)

        def unary(ub, x):           # UnaryBlock.forward (kpconv.py:1288-1293)
            y = ops.LinearFunction.apply(x, ub.mlp.weight, None) if hip else ub.mlp(x)
            return bn_act(ub.batch_norm, y, None if ub.no_relu else ub.l_relu)

        def padded(x):              # the shadow neighbour's zero feature row (kpconv.py:809-811, 848-850)
            return torch.cat([x, torch.zeros_like(x[:1])], 0)

        x = batch.features.to(dev, torch.float32).contiguous()
        skip_x = []
        for bi, blk in enumerate(self.encoder_blocks):
            if bi in self.encoder_skips:
                skip_x.append(x)
            L = blk.layer_ind
            strided = 'strided' in blk.block_name
            q_pts = pts[L + 1] if strided else pts[L]
            inds = pools[L] if strided else nbrs[L]
            conv = blk.KPConv
            if conv.deformable:
                if self.cfg.KP_influence != 'linear' or self.cfg.get('aggregation_mode', 'sum') != 'sum':
                    raise NotImplementedError("KPFCNN (MI355X build): deformable training with KP_influence 'linear' / sum only")
                kp_apply = lambda xin, conv=conv, q_pts=q_pts, sp=pts[L], inds=inds: self._deformable_train(conv, q_pts, sp, inds, xin, infl)
            else:
                kp_apply = lambda xin, conv=conv, q_pts=q_pts, sp=pts[L], inds=inds: ops.KPConvFunction.apply(
                    xin, conv.weights, q_pts, sp, inds, conv.kernel_points, conv.KP_extent, infl)
            if isinstance(blk, SimpleBlock):
                x = bn_act(blk.batch_norm, kp_apply(x), lr)
                continue
            y = x if isinstance(blk.unary1, nn.Identity) else unary(blk.unary1, x)
            y = bn_act(blk.batch_norm_conv, kp_apply(y), lr)
            y = unary(blk.unary2, y)
            if not strided:
                sc = x
            elif hip:
                sc = ops.GatherPoolFunction.apply(x, inds, "max")                       # max_pool (kpconv.py:841-858)
            else:
                sc = padded(x)[inds.long()].max(1)[0]
            if not isinstance(blk.unary_shortcut, nn.Identity):
                sc = unary(blk.unary_shortcut, sc)
            x = F.leaky_relu(y + sc, lr)
        for bi, blk in enumerate(self.decoder_blocks):
            if bi in self.decoder_concats:
                x = torch.cat([x, skip_x.pop()], dim=1)
            if isinstance(blk, NearestUpsampleBlock):
                up = ups[blk.layer_ind - 1]
                if hip:
                    x = ops.GatherPoolFunction.apply(x, up, "closest")                  # closest_pool (kpconv.py:821-838)
                else:
                    x = padded(x)[up[:, 0].long()]
            else:
                x = unary(blk, x)
        return unary(self.head_softmax, unary(self.head_mlp, x))

    @staticmethod
    def _deformable_geometry(conv, q_pts, s_pts, inds, x, infl, need_sq=True):
        """Offsets -> deformed kernel points and their squared distances to the neighbours (kpconv.py:1011-1066): sets
        ``conv.deformed_KP`` [Nq, K, 3] and ``conv.min_d2`` [Nq, K], returns ([Nq, H, K] squared distances, modulations).
        ``need_sq=False`` (the HIP training path): the [Nq, H, K] tensor is NOT built -- the aggregation
        (``ops.KPConvDeformedFunction``) and the regulariser (``ops.OffsetRegulariserFunction``, which also yields ``min_d2``) work
        from ``deformed_KP`` and the block's geometry, kept in ``conv._reg_geom``; returns (None, modulations)."""
        K = conv.K
        oc = conv.offset_conv
        off = ops.KPConvFunction.apply(x, oc.weights, q_pts, s_pts, inds, oc.kernel_points, oc.KP_extent, infl) + conv.offset_bias
        if conv.modulated:
            unscaled, mod = off[:, :3 * K].reshape(-1, K, 3), 2 * torch.sigmoid(off[:, 3 * K:])
        else:
            unscaled, mod = off.reshape(-1, K, 3), None
        conv.deformed_KP = unscaled * conv.KP_extent + conv.kernel_points                 # [Nq, K, 3]
        conv._reg_geom = (q_pts, s_pts, inds)
        if not need_sq:
            conv.min_d2 = None
            return None, mod
        far = torch.cat([s_pts, torch.zeros_like(s_pts[:1]) + 1e6], 0)                    # the shadow neighbour's position
        nb = far[inds.long()] - q_pts.unsqueeze(1)                                        # [Nq, H, 3]
        sq = ((nb.unsqueeze(2) - conv.deformed_KP.unsqueeze(1)) ** 2).sum(3)              # [Nq, H, K]
        conv.min_d2 = sq.min(1)[0]
        return sq, mod

    @staticmethod
    def _deformable_train(conv, q_pts, s_pts, inds, x, infl):
        """One deformable KPConv in training mode (kpconv.py:1011-1066, 1105-1159, linear influence, sum aggregation)."""
        conv._geom_inputs = None
        hip_ops = os.environ.get("ML3D_TRAIN_OPS", "hip").strip().lower() != "torch"
        sq, mod = KPFCNN._deformable_geometry(conv, q_pts, s_pts, inds, x, infl, need_sq=not hip_ops)
        if hip_ops:
            # round 5: the aggregation and its adjoint -- with respect to the features AND the deformed kernel points, which is what
            # trains the offset convolution -- on csrc/train.hip; no [Nq, H, Cin] gather (ops.KPConvDeformedFunction)
            K, cin, cout = conv.weights.shape
            wf = ops.KPConvDeformedFunction.apply(x, conv.deformed_KP, q_pts, s_pts, inds, conv.KP_extent)      # [Nq, K * Cin]
            if mod is not None:
                wf = (wf.view(-1, K, cin) * mod.unsqueeze(2)).reshape(-1, K * cin)
            return ops.LinearFunction.apply(wf, conv.weights.reshape(K * cin, cout).t(), None)
        w = torch.clamp(1 - torch.sqrt(sq) / conv.KP_extent, min=0.0).transpose(1, 2)     # [Nq, K, H]
        nx = torch.cat([x, torch.zeros_like(x[:1])], 0)[inds.long()]                      # [Nq, H, Cin]
        wf = torch.matmul(w, nx)                                                          # [Nq, K, Cin]
        if mod is not None:
            wf = wf * mod.unsqueeze(2)
        return torch.matmul(wf.permute(1, 0, 2), conv.weights).sum(0)

    def _offset_regulariser(self):
        """``p2p_fitting_regularizer`` (kpconv.py:2167-2206) over the deformable blocks of the last training forward: every
        deformed kernel point should sit on an input point (L1 of the normalised squared distance to the closest neighbour,
        weight 2) and keep its distance from the other kernel points (squared hinge at ``repulse_extent``, the others detached)."""
        cfg = self.cfg
        l1 = torch.nn.L1Loss()
        fitting, repulsive, K = 0, 0, int(cfg.num_kernel_points)
        for m in self.modules():
            if isinstance(m, KPConv) and m.deformable:
                hip_ops = os.environ.get("ML3D_TRAIN_OPS", "hip").strip().lower() != "torch"
                if getattr(m, 'min_d2', None) is None and getattr(m, 'deformed_KP', None) is None:
                    geom = getattr(m, '_geom_inputs', None)
                    if geom is None:
                        raise RuntimeError("KPFCNN.get_loss: no forward has run through the deformable blocks")
                    with torch.no_grad():          # an eval-mode forward: the current batch's geometry, no gradient
                        self._deformable_geometry(m, *geom, need_sq=not hip_ops)
                    m._geom_inputs = None
                if hip_ops and m.deformed_KP.is_cuda and K <= 16 and getattr(m, '_reg_geom', None) is not None:
                    # round 6: both terms and their gradient in ONE HIP kernel (no [Nq, H, K] distances, no per-kernel-point loop)
                    q_pts, s_pts, inds = m._reg_geom
                    terms, m.min_d2 = ops.OffsetRegulariserFunction.apply(m.deformed_KP, q_pts, s_pts, inds, float(m.KP_extent),
                                                                          float(cfg.get('repulse_extent', 1.2)))
                    fitting = fitting + terms[0]
                    repulsive = repulsive + terms[1]
                    continue
                if m.min_d2 is None:
                    with torch.no_grad():
                        far = torch.cat([m._reg_geom[1], torch.zeros_like(m._reg_geom[1][:1]) + 1e6], 0)
                    nb = far[m._reg_geom[2].long()] - m._reg_geom[0].unsqueeze(1)
                    m.min_d2 = ((nb.unsqueeze(2) - m.deformed_KP.unsqueeze(1)) ** 2).sum(3).min(1)[0]
                d2 = m.min_d2 / (m.KP_extent ** 2)
                fitting = fitting + l1(d2, torch.zeros_like(d2))
                locs = m.deformed_KP / m.KP_extent
                for i in range(K):
                    others = torch.cat([locs[:, :i], locs[:, i + 1:]], 1).detach()
                    dist = torch.sqrt(((others - locs[:, i:i + 1]) ** 2).sum(2))
                    rep = (torch.clamp_max(dist - cfg.get('repulse_extent', 1.2), 0.0) ** 2).sum(1)
                    repulsive = repulsive + l1(rep, torch.zeros_like(rep)) / K
        return cfg.get('deform_fitting_power', 1.0) * (2 * fitting + repulsive)

    # ---- the reference's data path around forward (kpconv.py:353-633), on the GPU ops ---------------------------------
    def preprocess(self, data, attr):
        """kpconv.py:353-396: grid subsample at ``first_subsampling_dl``, search structure, raw -> sub projection."""
        from ._datapath import preprocess_segmentation
        return preprocess_segmentation(data, attr, self.cfg.first_subsampling_dl, self.device,
                                       proj_splits=("test", "testing", "validation", "valid"),
                                       host_index=self.cfg.get('sampler_index', 'gpu') == 'sklearn')

    def _draw_augmentation(self, n, dim):
        """The np.random draws of the reference's augmentation (kpconv.py:647-712), in its order -- rotation angle(s), scale,
        symmetry flips, per-point noise -- so that a seeded run consumes the generator exactly like the reference does.
        Returns (R [dim, dim] f32, scale [dim] f32, noise [n, dim] f32)."""
        cfg, two_pi = self.cfg, 2 * np.pi
        R = np.eye(dim)
        if dim == 3 and cfg.augment_rotation == 'vertical':
            a = np.random.rand() * two_pi
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
        elif dim == 3 and cfg.augment_rotation == 'all':
            from ._datapath import create_3D_rotations
            a = np.random.rand() * two_pi
            b = (np.random.rand() - 0.5) * np.pi
            axis = np.array([np.cos(a) * np.cos(b), np.sin(a) * np.cos(b), np.sin(b)])
            R = create_3D_rotations(axis.reshape(1, -1), (np.random.rand() * two_pi).reshape(1, -1))[0]
        lo, hi = cfg.augment_scale_min, cfg.augment_scale_max
        if cfg.augment_scale_anisotropic:
            scale = np.random.rand(dim) * (hi - lo) + lo
        else:
            scale = np.random.rand() * (hi - lo) - lo          # sic: the reference subtracts min (kpconv.py:701)
        flips = np.array(cfg.augment_symmetries).astype(np.int32) * np.random.randint(2, size=dim)
        scale = (scale * (1 - flips * 2)).astype(np.float32)
        noise = (np.random.randn(n, dim) * cfg.augment_noise).astype(np.float32)
        return R.astype(np.float32), scale, noise

    def augmentation_transform(self, points, normals=None, verbose=False, is_test=False):
        """kpconv.py:647-744: random rotation / scale / flips / noise of one input sphere -> (points, scale, R).  With
        ``is_test`` the points come back untouched (the draws still advance ``np.random``).  NOTE the reference's pipelines
        call ``transform`` WITHOUT ``is_test`` also at inference (torch_dataloader.py:85), i.e. the augmentation is live in
        ``run_inference`` / ``run_test``; only the legacy ``inference_preprocess`` passes ``is_test=True``."""
        if normals is not None:
            raise NotImplementedError("KPFCNN (MI355X build): normals are not part of the segmentation path")
        R, scale, noise = self._draw_augmentation(points.shape[0], points.shape[1])
        if is_test:
            return points, scale, R
        return np.sum(points[:, :, None] * R, axis=1) * scale + noise, scale, R

    def _crop_sphere(self, data, budget, min_pts, split, is_test):
        """ONE input sphere around the sampler's next centre (the body of the loop of kpconv.py:446-531): recentred on the
        centre, normalised (``trans_normalize``, ml3d/datasets/utils/transforms.py:7-26, in place on the cached features like
        the reference), cut down to ``budget`` points at random, augmented."""
        cloud, labels, colours = data['point'], data['label'], data['feat']
        sphere, picked, centre = self.trans_point_sampler(pc=cloud.copy(), feat=colours, label=labels,
                                                          search_tree=data['search_tree'], num_points=min_pts,
                                                          radius=self.cfg.in_radius)
        sphere = sphere - centre
        norm = self.cfg.get('t_normalize', {})
        axes = norm.get('recentering', [0, 1, 2])
        sphere[:, axes] = sphere[:, axes] - sphere.mean(0)[axes]
        method = norm.get('method', None)
        if method == 'linear':
            if norm.get('normalize_points', False):
                sphere -= sphere.mean()
                sphere /= (sphere.max(0) - sphere.min(0)).max()
            if colours is not None:
                colours -= norm.get('feat_bias', 0)
                colours /= norm.get('feat_scale', 1)
        elif method == 'coords_only':
            colours = None
        columns = sphere.copy() if colours is None else np.hstack((sphere, colours[picked, :]))
        sphere_labels = labels[picked]
        if sphere.shape[0] > budget:
            keep = np.random.choice(sphere.shape[0], size=budget, replace=False)
            sphere, columns, sphere_labels, picked = sphere[keep, :], columns[keep, :], sphere_labels[keep], picked[keep]
        out_pts, scale, R = self.augmentation_transform(sphere, is_test=is_test)
        if np.random.rand() > self.cfg.augment_color:
            columns[:, 3:] *= 0
        return dict(p_list=out_pts, f_list=columns, l_list=np.squeeze(sphere_labels), p0_list=centre, s_list=scale, R_list=R,
                    r_inds_list=data['proj_inds'] if split in ['test'] else np.zeros((0,)), r_mask_list=picked,
                    val_labels_list=labels.astype(np.int32))

    def transform(self, data, attr, is_test=False):
        """kpconv.py:398-533: input spheres around the sampler's centres until ``min_in_points`` points are collected (at most
        ``min(batch_limit, max_in_points)``); returns the reference's dict of per-sphere lists (numpy) + ``cfg``, which is
        what ``KPConvBatch`` (this module: GPU neighbour / pooling build) or the reference's ``ConcatBatcher`` consume."""
        if getattr(self, 'trans_point_sampler', None) is None:
            raise RuntimeError("KPFCNN.transform: set model.trans_point_sampler (the pipeline takes it from the dataset "
                               "split's sampler) or use inference_begin()")
        keys = ('p_list', 'f_list', 'l_list', 'p0_list', 's_list', 'R_list', 'r_inds_list', 'r_mask_list', 'val_labels_list')
        result = {k_: [] for k_ in keys}
        result['cfg'] = self.cfg
        cap = min(self.cfg.batch_limit, self.cfg.max_in_points)
        want = min(self.cfg.get('min_in_points', 3), self.cfg.max_in_points)
        have = 0
        while have < want:
            sphere = self._crop_sphere(data, cap - have, want, attr['split'], is_test)
            have += sphere['p_list'].shape[0]
            for k_ in keys:
                result[k_].append(sphere[k_])
        return result

    def make_batch(self, transformed):
        """``ConcatBatcher.collate_fn`` for ONE transformed cloud -> the GPU ``KPConvBatch`` (``ml3d.torch.dataloaders``)."""
        from ..dataloaders import ConcatBatcher
        return ConcatBatcher(self.device, 'KPFCNN').collate_fn([{'data': transformed, 'attr': {}}])['data']

    def update_probs(self, inputs, results, test_probs):
        """kpconv.py:560-587: per input sphere, smooth the votes of the points it covers (float16 accumulator)."""
        self.test_smooth = 0.95
        batch = inputs['data']
        lengths = batch.lengths[0].cpu().numpy() if isinstance(batch.lengths[0], torch.Tensor) else np.asarray(batch.lengths[0])
        dev = self.device
        on_host = isinstance(test_probs, np.ndarray)
        tp = torch.from_numpy(np.ascontiguousarray(test_probs)).to(dev) if on_host else test_probs
        logits = results.to(dev, torch.float32).contiguous()
        i0 = 0
        for b_i, length in enumerate(lengths):
            length = int(length)
            mask = torch.as_tensor(np.asarray(batch.reproj_masks[b_i]), dtype=torch.int32).to(dev)
            ops.vote_update(tp, mask, logits[i0:i0 + length], self.test_smooth)
            i0 += length
        return tp.cpu().numpy() if on_host else tp

    def inference_begin(self, data):
        self.test_smooth = 0.98
        attr = {'split': 'test'}
        self.inference_ori_data = data
        self.inference_data = self.preprocess(data, attr)
        self.inference_proj_inds = self.inference_data['proj_inds']
        num_points = self.inference_data['search_tree'].data.shape[0]
        self.possibility = np.random.rand(num_points) * 1e-3
        self.test_probs = torch.zeros((num_points, self.cfg.num_classes), dtype=torch.float16, device=self.device)
        if getattr(self, 'trans_point_sampler', None) is None:
            self.trans_point_sampler = self._possibility_sampler

    def _possibility_sampler(self, pc, feat, label, search_tree, num_points, radius=None):
        """Radius-based spatially regular sampler on ``self.possibility`` (semseg_spatially_regular.py:62-108)."""
        n = 0
        while n < 2:
            center_id = int(np.argmin(self.possibility))
            center_point = pc[center_id, :].reshape(1, -1)
            idxs = search_tree.query_radius(center_point, r=radius)[0]
            n = len(idxs)
            if n < 2:
                self.possibility[center_id] += 0.001
        idxs = np.random.permutation(idxs)
        pc = pc[idxs]
        dists = np.sum(np.square((pc - center_point).astype(np.float32)), axis=1)
        self.possibility[idxs] += np.square(1 - dists / np.max(dists))
        return pc, idxs, center_point

    def inference_preprocess(self):
        attr = {'split': 'test'}
        data = self.transform(self.inference_data, attr, is_test=True)
        self.inference_input = {'data': self.make_batch(data), 'attr': attr}
        return self.inference_input

    def inference_end(self, inputs, results):
        self.update_probs(inputs, results, self.test_probs)
        if np.min(self.possibility) > 0.5:
            probs = self.test_probs.cpu().numpy()
            pred_labels = np.argmax(probs, 1)[self.inference_proj_inds]
            self.inference_result = {'predict_labels': pred_labels, 'predict_scores': probs[self.inference_proj_inds]}
            return True
        return False

    def get_optimizer(self, cfg_pipeline):
        """kpconv.py:293-313: SGD with a separate learning rate for the deformable offsets' parameters."""
        deform = [v for k, v in self.named_parameters() if 'offset' in k]
        other = [v for k, v in self.named_parameters() if 'offset' not in k]
        optimizer = torch.optim.SGD([{'params': other}, {'params': deform, 'lr': cfg_pipeline.learning_rate * cfg_pipeline.deform_lr_factor}],
                                    lr=cfg_pipeline.learning_rate, momentum=cfg_pipeline.momentum, weight_decay=cfg_pipeline.weight_decay)
        return optimizer, torch.optim.lr_scheduler.ExponentialLR(optimizer, cfg_pipeline.scheduler_gamma)

    def get_loss(self, Loss, results, inputs, device):
        """kpconv.py:315-351: class-weighted cross entropy over the non-ignored points + the point-to-point regulariser of the
        deformable offsets (kpconv.py:2167-2206; zero for rigid architectures), which reads the ``min_d2`` / ``deformed_KP``
        of the LAST forward, eval included, as on the reference (kpconv.py:1058,1074): the training forward leaves them on the
        deformable blocks; the fused inference kernels do not materialise them, so after an eval-mode forward they are
        derived here, without gradient, from the inputs that forward handed to the block (validation loss of run_train)."""
        from ..modules import valid_scores_and_labels
        cfg = self.cfg
        scores, labels = valid_scores_and_labels(results, inputs['data'].labels, cfg.num_classes, cfg.ignored_label_inds, device)
        self.output_loss = Loss.weighted_CrossEntropyLoss(scores, labels)
        if any('deformable' in b for b in cfg.architecture):
            if cfg.get('deform_fitting_mode', 'point2point') != 'point2point':
                raise ValueError('Unknown fitting mode: ' + str(cfg.deform_fitting_mode))
            self.reg_loss = self._offset_regulariser()
        else:
            self.reg_loss = torch.zeros((), device=scores.device)
        return self.output_loss + self.reg_loss, labels, scores


class KPConvBatch:
    """GPU construction of the network inputs of ``KPConvBatch.segmentation_inputs``
    (ml3d/torch/dataloaders/concat_batcher.py:186-305): per layer the stacked points, conv neighbours
    (radius r), pooled points (grid 2r/conv_radius, randomly oriented like ``batch_grid_subsampling``), pool
    neighbours (r) and upsample neighbours (2r), as int32 matrices on the device.

    ``rotations``: "random" draws the grid orientations from ``np.random`` with the reference's call sequence
    (kpconv.py:2059-2080), ``None`` keeps axis-aligned grids, or a list of float32 [B,3,3] arrays per pooling
    layer."""

    def __init__(self, points, lengths, cfg, features=None, rotations="random", device='cuda', one_call=True, buffers=None):
        """``one_call=False`` forces the per-layer path (the whole-batch library call is tried first otherwise; both produce the
        same matrices -- tests/test_gpu_kpconv.py, tests/test_emulated_api.py).  ``buffers``: see ``ops.kpconv_batch_build`` (a
        pipeline's reusable workspace / arena; the batch's tensors then alias the arena until that dict is used again)."""
        dev = torch.device(device)
        _abi.require_gpu(dev, "KPConvBatch")
        self.cfg = cfg
        pts = torch.as_tensor(points, dtype=torch.float32).to(dev).contiguous()
        lens = [int(v) for v in lengths]
        if features is None:
            features = torch.ones((pts.shape[0], 1), dtype=torch.float32, device=dev)     # in_features_dim == 1
        self.features = torch.as_tensor(features, dtype=torch.float32).to(dev).contiguous()
        self.points, self.neighbors, self.pools, self.upsamples, self.lengths, self.rotations = [], [], [], [], [], []
        if one_call and self._build_in_one_call(pts, lens, cfg, rotations, dev, buffers):
            return
        r_normal = cfg['first_subsampling_dl'] * cfg['conv_radius']
        layer_blocks = []
        e_i = torch.empty((0, 1), dtype=torch.int32, device=dev)
        deform = cfg.get('deform_radius', 6.0) / cfg['conv_radius']
        for block in cfg['architecture']:
            if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
                layer_blocks.append(block)
                continue
            # layers with a deformable block search with deform_radius instead of conv_radius (concat_batcher.py:219-252)
            r_conv = r_normal * deform if any('deformable' in b for b in layer_blocks) else r_normal
            r_pool = r_normal * deform if 'deformable' in block else r_normal
            # the conv search's sizes are read together with the subsampling's, the pool and upsample searches' together as
            # well: 2 host read-backs per pooling layer, 9 per 5-layer batch
            conv_plan = ops.radius_plan_dense(pts, pts, lens, lens, r_conv, long_rows=r_conv > r_normal) if layer_blocks else None
            if 'pool' in block or 'strided' in block:
                dl = 2 * r_normal / cfg['conv_radius']
                li = len(self.points)
                if isinstance(rotations, str):
                    R = random_grid_rotations(len(lens))
                elif rotations is None:
                    R = None
                else:
                    R = rotations[li]
                Rt = None if R is None else torch.as_tensor(R, dtype=torch.float32).to(dev)
                # ONE read-back for the subsampling's sizes (pooled points, per-item lengths) and the conv search's two
                sub = ops.grid_subsampling_plan(pts, lens, dl, Rt)
                parts = [sub.stats, sub.out_len] + ([conv_plan.stats] if conv_plan is not None else [])
                vals = torch.cat(parts).tolist()
                sub.resolve(vals[:2])
                pool_lens = [int(v) for v in vals[2:2 + len(lens)]]
                pool_p, _ = sub.fill()
                if conv_plan is not None:
                    conv_plan.resolve(vals[2 + len(lens):])
                    conv_i = ops.radius_fill_dense(conv_plan, pts.shape[0])
                else:
                    conv_i = e_i
                # (same supports and, unless only one of the two is deformable, the same radius as the conv search: its grid is
                #  searched again; the upsample radius is twice the POOL radius, concat_batcher.py:262-263)
                pool_plan = ops.radius_plan_dense(pool_p, pts, pool_lens, lens, r_pool, long_rows=r_pool > r_normal,
                                                  grid_from=conv_plan if (r_pool == r_conv and r_pool == r_normal) else None)
                up_plan = ops.radius_plan_dense(pts, pool_p, lens, pool_lens, 2 * r_pool, long_rows=r_pool > r_normal)
                ops.resolve_plans(pool_plan, up_plan)
                pool_i = ops.radius_fill_dense(pool_plan, pts.shape[0])
                up_i = ops.radius_fill_dense(up_plan, pool_p.shape[0])
                self.rotations.append(R)
            else:
                conv_i = ops.radius_fill_dense(conv_plan, pts.shape[0]) if conv_plan is not None else e_i
                pool_p = torch.empty((0, 3), dtype=torch.float32, device=dev)
                pool_lens, pool_i, up_i = [], e_i, e_i
            self.points.append(pts)
            self.neighbors.append(conv_i)
            self.pools.append(pool_i)
            self.upsamples.append(up_i)
            self.lengths.append(torch.tensor(lens, dtype=torch.int32))
            pts, lens = pool_p, pool_lens
            r_normal *= 2
            layer_blocks = []
            if 'global' in block or 'upsample' in block:
                break


def _kpconv_build_in_one_call(self, pts, lens, cfg, rotations, dev, buffers=None):
    """The rigid architectures' batch build as ONE library call (``ops.kpconv_batch_build`` -> ``ml3d_kpconv_batch_build``: every
    launch of the 5-layer chain enqueued from C++, one blocking size read-back per layer instead of two per pooling layer, no
    interpreter between two kernels).  Returns False -- nothing appended, no random draw consumed -- when the architecture is
    not the plain ``[blocks..., pool/strided]* [blocks..., global/upsample]`` shape or has deformable blocks (their 6 x wider
    searches take the two-phase path), and when a row outgrows the one-traversal stash: the per-layer loop then runs."""
    arch = list(cfg['architecture'])
    if any('deformable' in b for b in arch):
        return False
    has_conv, closing, blocks = [], [], 0
    for block in arch:
        if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
            blocks += 1
            continue
        has_conv.append(blocks > 0)
        closing.append('pool' in block or 'strided' in block)
        blocks = 0
        if not closing[-1]:
            break
    L = len(closing)
    if L == 0 or closing[-1] or not all(closing[:-1]) or L > _abi.KPBATCH_MAX_LAYERS or not lens:
        return False
    r_normal = cfg['first_subsampling_dl'] * cfg['conv_radius']
    radii, dls = [], []
    for l in range(L):
        radii.append(r_normal)
        dls.append(2 * r_normal / cfg['conv_radius'])
        r_normal *= 2
    state = np.random.get_state() if isinstance(rotations, str) else None
    if isinstance(rotations, str):
        R = [random_grid_rotations(len(lens)) for _ in range(L - 1)]          # (the per-layer loop draws them in this order too)
    elif rotations is None:
        R = [None] * (L - 1)
    else:
        R = [rotations[l] for l in range(L - 1)]
    res = ops.kpconv_batch_build(pts, lens, radii, dls, has_conv, R, buffers=buffers)
    if res is None:
        if state is not None:
            np.random.set_state(state)          # the per-layer path draws the same orientations again
        return False
    self.points, self.neighbors, self.pools, self.upsamples = res.points, res.neighbors, res.pools, res.upsamples
    self.lengths = res.lengths
    self.rotations = R
    self._arena = res.arena                     # (the views keep it alive anyway)
    self.host_syncs = res.host_syncs
    return True


def _kpconv_batch_to(self, device):
    """``KPConvBatch.to`` of the reference (concat_batcher.py:327-341; the pipelines call it, semantic_segmentation.py:236):
    the matrices are built on the device already -- only a different device would mean a copy."""
    dev = torch.device(device)
    if dev.type == 'cuda' and self.points and self.points[0].device != dev and dev.index is not None:
        for name in ('points', 'neighbors', 'pools', 'upsamples'):
            setattr(self, name, [t.to(dev) for t in getattr(self, name)])
        self.features = self.features.to(dev)
    return self


KPConvBatch._build_in_one_call = _kpconv_build_in_one_call
KPConvBatch.to = _kpconv_batch_to
KPConvBatch.pin_memory = lambda self: self


def random_grid_rotations(B):
    """The np.random draws of ``batch_grid_subsampling`` (kpconv.py:2059-2080) followed by the closed-form axis-angle
    rotation of ``create_3D_rotations`` (ml3d/datasets/utils/operations.py:21-40): same draw order and the same float64
    operation order (temporaries t1 .. t24 as there), because the bit-exact batch golden depends on every rounding."""
    theta = np.random.rand(B) * 2 * np.pi
    phi = (np.random.rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)]).T
    alpha = np.random.rand(B) * 2 * np.pi
    t1 = np.cos(alpha)
    t2 = 1 - t1
    t3 = u[:, 0] * u[:, 0]
    t6 = t2 * u[:, 0]
    t7 = t6 * u[:, 1]
    t8 = np.sin(alpha)
    t9 = t8 * u[:, 2]
    t11 = t6 * u[:, 2]
    t12 = t8 * u[:, 1]
    t15 = u[:, 1] * u[:, 1]
    t19 = t2 * u[:, 1] * u[:, 2]
    t20 = t8 * u[:, 0]
    t24 = u[:, 2] * u[:, 2]
    R = np.stack([t1 + t2 * t3, t7 - t9, t11 + t12, t7 + t9, t1 + t2 * t15, t19 - t20, t11 - t12, t19 + t20,
                  t1 + t2 * t24], axis=1)
    return np.reshape(R, (-1, 3, 3)).astype(np.float32)
