"""RandLA-Net (inference) on MI355X — host-side mirror of the reference model class.

Same constructor arguments, parameter names and state_dict layout as
``ml3d/torch/models/randlanet.py:17-113,471-692`` of the reference, so
``ml3d/configs/randlanet_*.yml`` and published ``.pth`` checkpoints load unchanged.
The module tree below only OWNS the parameters; ``forward`` does not run PyTorch ops —
it folds BatchNorm into packed weights once (``_randla_pack``) and calls the hand-written
HIP kernels through the C ABI (``ml3d.ops.randla_knn_pyramid`` + ``randla_forward``).
There is no CPU execution path: on a non-GPU device ``forward`` raises.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import _abi
from ... import ops
from . import _randla_pack


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
            raise RuntimeError("RandLANet (MI355X build) implements the inference forward only; call .eval()")
        dev = self.device
        if dev.type != 'cuda':
            raise RuntimeError("RandLANet.forward needs an MI355X device; there is no CPU fallback")
        coords = inputs['coords'][0] if isinstance(inputs['coords'], (list, tuple)) else inputs['coords']
        pts = coords.to(dev, torch.float32).contiguous()
        feat = inputs['features'].to(dev, torch.float32).contiguous()
        if 'neighbor_indices' in inputs and 'interp_idx' in inputs:
            nbr = [t.to(dev).to(torch.int32).contiguous() for t in inputs['neighbor_indices']]
            itp = [t.to(dev).to(torch.int32).contiguous() for t in inputs['interp_idx']]
        else:
            nbr, itp = self.neighbor_pyramid(pts)
        B, N, _ = pts.shape
        desc = _abi.make_desc(self.cfg, B, N)
        return ops.randla_forward(desc, self.packed_params(dev), feat, pts, nbr, itp)
