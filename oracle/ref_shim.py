"""Import the REAL reference (``/root/reference/ml3d``) in this container — oracle tooling only.

The reference's Python needs the un-installable ``open3d`` wheel, ``addict`` and
``tensorboard`` at import time.  This module injects minimal
stand-ins into ``sys.modules`` and wires the oracle's CPU ops in where the reference
would call open3d's, so that the reference's own PyTorch model code can be run on CPU to
pin the oracle and to generate tests/golden/*.npz.  It is used ONLY by
``oracle/gen_golden.py``; nothing at test/bench/product run time imports it
(/root/reference does not exist on the GPU box).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("ML3D_REFERENCE_ROOT", "/root/reference")


class _AttrDict(dict):
    """Stand-in for addict.Dict with the semantics ml3d/utils/config.py:12-27 relies on."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            if a is None:
                continue
            for k, v in dict(a).items():
                self[k] = self._wrap(v)
        for k, v in kwargs.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            return self.__missing__(name)

    def __missing__(self, name):
        v = type(self)()
        # addict creates children lazily; the reference's ConfigDict overrides __missing__ to raise
        return v

    def __setattr__(self, name, value):
        self[name] = self._wrap(value)

    def __setitem__(self, name, value):
        super().__setitem__(name, self._wrap(value))

    def copy(self):
        return type(self)(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _AttrDict) else v) for k, v in self.items()}


def install():
    if "open3d" in sys.modules and getattr(sys.modules["open3d"], "_ml3d_oracle_shim", False):
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference checkout not found at %s" % REF_ROOT)
    import torch

    from . import ops as oops

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("addict", Dict=_AttrDict)

    class _O3Tensor:
        def __init__(self, a):
            self.a = a

        @staticmethod
        def from_numpy(a):
            return _O3Tensor(a)

        def numpy(self):
            return self.a

    class _NNS:
        def __init__(self, pts):
            self.pts = pts.a

        def knn_index(self):
            return True

        def knn_search(self, q, k):
            idx, d2 = oops.knn_search(self.pts, q.a, k, return_distances=True)
            return _O3Tensor(idx.astype("int64")), _O3Tensor(d2)

    class _FRS(torch.nn.Module):
        def forward(self, points, queries, radius, points_row_splits=None, queries_row_splits=None):
            r = oops.fixed_radius_search(points.numpy(), queries.numpy(), radius,
                                         None if points_row_splits is None else points_row_splits.numpy(),
                                         None if queries_row_splits is None else queries_row_splits.numpy())
            return type(r)(torch.from_numpy(r.neighbors_index), torch.from_numpy(r.neighbors_row_splits),
                           torch.from_numpy(r.neighbors_distance))

    def t_voxelize(points, row_splits, voxel_size, rmin, rmax, max_points_per_voxel=2**62, max_voxels=2**62):
        r = oops.voxelize(points.numpy(), row_splits.numpy(), voxel_size.numpy(), rmin.numpy(), rmax.numpy(),
                          max_points_per_voxel, max_voxels)
        return type(r)(*[torch.from_numpy(x) for x in r])

    def t_ragged_to_dense(values, row_splits, out_col_size, default_value):
        return torch.from_numpy(oops.ragged_to_dense(values.numpy(), row_splits.numpy(), out_col_size,
                                                     default_value.numpy()))

    def t_nms(boxes, scores, thr):
        return torch.from_numpy(oops.nms(boxes.detach().numpy(), scores.detach().numpy(), thr))

    def _unsupported(*a, **k):
        raise NotImplementedError("out-of-scope open3d op (oracle shim)")

    o3d = mod("open3d", _build_config={"BUILD_GUI": False, "BUILD_PYTORCH_OPS": True,
                                       "BUILD_TENSORFLOW_OPS": False, "BUILD_CUDA_MODULE": False},
              __version__="0.0.0-oracle-shim", _ml3d_oracle_shim=True)
    core = mod("open3d.core", Tensor=_O3Tensor)
    core.cuda = mod("open3d.core.cuda", device_count=lambda: 0, is_available=lambda: False)
    core.nns = mod("open3d.core.nns", NearestNeighborSearch=_NNS)
    o3d.core = core
    ml = mod("open3d.ml")
    ml.contrib = mod("open3d.ml.contrib", subsample=oops.subsample, subsample_batch=oops.subsample_batch,
                     iou_bev_cpu=oops.iou_bev, iou_3d_cpu=oops.iou_3d, iou_bev_cuda=oops.iou_bev,
                     iou_3d_cuda=oops.iou_3d)
    mlt = mod("open3d.ml.torch")
    mlt.ops = mod("open3d.ml.torch.ops", voxelize=t_voxelize, ragged_to_dense=t_ragged_to_dense, nms=t_nms,
                  knn_search=_unsupported, reduce_subarrays_sum=_unsupported, roi_pool=_unsupported,
                  furthest_point_sampling=_unsupported, three_nn=_unsupported, three_interpolate=_unsupported,
                  three_interpolate_grad=_unsupported, ball_query=_unsupported,
                  trilinear_devoxelize_forward=_unsupported, trilinear_devoxelize_backward=_unsupported)

    class _Sparse(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mlt.layers = mod("open3d.ml.torch.layers", FixedRadiusSearch=_FRS, SparseConv=_Sparse,
                     SparseConvTranspose=_Sparse)
    ml.torch = mlt
    o3d.ml = ml
    vis = mod("open3d.visualization")
    tbp = mod("open3d.visualization.tensorboard_plugin")
    tbp.summary = mod("open3d.visualization.tensorboard_plugin.summary")
    vis.tensorboard_plugin = tbp
    o3d.visualization = vis
    o3d.io = mod("open3d.io")
    o3d.t = mod("open3d.t")
    o3d.geometry = mod("open3d.geometry")
    o3d.utility = mod("open3d.utility")

    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        class _SW:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, n):
                return lambda *a, **k: None

        tb = mod("torch.utils.tensorboard", SummaryWriter=_SW)
        import torch.utils as tu
        tu.tensorboard = tb

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def reference_modules():
    """Returns the reference's (randlanet, kpconv, point_pillars) model modules."""
    install()
    import importlib
    rl = importlib.import_module("ml3d.torch.models.randlanet")
    assert os.path.abspath(rl.__file__).startswith(os.path.abspath(REF_ROOT)), rl.__file__
    return rl
