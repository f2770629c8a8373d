"""The three inference hot-path models of the reference (``ml3d/torch/models/{randlanet,kpconv,point_pillars}.py``),
MI355X-native: same constructor arguments, parameter names / state_dict layout and data-path methods."""
from .kpconv import KPFCNN, KPConvBatch
from .point_pillars import PointPillars
from .randlanet import RandLANet

__all__ = ["RandLANet", "KPFCNN", "KPConvBatch", "PointPillars"]
