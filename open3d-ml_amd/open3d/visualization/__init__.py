"""``open3d.visualization`` — absent on purpose (out of scope; ``_build_config['BUILD_GUI']`` is False so
``ml3d/vis`` never asks for ``gui`` / ``rendering``).  Only the tensorboard plugin's import target exists."""
from . import tensorboard_plugin   # noqa: F401
