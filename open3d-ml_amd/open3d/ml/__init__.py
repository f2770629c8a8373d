"""``open3d.ml`` — ``contrib`` (native ops, this repository) plus, when ``OPEN3D_ML_ROOT`` names an Open3D-ML checkout,
that checkout's framework-independent packages re-exported unchanged (upstream's ``open3d/ml/__init__.py`` does the same:
``utils``, ``datasets``, ``vis``, ``configs`` come from the Open3D-ML tree, not from the wheel)."""
import os as _os
import sys as _sys

from . import contrib   # noqa: F401


def _checkout():
    root = _os.environ.get("OPEN3D_ML_ROOT")
    if not root:
        return None
    root = _os.path.abspath(root)
    if not _os.path.isdir(_os.path.join(root, "ml3d")):
        raise ImportError("OPEN3D_ML_ROOT=%s does not contain an ml3d/ package" % root)
    if root not in _sys.path:
        # the checkout's `ml3d` must win over any other package of that name on the path (its modules import each other
        # as `ml3d....`); this repository's product package is loaded under the alias `ml3d_amd` (open3d/_product.py)
        _sys.path.insert(0, root)
    m = _sys.modules.get("ml3d")
    if m is not None and not _os.path.abspath(getattr(m, "__file__", "") or "").startswith(root):
        raise ImportError("a different `ml3d` package is already imported (%s); import open3d.ml before it when "
                          "OPEN3D_ML_ROOT is set" % getattr(m, "__file__", "?"))
    return root


if _checkout():
    from ml3d import configs    # noqa: F401
    from ml3d import datasets   # noqa: F401
    from ml3d import utils      # noqa: F401
    try:
        from ml3d import vis    # noqa: F401   (needs the GUI build of Open3D; _build_config['BUILD_GUI'] is False here)
    except Exception:           # pragma: no cover
        vis = None
else:
    class _NoCheckout:
        """Placeholder for ``open3d.ml.utils`` / ``datasets`` without a checkout: the pipelines, datasets and configs are
        the reference's own files (out of this repository's scope); point OPEN3D_ML_ROOT at an Open3D-ML tree."""

        def __init__(self, name):
            self._n = name

        def __getattr__(self, item):
            raise ImportError("open3d.ml.%s comes from an Open3D-ML checkout: set OPEN3D_ML_ROOT (this repository ships the "
                              "native ops and the three hot-path models, not the pipelines / datasets)" % self._n)

    utils = _NoCheckout("utils")
    datasets = _NoCheckout("datasets")
    vis = _NoCheckout("vis")
