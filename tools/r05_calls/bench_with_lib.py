"""Run bench.py against another build of the library (ML3D_DIAG_LIB=path; A/B helper)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open3d-ml_amd")]
from ml3d import _abi
if os.environ.get("ML3D_DIAG_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["ML3D_DIAG_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
