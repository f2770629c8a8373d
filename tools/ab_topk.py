"""GPU box helper: bench.py --workload pointpillars with the nms_pre cut done by ml3d_topk_rows ("new", the product) or by
torch.topk ("torch", what the path used until ABI 7) -- A/B of the hand-written radix select inside the timed step.
usage: python tools/ab_topk.py new|torch"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
mode = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py"), "--workload", "pointpillars", "--steps", "30", "--warmup", "8", "--no-cpu-baseline"]
if mode == "torch":
    import torch
    from ml3d.ops import detection
    detection.topk_rows = lambda v, k, with_values=False: torch.topk(v, int(k), dim=1)[1].contiguous()
runpy.run_path(sys.argv[0], run_name="__main__")
