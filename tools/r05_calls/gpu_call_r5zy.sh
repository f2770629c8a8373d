#!/bin/bash
# round 5: PointPillars with the canvas zeroed by the library's fill kernel (HEAD) against hipMemsetAsync (the build before), alternating on one box
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zy
mkdir -p $O
for rep in 1 2; do
  for v in head memset; do
    L=""; [ $v = memset ] && L=open3d-ml_amd/ml3d/lib/ab/canvas_memset.so
    ( ML3D_DIAG_LIB=$L timeout 100 python tools/r05_calls/bench_with_lib.py --workload pointpillars --steps 40 --warmup 10 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 ) > $O/pp_${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/pp_${v}_$rep.json")); r=d["roofline"]
print("$v", "$rep", "%.0f frames/s" % d["value"], "step %.2f ms" % d["ms_per_step"], "conv in step %.3f ms" % r["avg_launch_ms"])
PY
  done
done
