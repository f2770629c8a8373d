#!/bin/bash
# round 5: PointPillars parity tests on the bf16x3 path + alternating same-box A/B of the workload (f32 MFMA vs bf16x3, both with
# the XCD-aware tile order)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5bi
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_configs.py -k "pointpillars or PointPillars or pillars" -x -q 2>&1 | tail -8 ) > $O/pytest.log; cat $O/pytest.log
for rep in 1 2; do
  for p in f32 bf16x3; do
    ( ML3D_PP_CONV=$p timeout 300 python bench.py --workload pointpillars --steps 40 --warmup 10 --no-cpu-baseline --no-latency 2>$O/pp_${p}_$rep.err | tail -1 ) > $O/pp_${p}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/pp_${p}_$rep.json"))
print("$p", "$rep", d["value"], d["ms_per_step"], d.get("roofline", {}).get("achieved"), d.get("roofline", {}).get("avg_launch_ms"))
PY
  done
done
