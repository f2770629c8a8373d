#!/bin/bash
# round 5: SECOND's convolutions on the bf16 matrix pipe (three-way split) -- accuracy + times per layer, the PointPillars parity
# tests on that path, and an alternating same-box A/B of the PointPillars workload
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5bf
mkdir -p $O
( timeout 300 python tools/bf16x3_check.py 16 2>&1 | tail -12 ) > $O/conv.log; cat $O/conv.log
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
