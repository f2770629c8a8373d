#!/bin/bash
# round 5: PointPillars with the transposed convolutions and the head Linear on the bf16x3 path too: parity tests + same-box A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5bn
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_configs.py -k "pointpillars or PointPillars or pillars" -x -q 2>&1 | tail -8 ) > $O/pytest.log; cat $O/pytest.log
for rep in 1 2; do
  for p in f32 bf16x3; do
    ( ML3D_PP_CONV=$p timeout 300 python bench.py --workload pointpillars --steps 40 --warmup 10 --no-cpu-baseline --no-latency 2>$O/pp_${p}_$rep.err | tail -1 ) > $O/pp_${p}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/pp_${p}_$rep.json"))
r=d.get("roofline", {})
print("$p", "$rep", d["value"], d["ms_per_step"], r.get("achieved"), r.get("frac"), r.get("f32_equivalent_tflops"), r.get("avg_launch_ms"), r.get("avg_launch_ms_alone_lane_shape"))
PY
  done
done
