#!/bin/bash
# round 5: KPConv's [15 cin, cout] contractions (cin >= 64) on the bf16x3 kernel with split-K: parity tests + same-box A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5bo
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_configs.py -k "kpconv or KPConv or kpfcnn" -x -q 2>&1 | tail -8 ) > $O/pytest.log; cat $O/pytest.log
for rep in 1 2; do
  for p in f32 bf16x3; do
    ( ML3D_KP_GEMM=$p timeout 400 python bench.py --workload kpconv --steps 100 --warmup 30 --no-cpu-baseline --no-latency 2>$O/kp_${p}_$rep.err | tail -1 ) > $O/kp_${p}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/kp_${p}_$rep.json"))
print("$p", "$rep", d["value"], d["ms_per_step"], d.get("step_ms_median"))
PY
  done
done
for p in f32 bf16x3 f32 bf16x3; do echo $p; ML3D_KP_GEMM=$p timeout 200 python tools/kp_forward_seq.py 20 2>&1 | tail -2; done > $O/kp_forward_alone.log
cat $O/kp_forward_alone.log
