#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
for b in 64 96 128 64 128; do
  echo "kpconv spheres per step $b: $(timeout 600 python bench.py --workload kpconv --frames-per-step $b --steps 24 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f spheres/s median %.3f p95 %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95']))")"
done > $O/kp_batch.log 2>&1
cat $O/kp_batch.log
echo "headline default: $(timeout 600 python bench.py --steps 20 --warmup 5 --no-workloads --no-latency --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330)"
