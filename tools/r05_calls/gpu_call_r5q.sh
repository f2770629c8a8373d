#!/bin/bash
# cold box, ONE configuration as the first GPU process (env from the caller), then the same again warm
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
for i in cold warm; do
  timeout 600 python bench.py --workload kpconv --steps 30 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('builders=$ML3D_KP_BUILDERS fwd=$ML3D_KP_FORWARD_STREAMS $i: %.1f spheres/s median %.2f p95 %.2f max %.1f' % (d['value'], d['step_ms_median'], d['step_ms_p95'], max(d['step_ms_all'])))"
done >> $O/cold_configs.log 2>&1
tail -2 $O/cold_configs.log
