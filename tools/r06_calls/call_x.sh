#!/bin/bash
# round 6, call X: where the next batch's search starts inside the running forward (ML3D_SEARCH_GATE), re-swept after the forward moved to the bf16 pipe
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6x
mkdir -p $O
for g in 9 -1 1 3 11 17 19 25 27 1001 1100 9 17; do
  ( ML3D_SEARCH_GATE=$g timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-workloads --no-latency 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('gate $g', 'frames/s %.0f' % d['value'], 'step_med %.2f p95 %.2f' % (d.get('step_ms_median',0), d.get('step_ms_p95',0)), 'knn in step %.3f' % r['avg_launch_ms'])"
done 2>&1 | tee $O/gate.log
