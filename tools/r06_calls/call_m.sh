#!/bin/bash
# round 6, call M: where the next batch's search may start inside the running forward (ML3D_SEARCH_GATE), re-tuned for the bf16x3 kernels
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6m
mkdir -p $O
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'frames/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0))"; }
for g in 9 -1 1 2 10 17 18 25 1100 9; do ML3D_SEARCH_GATE=$g run "gate=$g"; done > $O/sweep.log 2>&1
cat $O/sweep.log
