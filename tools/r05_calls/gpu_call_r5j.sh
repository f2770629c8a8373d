#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-workloads --no-latency --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%.1f frames/s median %.3f p95 %.3f' % (d['value'], d['step_ms_median'], d['step_ms_p95']))"; }
for b in 64 96 128 64 128 192; do
  echo "frames per step $b: $(run --frames-per-step $b)"
done > $O/batch.log 2>&1
cat $O/batch.log
