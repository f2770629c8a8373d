#!/bin/bash
# round 6, call L: run-time knobs re-tuned for the bf16x3 kernels (tile order levels, frames per step)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6l
mkdir -p $O
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'frames/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0))"; }
for t in 2 0 1 3 all 2; do ML3D_TILE_ORDER=$t run "tile_order=$t"; done > $O/sweep.log 2>&1
for b in 96 128 192 256 128; do ML3D_BENCH_BATCH=$b run "batch=$b"; done >> $O/sweep.log 2>&1
cat $O/sweep.log
( timeout 100 python -m pytest tests/test_gpu_training.py -q -k "regulariser or deformable" 2>&1 | tail -2 ) | cut -c1-200
