#!/bin/bash
# cold box: the kpconv workload as the FIRST GPU process, then again -- per-step intervals; then the pipeline parity tests
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --workload kpconv --steps 30 --warmup 30 --no-cpu-baseline 2>$O/err_$i.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('run $i: %.1f spheres/s median %.2f p95 %.2f' % (d['value'], d['step_ms_median'], d['step_ms_p95'])); print('   ', d['step_ms_all'])"
done > $O/cold.log 2>&1
cat $O/cold.log; tail -3 $O/err_1.log
( timeout 600 python -m pytest tests/test_gpu_kpconv.py -x -q 2>&1 | tail -2 ) > $O/pytest.log; cat $O/pytest.log
