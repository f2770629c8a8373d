#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pointpillars.py -x -q 2>&1 | tail -2 ) > $O/pytest.log; cat $O/pytest.log
for t in 0 1 0 1 0 1; do
  echo "pp threaded=$t: $(ML3D_PP_THREADED=$t timeout 300 python bench.py --workload pointpillars --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f frames/s, step median %.3f p95 %.3f ms; voxelize in-step %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95'], d['roofline_other'][0]['avg_launch_ms']))")"
done > $O/pp_threaded.log 2>&1
cat $O/pp_threaded.log
