#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
run() { timeout 300 python bench.py --workload pointpillars --steps 30 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f frames/s, step median %.3f p95 %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95']))"; }
{
echo "lanes 3, 24 sweeps: $(ML3D_PP_LANES=3 run --frames-per-step 24)"
echo "lanes 2, 16 sweeps: $(ML3D_PP_LANES=2 run)"
echo "lanes 3, 36 sweeps: $(ML3D_PP_LANES=3 run --frames-per-step 36)"
echo "lanes 3, 48 sweeps: $(ML3D_PP_LANES=3 run --frames-per-step 48)"
echo "lanes 3, 24 sweeps: $(ML3D_PP_LANES=3 run --frames-per-step 24)"
echo "lanes 2, 32 sweeps: $(ML3D_PP_LANES=2 run --frames-per-step 32)"
echo "lanes 3, 30 sweeps: $(ML3D_PP_LANES=3 run --frames-per-step 30)"
} > $O/pp_lanes2.log 2>&1
cat $O/pp_lanes2.log
