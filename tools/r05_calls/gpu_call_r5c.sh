#!/bin/bash
# round 5, third GPU call: several KPConv batch builds in flight (KPConvPipelineN) -- parity, then 1 / 2 / 3 builders alternating
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kpconv.py -x -q 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for t in 1 2 3 2 1 3; do
  echo "builders=$t: $(ML3D_KP_BUILDERS=$t timeout 300 python bench.py --workload kpconv --steps 30 --warmup 8 --no-cpu-baseline 2>$O/kp_$t.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f spheres/s, step median %.3f p95 %.3f ms, single sphere %.3f ms, a10 in-region %.3f alone %.3f ms, a11 %.3f / %.3f, block %.3f / %.3f' % (d['value'], d['step_ms_median'], d['step_ms_p95'], d['latency_single_sphere_ms']['median'], d['roofline_other'][0]['avg_launch_ms'], d['roofline_other'][0]['avg_launch_ms_alone'], d['roofline_other'][1]['avg_launch_ms'], d['roofline_other'][1]['avg_launch_ms_alone'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_alone']))")"
done > $O/kp_ab.log 2>&1
cat $O/kp_ab.log
tail -3 $O/kp_2.err
