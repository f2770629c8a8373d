#!/bin/bash
# round 5, second GPU call: the one-call KPConv batch build (ml3d_kpconv_batch_build) -- parity, then the kpconv bench with the
# forward enqueued on the caller's thread / on a worker thread, alternating, and the host timeline of a step
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_configs.py tests/test_gpu_pipelines.py tests/test_gpu_training.py -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for t in 0 1 0 1; do
  echo "threaded=$t: $(ML3D_KP_THREADED=$t timeout 300 python bench.py --workload kpconv --steps 30 --warmup 8 --no-cpu-baseline 2>$O/kp_$t.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f spheres/s, step median %.3f p95 %.3f ms, single sphere %.3f ms, build %s' % (d['value'], d['step_ms_median'], d['step_ms_p95'], d['latency_single_sphere_ms']['median'], d.get('build'))); print(json.dumps(d['roofline_other']))")"
done > $O/kp_ab.log 2>&1
cat $O/kp_ab.log | cut -c1-600
( ML3D_KP_THREADED=0 timeout 300 python tools/kp_host_profile.py 20 2>&1 | head -40 ) > $O/kp_host_profile.log
head -3 $O/kp_host_profile.log
( ML3D_KP_THREADED=1 timeout 300 python tools/kp_host_profile.py 20 2>&1 | head -4 ) > $O/kp_host_profile_threaded.log
head -3 $O/kp_host_profile_threaded.log
tail -3 $O/kp_0.err
