#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_kpconv.py -x -q -k "several_builds or 64_spheres" 2>&1 | tail -2 ) > $O/pytest.log; cat $O/pytest.log
for f in 2 1 2 1 2 1; do
  echo "kpconv forward streams $f: $(ML3D_KP_FORWARD_STREAMS=$f timeout 600 python bench.py --workload kpconv --steps 30 --warmup 10 --no-cpu-baseline 2>$O/err_$f.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f spheres/s median %.3f p95 %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95']))")"
done > $O/kp_fwd_streams2.log 2>&1
cat $O/kp_fwd_streams2.log
echo "builders 3 + fwd 2: $(ML3D_KP_BUILDERS=3 ML3D_KP_FORWARD_STREAMS=2 timeout 600 python bench.py --workload kpconv --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f spheres/s median %.3f p95 %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95']))")" | tee -a $O/kp_fwd_streams2.log
