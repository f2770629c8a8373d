cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/kp_old.so $LIB/libml3d_hip.so; fi
  echo "== $v: $(python tools/roofline_ops.py kp 9 2>&1 | tail -1)"
  python bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{}); print('   %.0f %s step %.2f ms; block %.3f ms frac %.3f alone %.3f ms' % (d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_ms',-1), r.get('frac',-1), r.get('avg_launch_ms_alone',-1)))"
done
cp /tmp/new.so $LIB/libml3d_hip.so
timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_configs.py -x -q -k "kpconv" 2>&1 | tail -2
