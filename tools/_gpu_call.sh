cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
run() {
  echo "== $1: $(python tools/roofline_ops.py kp 9 2>&1 | tail -1)"
  echo "   bench: $(python bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{}); print('%.0f %s step %.2f ms; block %.3f ms frac %.3f alone %.3f ms' % (d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_ms',-1), r.get('frac',-1), r.get('avg_launch_ms_alone',-1)))")"
}
ML3D_KP_FUSED32=0 run unfused
run fused_d2
cp $LIB/ab/kf_d3.so $LIB/libml3d_hip.so
run fused_d3
cp /tmp/base.so $LIB/libml3d_hip.so
