#!/bin/bash
# A/B on one box, alternating: the k-NN of round 4 (lib/ab/knn_old.so) against HEAD's -- headline frames/s, search gate variants
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/new.so
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-workloads --no-latency --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%.1f frames/s median %.3f p95 %.3f | knn in-region %.3f alone %.3f' % (d['value'], d['step_ms_median'], d['step_ms_p95'], r['avg_launch_ms'], r['avg_launch_ms_alone']))"; }
for v in new old new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/knn_old.so $LIB/libml3d_hip.so; fi
  echo "knn $v: $(run)"
done > $O/knn_ab.log 2>&1
cp /tmp/new.so $LIB/libml3d_hip.so
cat $O/knn_ab.log
for g in 9 1 17 -1; do
  echo "new, search gate $g: $(ML3D_SEARCH_GATE=$g run)"
done > $O/gate.log 2>&1
cat $O/gate.log
