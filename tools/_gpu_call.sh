cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/knn_old.so $LIB/libml3d_hip.so; fi
  echo "== $v: $(python tools/knn_only.py 9 2>&1 | tail -1)"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   %.0f frames/s step %.2f ms' % (d['value'], d['ms_per_step']))"
done
cp /tmp/new.so $LIB/libml3d_hip.so
