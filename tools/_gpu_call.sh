cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p $O
LIB=open3d-ml_amd/ml3d/lib
( timeout 200 python -m pytest tests/test_gpu_api.py::test_device_resident_patch_loop_equals_the_host_loop_at_the_yaml_size -x -q 2>&1 | tail -4 ) > $O/t1.log 2>&1
cp $LIB/libml3d_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/mean_old.so $LIB/libml3d_hip.so; fi
  timeout 120 python tools/latency_only.py 200 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v  B=1 median %.3f p95 %.3f ms   B=4 median %.3f p95 %.3f ms per frame' % (d['batch_1']['ms_per_frame_median'], d['batch_1']['ms_per_frame_p95'], d['batch_4']['ms_per_frame_median'], d['batch_4']['ms_per_frame_p95']))" >> $O/ab.log 2>&1
done
cp /tmp/new.so $LIB/libml3d_hip.so
cat $O/t1.log $O/ab.log
