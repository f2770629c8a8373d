cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $LIB/libml3d_hip.so; else cp $LIB/ab/rl_old.so $LIB/libml3d_hip.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']; print('$v %.0f frames/s step %.2f ms; stage1 %.3f stage2 %.3f head %.3f' % (d['value'], d['ms_per_step'], b['fwd:1'], b['fwd:2'], b['fwd:1000']))"
done
cp /tmp/new.so $LIB/libml3d_hip.so
timeout 600 python -m pytest tests/test_gpu_randlanet.py -x -q 2>&1 | tail -2
