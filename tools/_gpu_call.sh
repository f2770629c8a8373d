cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
for b in 256 128 64; do
  echo "== general kernel block=$b: $(ML3D_KNN_PYRAMID=0 ML3D_KNN_BLOCK=$b python tools/knn_only.py 7 2>&1 | tail -1)"
done
for b in 256 64 256 64; do
  echo "== bench block=$b: $(ML3D_KNN_PYRAMID=0 ML3D_KNN_BLOCK=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=[d['roofline']]+d['roofline_other']
print('%.0f frames/s step %.2f ms' % (d['value'], d['ms_per_step']), ' | '.join('%s %.3f ms (alone %.3f)' % (x['kernel'][:18], x['avg_launch_ms'], x['avg_launch_ms_alone'] or 0) for x in r))")"
done
