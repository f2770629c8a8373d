#!/bin/bash
# A/B helper (GPU box): frames/s + the k-NN launch(es) in step / alone for each library variant
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
LIB=$ROOT/open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  timeout 200 python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency 2>/tmp/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=[d['roofline']]+d.get('roofline_other',[])
k=[e for e in r if 'knn' in e.get('kernel','')][0]
print('$v', 'frames/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), 'knn in step %.3f ms, alone %.3f ms' % (k['avg_launch_ms'], k.get('avg_launch_ms_alone') or 0), 'agree', d.get('label_agreement'))
" || tail -3 /tmp/ab_err.log
done
cp /tmp/base.so $LIB/libml3d_hip.so
