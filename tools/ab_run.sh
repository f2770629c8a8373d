#!/bin/bash
# A/B helper (GPU box): for each variant .so run bench.py --breakdown and print frames/s + the attention-kernel tags
# usage: tools/ab_run.sh base xv iglp ...     ("base" = the in-tree build)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LIB=$ROOT/open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/variants/$v.so $LIB/libml3d_hip.so; fi
  python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-overlap --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']
keys=['fwd:1','fwd:2','fwd:9','fwd:10','fwd:17','fwd:18','fwd:25','fwd:26']
print('$v', 'frames/s %.0f' % d['value'], ' '.join('%s=%.3f'%(k,b.get(k,0)) for k in keys), 'attn_sum=%.3f'%sum(b.get(k,0) for k in keys), 'fc1=%.3f'%b.get('fwd:1200',0))
"
done
cp /tmp/base.so $LIB/libml3d_hip.so
