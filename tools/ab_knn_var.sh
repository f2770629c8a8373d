#!/bin/bash
# A/B helper (GPU box): kNN pyramid launch times (bench.py --breakdown knn:* tags) for each variant .so ("base" = in-tree build)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LIB=$ROOT/open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/variants/$v.so $LIB/libml3d_hip.so; fi
  python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-overlap --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']
ks=[k for k in b if k.startswith('knn:')]
print('$v', 'frames/s %.0f' % d['value'], ' '.join('%s=%.3f'%(k[4:],b[k]) for k in ks[:7]))
"
done
cp /tmp/base.so $LIB/libml3d_hip.so
