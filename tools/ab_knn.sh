#!/bin/bash
# GPU box helper: kNN pyramid launch times (bench.py --breakdown knn:* tags) under each env setting
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in "$@"; do
  envs=""; if [ "$v" != "-" ]; then envs=$(echo $v | tr ',' ' '); fi
  env $envs python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-overlap --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']
ks=[k for k in b if k.startswith('knn:')]
print('$v', 'frames/s %.0f' % d['value'], ' '.join('%s=%.3f'%(k[4:],b[k]) for k in ks[:7]))
"
done
