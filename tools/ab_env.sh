#!/bin/bash
# A/B helper (GPU box): run bench.py --breakdown under each "NAME=VALUE[,NAME=VALUE...]" environment setting ("-" = none)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in "$@"; do
  envs=""; if [ "$v" != "-" ]; then envs=$(echo $v | tr ',' ' '); fi
  env $envs python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-overlap --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']
keys=['fwd:1','fwd:2','fwd:9','fwd:10','fwd:17','fwd:18','fwd:23','fwd:25','fwd:26','fwd:31']
print('$v', 'frames/s %.0f' % d['value'], ' '.join('%s=%.3f'%(k,b.get(k,0)) for k in keys), 'attn_sum=%.3f'%sum(b.get(k,0) for k in keys))
"
done
