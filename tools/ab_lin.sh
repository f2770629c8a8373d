#!/bin/bash
# GPU box helper: per-tag times of the Linear / pool / decoder launches of the RandLA forward (bench.py --breakdown)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in "$@"; do
  envs=""; if [ "$v" != "-" ]; then envs=$(echo $v | tr ',' ' '); fi
  env $envs python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-overlap --breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['breakdown_ms']
lin=[k for k in b if k.startswith('fwd:') and (int(k[4:])>=1000 or int(k[4:])%8 in (0,3,4,5,6))]
print('$v', 'frames/s %.0f' % d['value'], ' '.join('%s=%.3f'%(k[4:],b[k]) for k in lin), 'lin_sum=%.3f'%sum(b[k] for k in lin if k not in ('fwd:1201','fwd:1202')))
"
done
