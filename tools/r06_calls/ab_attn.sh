#!/bin/bash
# A/B helper (GPU box): for each library variant run bench.py --breakdown; prints frames/s + the attention tags (fwd:8l+1 / 8l+2)
# usage: tools/r06_calls/ab_attn.sh base attn_f32 ...   ("base" = the in-tree build; others = ml3d/lib/ab/<name>.so)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
LIB=$ROOT/open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  timeout 200 python $ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --breakdown 2>/tmp/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d.get('breakdown_ms',{})
keys=['fwd:1','fwd:2','fwd:9','fwd:10','fwd:17','fwd:18','fwd:25','fwd:26']
print('$v', 'frames/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), ' '.join('%s=%.3f'%(k,b.get(k,0)) for k in keys), 'attn_sum=%.3f'%sum(b.get(k,0) for k in keys), 'agree', d.get('label_agreement'))
" || tail -3 /tmp/ab_err.log
done
cp /tmp/base.so $LIB/libml3d_hip.so
