#!/bin/bash
# round 6, call AA: the RCCL calls of the N > 1 path on real hardware from a one-GPU box (ML3D_DIST_FORCE_GROUP=1: process group of one over nccl,
# async gather of the labels, barriers, all-reduce of the time, all_gather of the checksums, gather_ragged of the side workloads)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6aa
mkdir -p $O
export ML3D_DIST_FORCE_GROUP=1
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-workloads --no-latency 2>$O/randla.err | tail -1 ) > $O/randla.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6aa/randla.json').read())
print('randla', d['value'], d.get('gather_self_check'), d['ranks_seen']['world_size'])
PY
tail -3 $O/randla.err
for w in kpconv pointpillars; do
  ( timeout 300 python bench.py --workload $w --steps 10 --warmup 4 --no-cpu-baseline 2>$O/$w.err | tail -1 ) > $O/$w.json
  python - $w <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6aa/%s.json' % sys.argv[1]).read())
print(sys.argv[1], d.get('value'), {k: d[k] for k in d if 'gather' in k}, d.get('error'))
PY
  tail -2 $O/$w.err
done
( timeout 300 python bench.py --train --steps 3 --warmup 1 2>$O/train.err | tail -1 ) | cut -c1-400
tail -2 $O/train.err
