#!/bin/bash
# round 6, call E: sequential per-tag breakdown of the RandLA step (what is solo time, what is co-running interference), KPConv after the
# subsampling size classes, the k-NN / subsample GPU tests
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6e
mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_knn.py tests/test_gpu_prims.py tests/test_gpu_kpconv.py -q -x 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
( timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-workloads --no-latency --no-overlap --breakdown 2>$O/b1.err | tail -1 ) > $O/bench_seq.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e/bench_seq.json').read())
b=d.get('breakdown_ms',{})
print('sequential frames/s', d['value'], 'ms/step', d['ms_per_step'])
tot=0
for k,v in sorted(b.items(), key=lambda kv:-kv[1]):
    print('  %-12s %.3f' % (k, v)); tot+=v
print('  sum', tot)
PY
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency 2>$O/b2.err | tail -1 | cut -c1-300 ) > $O/bench_ovl.json; cat $O/bench_ovl.json
( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>$O/bench_kp.err | tail -1 ) > $O/bench_kp.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e/bench_kp.json').read())
print('kpconv', d.get('value'), d.get('ms_per_step'), d.get('step_ms_median'))
for e in d.get('roofline_other', []): print(' ', e.get('component','')[:30], 'frac', e.get('frac'), 'alone', e.get('frac_alone'), 'achieved', e.get('achieved'))
PY
