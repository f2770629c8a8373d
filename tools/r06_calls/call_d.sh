#!/bin/bash
# round 6, call D: the whole GPU suite (bf16x3 attention kernels, per-item subsampling), kernel tables of the RandLA and KPConv steps, bench lines
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6d
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
for w in randla kp; do
  rm -rf /tmp/prof_$w
  if [ $w = randla ]; then A="--steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-latency"; else A="--workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency"; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py $A > /tmp/prof_$w.log 2>&1)
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $O/r06_${w}_kernel_stats.csv
  tail -1 /tmp/prof_$w.log | cut -c1-200
done
head -24 $O/r06_randla_kernel_stats.csv | cut -c1-150
head -30 $O/r06_kp_kernel_stats.csv | cut -c1-150
( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>$O/bench_kp.err | tail -1 ) > $O/bench_kp.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6d/bench_kp.json').read())
print('kpconv', d.get('value'), d.get('ms_per_step'), d.get('pipeline_matches_quiet_run'), d.get('error'))
for e in d.get('roofline_other', []): print(' ', e.get('component','')[:40], e.get('ms_in_step'), e.get('ms_alone'), e.get('frac'), e.get('frac_alone'))
PY
( timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-500 $O/bench.json
