#!/bin/bash
# round 5, re-entry call: full GPU suite + smoke, the default bench line at HEAD (bf16x3 convolutions), kernel tables of the two side workloads
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5za
mkdir -p $O
rm -f gpurun_out/parity_per_yaml.jsonl
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
python tools/parity_table.py gpurun_out/parity_per_yaml.jsonl > $O/parity_per_yaml.md 2>&1
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-300 $O/bench.json
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$name -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/prof_$name.log 2>&1)
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $O/r05_${name}_kernel_stats.csv
  tail -1 /tmp/prof_$name.log | cut -c1-200
}
stats pp --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline --no-latency
stats kp --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency
