#!/bin/bash
# round 6, call AG: closing confirmation -- GPU suite twice, smoke, the three bench commands
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6ag
mkdir -p $O
for i in 1 2; do ( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -1 ) | cut -c1-200; done | tee $O/suite.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
( timeout 600 python bench.py 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-300 $O/bench.json
( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>$O/bench_kp.err | tail -1 ) > $O/bench_kp.json
cut -c1-300 $O/bench_kp.json
( timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>$O/bench_pp.err | tail -1 ) > $O/bench_pp.json
cut -c1-300 $O/bench_pp.json
