#!/bin/bash
# round 6, call W: the round's evidence -- whole GPU suite, smoke(), kernel tables + PMC passes + traffic.json (tools/gpu_round_profiles.sh), the three bench lines
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6w
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $O/smoke.log
timeout 1500 bash tools/gpu_round_profiles.sh r6w > $O/profiles.log 2>&1
tail -25 $O/profiles.log | cut -c1-200
( timeout 600 python bench.py 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-600 $O/bench.json
( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>$O/bench_kp.err | tail -1 ) > $O/bench_kp.json
cut -c1-300 $O/bench_kp.json
( timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>$O/bench_pp.err | tail -1 ) > $O/bench_pp.json
cut -c1-300 $O/bench_pp.json
