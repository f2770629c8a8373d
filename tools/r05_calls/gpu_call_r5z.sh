#!/bin/bash
# round 5, closing call: full GPU suite + smoke, the default bench line, the round's profile set at HEAD
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5z
mkdir -p $O
rm -f gpurun_out/parity_per_yaml.jsonl
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
python tools/parity_table.py gpurun_out/parity_per_yaml.jsonl > $O/parity_per_yaml.md 2>&1
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-300 $O/bench.json
timeout 1500 bash tools/gpu_round_profiles.sh r05 > $O/profiles.log 2>&1
tail -12 $O/profiles.log | cut -c1-300
