#!/bin/bash
# round 5: the side workloads' pipeline self-check (last seconds of the GPU budget)
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r5zt
mkdir -p $O
( timeout 40 python bench.py --workload pointpillars --steps 3 --warmup 1 --no-cpu-baseline --no-latency 2>$O/pp.err | tail -1 ) > $O/pp.json
python -c "
import json; d=json.load(open('$O/pp.json')); print(d['value'], d.get('pipeline_matches_quiet_run'))"
