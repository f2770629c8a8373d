#!/bin/bash
# round 6, call N: pillar_pfn over the real rows only: PointPillars GPU tests + the PointPillars / KPConv bench lines
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6n
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_configs.py tests/test_gpu_pipelines.py -q -k "illar or pp or PointPillars or pointpillars" 2>&1 | tail -3 ) | cut -c1-200
( timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>$O/pp.err | tail -1 ) > $O/bench_pp.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6n/bench_pp.json').read())
print('pointpillars', d.get('value'), d.get('ms_per_step'), d.get('pipeline_matches_quiet_run'))
for e in d.get('roofline_other', []): print(' ', e.get('component','')[:34], 'ms', e.get('avg_launch_ms'), 'alone', e.get('avg_launch_ms_alone'), 'frac', e.get('frac'), 'frac_alone', e.get('frac_alone'))
PY
