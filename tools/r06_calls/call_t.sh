#!/bin/bash
# round 6, call T: pillar_pfn_v4 (three dependent round trips instead of eight, no LDS): GPU tests, op alone A/B, PointPillars A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6t
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_prims.py tests/test_gpu_configs.py -q -k "illar or pp or PointPillars or pointpillars" 2>&1 | tail -3 ) | cut -c1-300 | tee $O/tests.log
for v in 1 0 1 0; do
  echo "v4=$v $(ML3D_PFN_V4=$v timeout 120 python tools/roofline_ops.py pillars 40 2>/dev/null | head -1)"
done | tee $O/alone.log
rm -rf /tmp/prof_pil
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_pil -o pil -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py pillars 20 > /tmp/prof_pil.log 2>&1)
cp $(find /tmp/prof_pil -name "*kernel_stats.csv" | head -1) $O/pillars_kernel_stats.csv
head -4 $O/pillars_kernel_stats.csv | cut -c1-160
for v in 1 0 1 0; do
  ( ML3D_PFN_V4=$v timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/pp_$v.json
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6t/pp_%s.json' % sys.argv[1]).read())
e=[x for x in d.get('roofline_other', []) if 'a16' in x.get('component','')][0]
print('v4=%s' % sys.argv[1], 'frames/s %.0f' % d['value'], 'a16-a17 in step %.3f ms alone %.3f ms frac_alone %.4f' % (e['avg_launch_ms'], e['avg_launch_ms_alone'], e['frac_alone']), d.get('pipeline_matches_quiet_run',{}).get('sweeps_with_identical_labels'))
PY
done 2>&1 | tee $O/ab.log
