#!/bin/bash
# round 6, call R: fused voxelize (one launch per radix pass + one grouping launch): GPU tests, kernel table, PointPillars A/B against the launch chain
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6r
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_prims.py tests/test_gpu_pointpillars.py tests/test_gpu_corun.py tests/test_gpu_configs.py -q -k "vox or illar or pp or PointPillars or pointpillars or sort_scatter" 2>&1 | tail -5 ) | cut -c1-300 | tee $O/tests.log
rm -rf /tmp/prof_vox
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_vox -o vox -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py voxelize 20 > /tmp/prof_vox.log 2>&1)
tail -3 /tmp/prof_vox.log
cp $(find /tmp/prof_vox -name "*kernel_stats.csv" | head -1) $O/vox_kernel_stats.csv
head -16 $O/vox_kernel_stats.csv | cut -c1-200
for v in 1 0 1 0; do
  ( ML3D_VOX_FUSED=$v timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/pp_$v.json
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6r/pp_%s.json' % sys.argv[1]).read())
e=[x for x in d.get('roofline_other', []) if 'a15' in x.get('component','')][0]
print('fused=%s' % sys.argv[1], 'frames/s %.0f' % d['value'], 'a15 in step %.3f ms alone %.3f ms frac_alone %.4f' % (e['avg_launch_ms'], e['avg_launch_ms_alone'], e['frac_alone']), d.get('pipeline_matches_quiet_run',{}).get('sweeps_with_identical_labels'))
PY
done 2>&1 | tee $O/ab.log
