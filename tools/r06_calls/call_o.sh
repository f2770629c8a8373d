#!/bin/bash
# round 6, call O: voxelize on the bitmap path: GPU tests, then PointPillars A/B against the sort path
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6o
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_prims.py tests/test_gpu_pointpillars.py tests/test_gpu_corun.py tests/test_gpu_configs.py -q -k "vox or illar or pp or PointPillars or pointpillars or sort_scatter" 2>&1 | tail -3 ) | cut -c1-200
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base vox_sort base vox_sort; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  ( timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/pp_$v.json
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6o/pp_%s.json' % sys.argv[1]).read())
e=[x for x in d.get('roofline_other', []) if 'a15' in x.get('component','')][0]
print(sys.argv[1], 'frames/s %.0f' % d['value'], 'a15 in step %.3f ms alone %.3f ms frac_alone %.4f' % (e['avg_launch_ms'], e['avg_launch_ms_alone'], e['frac_alone']), d.get('pipeline_matches_quiet_run',{}).get('sweeps_with_identical_labels'))
PY
done
cp /tmp/base.so $LIB/libml3d_hip.so
