#!/bin/bash
# round 6, call AB: rad_scan with a 9-row specialisation (3 x 3 rows: 8 compare + select pairs per candidate instead of 15) vs the 16-row form only: op alone + KPConv A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6ab
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_prims.py tests/test_gpu_kpconv.py -q -k "radius or kp or KP" 2>&1 | tail -2 ) | cut -c1-200 | tee $O/tests.log
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base rows16 base rows16; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  echo "$v $(timeout 120 python tools/roofline_ops.py radius 40 2>/dev/null | head -1)"
done | tee $O/alone.log
for v in base rows16 base rows16; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  ( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=[x for x in d.get('roofline_other', []) if 'a10' in x.get('component','')][0]
print('$v', 'spheres/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), 'a10 alone %.3f' % e['avg_launch_ms_alone'])"
done 2>&1 | tee $O/ab.log
cp /tmp/base.so $LIB/libml3d_hip.so
