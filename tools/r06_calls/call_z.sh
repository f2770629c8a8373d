#!/bin/bash
# round 6, call Z: gemm_tile_bf3 does not multiply a wave's all-padding column tiles (the heads' N = 72 in a 128-wide tile): GPU tests, PointPillars + KPConv A/B against the previous commit
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6z
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_configs.py tests/test_gpu_kpconv.py -q 2>&1 | tail -2 ) | cut -c1-300 | tee $O/tests.log
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base prev base prev base prev; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  ( timeout 300 python bench.py --workload pointpillars --steps 60 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'frames/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), d.get('pipeline_matches_quiet_run',{}).get('sweeps_with_identical_labels'))"
done 2>&1 | tee $O/ab.log
cp /tmp/base.so $LIB/libml3d_hip.so
