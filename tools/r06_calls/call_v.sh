#!/bin/bash
# round 6, call V: radius_expand / gemm_reduce / gather_pool_v4 without a 64-bit division per element: GPU tests, KPConv A/B against the previous commit
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6v
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_prims.py tests/test_gpu_kpconv.py tests/test_gpu_configs.py -q -k "radius or kp or KP" 2>&1 | tail -3 ) | cut -c1-300 | tee $O/tests.log
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
for v in base prev base prev; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  echo "$v $(timeout 120 python tools/roofline_ops.py radius 40 2>/dev/null | head -1)"
done | tee $O/alone.log
for v in base prev base prev; do
  if [ $v = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  ( timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=[x for x in d.get('roofline_other', []) if 'a10' in x.get('component','')][0]
print('$v', 'spheres/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), 'a10 in step %.3f alone %.3f frac_alone %.4f' % (e['avg_launch_ms'], e['avg_launch_ms_alone'], e['frac_alone']), d.get('pipeline_matches_quiet_run',{}).get('logits_max_delta'))"
done 2>&1 | tee $O/ab.log
cp /tmp/base.so $LIB/libml3d_hip.so
rm -rf /tmp/prof_kp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline --no-latency > /tmp/prof_kp.log 2>&1)
cp $(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1) $O/kp_kernel_stats.csv
head -14 $O/kp_kernel_stats.csv | cut -c1-150
