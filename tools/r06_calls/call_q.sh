#!/bin/bash
# round 6, call Q: KPFCNN's deep Linears (decoder steps with their gathered residual, unary / shortcut) on the bf16 pipe: GPU tests + A/B by K threshold
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6q
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_configs.py tests/test_gpu_pipelines.py tests/test_gpu_prims.py tests/test_gpu_pointpillars.py -q -k "kp or KP or vox or illar or pp" 2>&1 | tail -3 ) | cut -c1-200
run() { ( ML3D_KP_LINEAR_B3=$1 timeout 300 python bench.py --workload kpconv --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('K>=$1', 'spheres/s %.0f' % d['value'], 'step_med %.2f' % d.get('step_ms_median',0), d.get('pipeline_matches_quiet_run',{}).get('logits_max_delta'))"; }
for k in 256 0 128 512 256 0; do run $k; done > $O/ab.log 2>&1
cat $O/ab.log
