#!/bin/bash
# round 5: training ops v2 (attention backward without grad_enc / grad_weight atomics, gemm_tn with per-slice partials): GPU tests,
# step time A/B, kernel table of the RandLANet step
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zk
mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_training.py -q 2>&1 | tail -4 ) > $O/pytest_training.log; cat $O/pytest_training.log | cut -c1-300
( timeout 200 python tools/train_step_ab.py randlanet 4 torch,hip,torch,hip 2>&1 | grep -v "return float" | tail -5 ) > $O/train_ab_randlanet.log; cat $O/train_ab_randlanet.log
( timeout 200 python tools/train_step_ab.py kpconv 8 torch,hip 2>&1 | grep -v "return float" | tail -3 ) > $O/train_ab_kpconv.log; cat $O/train_ab_kpconv.log
rm -rf /tmp/kt; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_ab.py randlanet 4 hip > /tmp/kt.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_train_randlanet_hip_kernel_stats.csv
head -14 $O/r05_train_randlanet_hip_kernel_stats.csv | cut -c1-200
( timeout 200 python -m pytest tests/test_gpu_pointpillars.py -q -k "both_conv" 2>&1 | tail -3 ) > $O/pytest_pp.log; cat $O/pytest_pp.log | cut -c1-300
