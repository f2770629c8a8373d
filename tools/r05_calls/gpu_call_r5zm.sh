#!/bin/bash
# round 5: training ops v4 (VGPR budget 256, no scratch, 8 weight rows in flight) (attention backward: LDS per width class, up to 2048 persistent workgroups, tree sum of the private partials; gemm_tn
# with atomics over <= 2048 waves): GPU tests, step time A/B, kernel table
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zm
mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_training.py -q 2>&1 | tail -4 ) > $O/pytest_training.log; cat $O/pytest_training.log | cut -c1-300
( timeout 200 python tools/train_step_ab.py randlanet 4 torch,hip,torch,hip 2>&1 | grep -v "return float" | tail -5 ) > $O/train_ab_randlanet.log; cat $O/train_ab_randlanet.log
rm -rf /tmp/kt; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_ab.py randlanet 4 hip > /tmp/kt.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_train_randlanet_hip_kernel_stats.csv
head -12 $O/r05_train_randlanet_hip_kernel_stats.csv | cut -c1-200
