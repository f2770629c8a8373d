#!/bin/bash
# round 5: kernel table of the RandLANet training step on the HIP ops at HEAD
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zx
mkdir -p $O
rm -rf /tmp/kt; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_ab.py randlanet 4 hip > /tmp/kt.log 2>&1)
tail -1 /tmp/kt.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_train_randlanet_hip_kernel_stats.csv
head -16 $O/r05_train_randlanet_hip_kernel_stats.csv | cut -c1-180
