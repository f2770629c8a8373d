#!/bin/bash
# round 5: KPFCNN training step at HEAD (A/B + kernel table of the HIP path)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zw
mkdir -p $O
( timeout 100 python tools/train_step_ab.py kpconv 8 torch,hip,hip 2>&1 | grep -v "return float" | tail -4 ) > $O/train_ab_kpconv.log; cat $O/train_ab_kpconv.log
rm -rf /tmp/kt; (cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_ab.py kpconv 8 hip > /tmp/kt.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_train_kpconv_hip_kernel_stats.csv
head -10 $O/r05_train_kpconv_hip_kernel_stats.csv | cut -c1-170
