#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6p
mkdir -p $O
rm -rf /tmp/prof_vox
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_vox -o vox -- python $GRAFT_REPO_ROOT/tools/roofline_ops.py voxelize 4 > /tmp/prof_vox.log 2>&1)
cp $(find /tmp/prof_vox -name "*kernel_stats.csv" | head -1) $O/vox_kernel_stats.csv
head -24 $O/vox_kernel_stats.csv | cut -c1-160
