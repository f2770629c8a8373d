#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5o
mkdir -p $O
for a in "1 128" "2 128" "1 128" "2 128" "2 256" "3 192" "1 256"; do
  timeout 300 python tools/randla_lanes.py $a 2>/dev/null | tail -1
done > $O/lanes.log 2>&1
cat $O/lanes.log
