#!/bin/bash
# round 6, call J: the whole co-run stress at 300 repetitions, twice
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6j
mkdir -p $O
for i in 1 2; do ( timeout 600 python -m pytest tests/test_gpu_corun.py -q 2>&1 | grep -E "^FAILED|^E  .*Assertion|passed|failed" | head -12 ) ; done > $O/corun_300.log 2>&1
cat $O/corun_300.log | cut -c1-300
