#!/bin/bash
# round 6, call AF: the whole GPU suite eight times on one box (hunting intermittent failures after the pipeline race of §11.11)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6af
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 1200 python -m pytest tests -m gpu -q --tb=short > $O/run_$i.log 2>&1
  tail -1 $O/run_$i.log | cut -c1-200
done | tee $O/summary.log
grep -l "failed" $O/run_*.log | head
