#!/bin/bash
# round 6, call K: argmax A/B, the DDP training step at N = 1, the whole GPU suite
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6k
mkdir -p $O
( tools/r06_calls/ab_knn.sh base argmax_naive base argmax_naive 2>&1 ) > $O/ab.log; cat $O/ab.log | cut -c1-200
( timeout 300 python bench.py --train --steps 5 --warmup 2 2>$O/train.err | tail -1 ) > $O/train.json; cut -c1-600 $O/train.json; tail -3 $O/train.err | cut -c1-300
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
