#!/bin/bash
# round 6, call Y: flakiness check -- the whole GPU suite twice more on a fresh box, the two-lane PointPillars / co-run tests five times
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6y
mkdir -p $O
for i in 1 2; do ( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2 ) | cut -c1-200; done | tee $O/suite.log
for i in 1 2 3 4 5; do ( timeout 600 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_corun.py -q -k "lane or vox or sort_scatter or pipeline" 2>&1 | tail -1 ) | cut -c1-200; done | tee $O/repeat.log
