#!/bin/bash
# round 6, call A: GPU suite at HEAD with the bench-configuration tests (128 frames / 96 spheres) and the co-run stress, the inactive-lane
# micro-experiment, a baseline bench line of this round's box
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_corun.py -q -x 2>&1 | tail -15 ) > $O/corun.log 2>&1; cat $O/corun.log | cut -c1-300
( timeout 120 tools/micro/lds_corun 2>&1 | tail -30 ) > $O/lds_corun.log; cat $O/lds_corun.log
( timeout 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_corun.py 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-250
( timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-400 $O/bench.json
