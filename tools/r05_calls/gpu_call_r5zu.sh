#!/bin/bash
# round 5: micro-benchmark of DESIGN 9.10 -- per-lane vertex lists in lane-interleaved LDS against the same lists in registers, under four co-runners
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r5zu
mkdir -p $O
( timeout 60 tools/micro/lds_corun 2>&1 | tail -24 ) > $O/lds_corun.log; cat $O/lds_corun.log
