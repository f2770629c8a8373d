#!/bin/bash
# round 5: localise the rare two-lane mismatch (forward / decode / hand-over)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zc
mkdir -p $O
( timeout 200 python tools/r05_calls/diag_two_lane2.py 2>&1 | tail -60 ) > $O/diag2.log
cut -c1-400 $O/diag2.log
