#!/bin/bash
# round 5: which kernel of the decode tail the co-running bf16x3 forward disturbs
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5ze
mkdir -p $O
( timeout 120 python tools/r05_calls/diag_decode_stage.py 2>&1 | tail -30 ) > $O/stage.log
cut -c1-600 $O/stage.log
