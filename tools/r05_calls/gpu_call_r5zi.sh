#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zi
mkdir -p $O
( timeout 100 python tools/r05_calls/diag_two_pipes.py 2>&1 | tail -10 ) > $O/two_pipes.log; cut -c1-600 $O/two_pipes.log
