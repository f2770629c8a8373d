#!/bin/bash
# round 5: which co-runner disturbs the decode tail (both matrix pipes)
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zd
mkdir -p $O
( timeout 120 python tools/r05_calls/diag_decode_corun.py 2>&1 | tail -30 ) > $O/corun_bf16x3.log
( ML3D_PP_CONV=f32 timeout 120 python tools/r05_calls/diag_decode_corun.py 2>&1 | tail -30 ) > $O/corun_f32.log
cut -c1-300 $O/corun_bf16x3.log; echo ---; cut -c1-300 $O/corun_f32.log
