#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
python tools/kp_forward_seq.py 20 > $O/alone.log 2>&1; cat $O/alone.log | tail -4
rm -rf /tmp/trf
(cd /tmp && rocprofv3 --kernel-trace -f csv -d /tmp/trf -- python $GRAFT_REPO_ROOT/tools/kp_forward_seq.py 4 > /tmp/trf.log 2>&1)
python tools/trace_sequence.py /tmp/trf kp_small_fused 1 > $O/forward_sequence.log 2>&1
head -120 $O/forward_sequence.log
