#!/bin/bash
# round 5: per-kernel times of the KPFCNN forward (64 spheres) with the contractions on the bf16x3 kernel
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5bp
mkdir -p $O
rm -rf /tmp/kt
(cd /tmp && timeout 250 rocprofv3 --kernel-trace -f csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/kp_forward_seq.py 3 > /tmp/kt.log 2>&1)
tail -2 /tmp/kt.log
python tools/trace_sequence.py /tmp/kt kp_small_fused 1 > $O/seq.log 2>&1 || true
head -80 $O/seq.log | cut -c1-150
