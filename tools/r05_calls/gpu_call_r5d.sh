#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
for b in 1 2; do
  rm -rf /tmp/tr_$b
  (cd /tmp && ML3D_KP_BUILDERS=$b rocprofv3 --kernel-trace -f csv -d /tmp/tr_$b -- python $GRAFT_REPO_ROOT/tools/kp_steps.py 24 > /tmp/tr_$b.log 2>&1)
  tail -2 /tmp/tr_$b.log
  python tools/trace_timeline.py /tmp/tr_$b 0.5 > $O/timeline_builders_$b.log 2>&1
  cat $O/timeline_builders_$b.log
done
