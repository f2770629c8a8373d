#!/bin/bash
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
for w in 1 0; do
  ( STATIC_WS=$w timeout 120 python tools/r05_calls/debug_graphs3.py pyramid 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/part_pyramid_ws$w.log 2>&1
  echo "== pyramid static_ws=$w"; cat $O/part_pyramid_ws$w.log | cut -c1-200
done
