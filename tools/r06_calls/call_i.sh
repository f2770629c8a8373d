#!/bin/bash
# round 6, call I: rotated NMS without LDS under the bf16x3 co-runner (1000 repetitions x 3 processes), NMS parity tests
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6i
mkdir -p $O
for i in 1 2 3; do ( timeout 300 python -m pytest tests/test_gpu_corun.py -q -k "topk or nms" 2>&1 | grep -E "^FAILED|^E  .*Assertion|passed|failed" | head -12 ) ; done > $O/corun_after.log 2>&1
cat $O/corun_after.log | cut -c1-300
( timeout 300 python -m pytest tests/test_gpu_prims.py tests/test_gpu_api.py tests/test_gpu_pointpillars.py -q -k "nms or iou or bboxes or decode or two_lane" 2>&1 | tail -3 ) | cut -c1-200
