cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r4x
mkdir -p $O
( timeout 45 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_prims.py -x -q -k "bboxes or nms or iou or topk or 16_sweeps or two_lane" 2>&1 | tail -4 ) > $O/t1.log 2>&1
cat $O/t1.log
