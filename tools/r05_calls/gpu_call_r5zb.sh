#!/bin/bash
# round 5: the two-lane PointPillars test that failed at HEAD in r5za: diagnosis (batch independence of the head maps on both
# matrix pipes, which lane configuration differs) + the GPU tests that -x had not reached
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zb
mkdir -p $O
( timeout 150 python tools/r05_calls/diag_two_lane.py 2>&1 | tail -60 ) > $O/diag_bf16x3.log
( ML3D_PP_CONV=f32 timeout 150 python tools/r05_calls/diag_two_lane.py 2>&1 | tail -40 ) > $O/diag_f32.log
( timeout 400 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_prims.py tests/test_gpu_randlanet.py tests/test_gpu_training.py -q --deselect tests/test_gpu_pointpillars.py::test_two_lane_stream_returns_the_single_lane_detections 2>&1 | tail -15 ) > $O/pytest_rest.log
cat $O/diag_bf16x3.log | cut -c1-250; echo ---; cat $O/diag_f32.log | cut -c1-250 | tail -25; echo ---; tail -5 $O/pytest_rest.log
