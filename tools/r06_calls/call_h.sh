#!/bin/bash
# round 6, call H: k-NN hand-off (3x3x3 block for all, further shells for the compacted open queries): exactness tests + A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6h
mkdir -p $O
( timeout 500 python -m pytest tests/test_gpu_knn.py tests/test_gpu_randlanet.py tests/test_gpu_prims.py tests/test_gpu_corun.py -q -x 2>&1 | tail -4 ) > $O/pytest.log 2>&1
cat $O/pytest.log | cut -c1-300
( tools/r06_calls/ab_knn.sh base knn_onelaunch knn_hand5 base knn_onelaunch 2>&1 ) > $O/ab.log; cat $O/ab.log | cut -c1-200
