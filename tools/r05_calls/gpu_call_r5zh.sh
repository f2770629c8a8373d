#!/bin/bash
# round 5: the training side on csrc/train.hip -- GPU tests (ops against torch's autograd, both models against the reference goldens on both
# ML3D_TRAIN_OPS paths), step time + peak memory A/B at the YAML sizes; the PointPillars tests touched by the NMS change
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5zh
mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_training.py -q 2>&1 | tail -12 ) > $O/pytest_training.log; cat $O/pytest_training.log | cut -c1-300
( timeout 200 python tools/train_step_ab.py randlanet 4 2>&1 | tail -6 ) > $O/train_ab_randlanet.log; cat $O/train_ab_randlanet.log
( timeout 200 python tools/train_step_ab.py kpconv 8 2>&1 | tail -6 ) > $O/train_ab_kpconv.log; cat $O/train_ab_kpconv.log
( timeout 300 python -m pytest tests/test_gpu_pointpillars.py -q -k "two_lane or both_conv" 2>&1 | tail -5 ) > $O/pytest_pp.log; cat $O/pytest_pp.log | cut -c1-300
