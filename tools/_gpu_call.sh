cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p $O
# 1. parity of the new top-k and of everything downstream of it (PointPillars decode)
( timeout 400 python -m pytest tests/test_gpu_prims.py tests/test_gpu_pointpillars.py "tests/test_gpu_configs.py::test_pointpillars_yaml" \
    tests/test_gpu_pipelines.py::test_pointpillars_kitti_sweep_through_forward_and_inference_end_matches_the_reference_pipeline -x -q 2>&1 | tail -6 ) > $O/t1.log 2>&1
# 2. A/B inside the timed PointPillars step: ml3d_topk_rows vs torch.topk
for v in new torch new torch; do
  timeout 150 python tools/ab_topk.py $v 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v %.0f frames/s step median %.3f ms p95 %.3f; single sweep %.3f ms' % (d['value'], d['step_ms_median'], d['step_ms_p95'], d['latency_single_sweep_ms']['median']))" >> $O/ab.log 2>&1
done
# 3. the whole GPU suite
( timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/t2.log 2>&1
cat $O/t1.log $O/ab.log $O/t2.log
