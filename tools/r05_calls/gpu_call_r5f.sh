#!/bin/bash
# round 5: HIP graphs around the per-patch sequence of the model-class API -- parity, then the latency loop with graphs on / off
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_pipelines.py -x -q 2>&1 | tail -12 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for g in 1 0 1 0; do
  echo "graphs=$g: $(ML3D_RANDLA_GRAPHS=$g timeout 300 python tools/latency_only.py 200 2>$O/lat_$g.err | tail -1)"
done > $O/latency_ab.log 2>&1
cat $O/latency_ab.log
tail -5 $O/lat_1.err
