#!/bin/bash
# round 6, call AH: the multi-stream / multi-thread tests (pipelines, lanes, frame stream) fifteen times each
cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
O=gpurun_out/r6ah
mkdir -p $O
for i in $(seq 1 15); do
  timeout 600 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pointpillars.py tests/test_gpu_pipelines.py tests/test_gpu_randlanet.py tests/test_gpu_api.py -q --tb=short -k "pipeline or lane or stream or bench_configuration or in_flight or graph" > $O/run_$i.log 2>&1
  tail -1 $O/run_$i.log | cut -c1-200
done | tee $O/summary.log
