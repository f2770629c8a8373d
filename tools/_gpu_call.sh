# Round 3, call C: fixed tests, full default bench with watchdog, latency profile, pipelines log, rocprof tables.
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pipelines.py tests/test_gpu_pointpillars.py -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 bash tools/gpu_ref_pipelines.sh > $O/pipelines.log 2>&1 < /dev/null; tail -8 $O/pipelines.log
ML3D_BENCH_PROFILE=1 timeout 120 python bench.py --no-workloads --no-cpu-baseline --steps 5 > $O/bench_lat.json 2> $O/bench_lat.err; tail -c 700 $O/bench_lat.json; grep -A30 "batch 4" $O/bench_lat.err | head -45
ML3D_BENCH_WATCHDOG=90 timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json; tail -40 $O/bench.err
