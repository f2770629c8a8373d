cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out/r3b
timeout 250 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_prims.py tests/test_gpu_api.py -x -q > gpurun_out/r3b/pytest.log 2>&1; tail -3 gpurun_out/r3b/pytest.log
for f in "" "--no-overlap" "" "--no-overlap"; do timeout 100 python bench.py --workload kpconv --steps 20 --warmup 3 --no-cpu-baseline $f 2>gpurun_out/r3b/err.log < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('kpconv $f %.0f /s  step %.3f ms block %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
except Exception as e: print('FAILED', e)"; done
tail -3 gpurun_out/r3b/err.log
