cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out/r3e
timeout 200 python -m pytest tests/test_gpu_kpconv.py -x -q > gpurun_out/r3e/pytest.log 2>&1; tail -3 gpurun_out/r3e/pytest.log
for f in "--frames-per-step 32" "" "--frames-per-step 32" ""; do timeout 100 python bench.py --workload kpconv --steps 20 --warmup 3 --no-cpu-baseline $f 2>gpurun_out/r3e/err.log < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('kpconv $f %.0f /s  step %.3f ms block %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
except Exception as e: print('FAILED', e)"; done
tail -3 gpurun_out/r3e/err.log
