cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out/r3k
timeout 250 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_randlanet.py tests/test_gpu_pointpillars.py -x -q > gpurun_out/r3k/pytest.log 2>&1; tail -2 gpurun_out/r3k/pytest.log
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-workloads 2>gpurun_out/r3k/err.log < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('randla %.0f /s median %.3f' % (d['value'], d['step_ms_median']))
except Exception as e: print('randla FAILED', e)"
timeout 100 python bench.py --workload kpconv --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/r3k/err.log < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('kpconv %.0f /s  step %.3f ms' % (d['value'], d['ms_per_step']))
except Exception as e: print('FAILED', e)"
done
