cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out/r2y
L=gpurun_out/r2y/ab.log; : > $L
timeout 300 python -m pytest tests/test_gpu_prims.py tests/test_gpu_kpconv.py -x -q > gpurun_out/r2y/pytest.log 2>&1; tail -2 gpurun_out/r2y/pytest.log >> $L
one() {  # env-string workload steps
  env $1 timeout 120 python bench.py --workload $2 --steps $3 --warmup 3 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('$1 $2 %.0f /s  step %.3f ms' % (d['value'], d['ms_per_step']))
except Exception as e: print('$1 $2 FAILED', e)" >> $L
}
for rep in 1 2; do
one "ML3D_X=0" kpconv 20
one "ML3D_GEMM_SPLIT=256,512" kpconv 20
one "ML3D_KP_SMALL_FUSED=0" kpconv 20
done
one "ML3D_X=0" pointpillars 15
one "ML3D_GEMM_SPLIT=256,512" pointpillars 15
one "ML3D_X=0" pointpillars 15
one "ML3D_GEMM_SPLIT=256,512" pointpillars 15
for e in "ML3D_X=0" "ML3D_GEMM_SPLIT=256,512" "ML3D_X=0" "ML3D_GEMM_SPLIT=256,512"; do
  env $e timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-workloads 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('$e randla %.0f /s median %.3f' % (d['value'], d['step_ms_median']))
except Exception as e: print('$e randla FAILED', e)" >> $L
done
cat $L
