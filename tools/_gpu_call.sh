cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
rm -f gpurun_out/r2v/ab2.log
for rep in 1 2; do
for e in "ML3D_GEMM_BIG_ROWS=0" "ML3D_GEMM_BIG_ROWS=1"; do
  env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$e randla frames/s %.0f median %.3f' % (d['value'], d['step_ms_median']))" >> gpurun_out/r2v/ab2.log
done
done
for rep in 1 2 3; do
for e in "ML3D_GEMM_BIG_ROWS=0" "ML3D_GEMM_BIG_ROWS=1" "ML3D_GEMM_BIG_ROWS=1 ML3D_GEMM_BIG_MIN_K=0"; do
  env $e timeout 300 python bench.py --workload kpconv --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$e kpconv spheres/s %.0f block %.3f ms frac %.3f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))" >> gpurun_out/r2v/ab2.log
done
done
for e in "ML3D_GEMM_BIG_ROWS=0" "ML3D_GEMM_BIG_ROWS=1"; do
  env $e timeout 300 python bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$e pointpillars frames/s %.0f' % d['value'])" >> gpurun_out/r2v/ab2.log
done
cat gpurun_out/r2v/ab2.log
