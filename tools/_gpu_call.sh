# Round 3, call F: knn store-mode A/B, KPConv with the shared conv/pool grid, PointPillars GEMM tile A/B, new tests.
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_prims.py tests/test_gpu_knn.py tests/test_gpu_kpconv.py tests/test_gpu_randlanet.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for m in 2 0 1 2; do
  ML3D_KNN_STORE=$m timeout 120 python bench.py --no-workloads --no-cpu-baseline --no-latency > $O/rl_$m.json 2> $O/rl_$m.err
  echo "randla knn_store=$m: $(python -c "import json; d=json.load(open('$O/rl_$m.json')); print(round(d['value'],1), round(d['step_ms_median'],3), [ (r['kernel'][:18], round(r['avg_launch_ms'],3)) for r in [d['roofline']]+d['roofline_other']])" 2>&1 | tail -1)"
done
timeout 120 python bench.py --workload kpconv --no-cpu-baseline > $O/kp.json 2> $O/kp.err; echo "kpconv: $(python -c "import json; d=json.load(open('$O/kp.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
for v in "0 256" "1 256" "1 384"; do set -- $v
  ML3D_GEMM_BIG_ROWS=$1 ML3D_GEMM_BIG_MIN_K=$2 timeout 120 python bench.py --workload pointpillars --no-cpu-baseline > $O/pp_$1_$2.json 2> $O/pp_$1_$2.err
  echo "pointpillars big_rows=$1 min_k=$2: $(python -c "import json; d=json.load(open('$O/pp_$1_$2.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
done
