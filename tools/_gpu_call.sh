cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pointpillars.py tests/test_gpu_randlanet.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for d in 1 2 1 2; do
  for w in kpconv pointpillars; do
    ML3D_GEMM_DEPTH=$d timeout 120 python bench.py --workload $w --no-cpu-baseline > $O/${w}_$d.json 2> $O/${w}_$d.err
    echo "$w depth=$d: $(python -c "import json; d=json.load(open('$O/${w}_$d.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
  done
  ML3D_GEMM_DEPTH=$d timeout 120 python bench.py --no-workloads --no-cpu-baseline --no-latency > $O/rl_$d.json 2> $O/rl_$d.err
  echo "randla depth=$d: $(python -c "import json; d=json.load(open('$O/rl_$d.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
done
