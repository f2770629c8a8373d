cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_pointpillars.py tests/test_gpu_pipelines.py -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
ML3D_GEMM_TAIL_DEBUG=1 timeout 60 python tools/roofline_ops.py pp 3 2>&1 | tail -3
for t in 1 0 1 0; do
  ML3D_GEMM_TAIL_SPLIT=$t timeout 120 python bench.py --workload pointpillars --no-cpu-baseline > $O/pp_$t.json 2> $O/pp_$t.err
  echo "pointpillars tail_split=$t: $(python -c "import json; d=json.load(open('$O/pp_$t.json')); print(round(d['value'],1), round(d['step_ms_median'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))" 2>&1 | tail -1)"
done
