cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
LIB=open3d-ml_amd/ml3d/lib
cp $LIB/libml3d_hip.so /tmp/base.so
timeout 200 python -m pytest tests/test_gpu_knn.py tests/test_gpu_randlanet.py tests/test_gpu_kpconv.py -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in base knn_branchy base knn_branchy; do
  if [ "$v" = base ]; then cp /tmp/base.so $LIB/libml3d_hip.so; else cp $LIB/ab/$v.so $LIB/libml3d_hip.so; fi
  timeout 120 python bench.py --no-workloads --no-cpu-baseline --no-latency > $O/rl_$v.json 2> $O/rl_$v.err
  echo "randla $v: $(python -c "import json; d=json.load(open('$O/rl_$v.json')); print(round(d['value'],1), round(d['step_ms_median'],3), [ (r['kernel'][:18], round(r['avg_launch_ms'],3)) for r in [d['roofline']]+d['roofline_other']])" 2>&1 | tail -1)"
  timeout 60 python tools/knn_only.py 5 2>/dev/null | tail -2
done
cp /tmp/base.so $LIB/libml3d_hip.so
timeout 120 python bench.py --workload kpconv --no-cpu-baseline > $O/kp.json 2> $O/kp.err; echo "kpconv: $(python -c "import json; d=json.load(open('$O/kp.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
