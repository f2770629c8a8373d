# Round 3, call E: tests after the one-pass radius / split decoder / parallel NMS mask / NHWC decode; A/B of the KPConv knobs.
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pointpillars.py tests/test_gpu_pipelines.py tests/test_gpu_prims.py tests/test_gpu_randlanet.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
  ML3D_RADIUS_ONE_PASS=$1 ML3D_KP_DECODER_SPLIT=$2 timeout 120 python bench.py --workload kpconv --no-cpu-baseline > $O/kp_$1$2.json 2> $O/kp_$1$2.err
  echo "kpconv one_pass=$1 dec_split=$2: $(python -c "import json; d=json.load(open('$O/kp_$1$2.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
done
timeout 120 python bench.py --workload pointpillars --no-cpu-baseline > $O/pp.json 2> $O/pp.err; echo "pointpillars: $(python -c "import json; d=json.load(open('$O/pp.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
timeout 120 python bench.py --no-workloads --no-cpu-baseline --no-latency > $O/rl.json 2> $O/rl.err; echo "randla: $(python -c "import json; d=json.load(open('$O/rl.json')); print(round(d['value'],1), round(d['step_ms_median'],3), d['roofline']['avg_launch_ms'])" 2>&1 | tail -1)"
