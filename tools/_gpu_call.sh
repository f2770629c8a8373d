cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
for v in "8 -1" "4 -1" "8 9" "8 17" "8 3" "8 -1"; do set -- $v
  GPU_MAX_HW_QUEUES=$1 ML3D_SEARCH_GATE=$2 timeout 120 python bench.py --no-workloads --no-cpu-baseline --no-latency > $O/rl_$1_$2.json 2> $O/rl_$1_$2.err
  echo "randla queues=$1 gate=$2: $(python -c "import json; d=json.load(open('$O/rl_$1_$2.json')); print(round(d['value'],1), round(d['step_ms_median'],3), [ (r['kernel'][:18], round(r['avg_launch_ms'],3)) for r in [d['roofline']]+d['roofline_other']])" 2>&1 | tail -1)"
done
