cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3aj; mkdir -p $O
export TMPDIR=/tmp
for spec in "" 16 8 24 0x55555555 0x0000ffff "" 12; do
  ML3D_SEARCH_CUS=$spec timeout 200 python bench.py --no-cpu-baseline --no-workloads --no-latency > $O/rl_$spec.json 2> $O/rl_$spec.err
  echo "cus='$spec': $(python -c "import json; d=json.load(open('$O/rl_$spec.json')); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],3), r['kernel'][:20], round(r['avg_launch_ms'],3))" 2>&1 | tail -1)"
done
