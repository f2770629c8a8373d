cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3al; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 200 python bench.py --no-workloads --no-latency --steps 20 --warmup 5 > $O/rl_$i.json 2> $O/rl_$i.err
  echo "run $i: $(python -c "import json; d=json.load(open('$O/rl_$i.json')); r=d['roofline']; c=d.get('cpu_baseline') or {}; print(round(d['value'],1), round(d['step_ms_median'],3), round(r['avg_launch_ms'],3), c.get('label_agreement'), c.get('miou_vs_cpu_oracle'))" 2>&1 | tail -1)"
done
