cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
for i in 1 2; do for sk in "" 1; do
  echo "skip='$sk': $(ML3D_X_SKIP_ARGMAX=$sk timeout 200 python bench.py --no-cpu-baseline --no-workloads --no-latency --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['step_ms_median'],3))")"
done; done
