cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3ah; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do for d in 0 1 2 3; do
  ML3D_PP_DUMMY_STREAMS=$d timeout 150 python bench.py --workload pointpillars --no-cpu-baseline --steps 30 --warmup 8 > $O/pp_${d}_$i.json 2> $O/pp_${d}_$i.err
  echo "dummy=$d run $i: $(python -c "import json; d=json.load(open('$O/pp_${d}_$i.json')); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
done; done
