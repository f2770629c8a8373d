cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
for l in 1 2 4 1 2; do
  ML3D_PP_LANES=$l timeout 120 python bench.py --workload pointpillars --no-cpu-baseline > $O/pp_$l.json 2> $O/pp_$l.err
  echo "pointpillars lanes=$l: $(python -c "import json; d=json.load(open('$O/pp_$l.json')); print(round(d['value'],1), round(d['ms_per_step'],3))" 2>&1 | tail -1)"
done
