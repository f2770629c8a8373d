cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3am; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/pf$i -o pp -- python $R/bench.py --workload pointpillars --no-cpu-baseline --steps 30 --warmup 8 > $R/$O/pp_$i.json 2> /dev/null
  cd $R
  python profiles/summarize_rocpd.py $O/pf$i/pp_results.db > $O/pp_$i.csv 2>/dev/null
  echo "run $i: $(python -c "import json; d=json.loads([l for l in open('$O/pp_$i.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['step_ms_median'],3))" 2>&1 | tail -1)"
  head -4 $O/pp_$i.csv | tail -3 | cut -d, -f1-4 | cut -c1-60,120-200
  rm -rf $O/pf$i
done
