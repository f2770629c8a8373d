# Round 3, validation call: the whole GPU suite, smoke, the default bench line, rocprofv3 kernel tables of the three workloads and the
# PMC traffic passes behind profiles/r03_* / profiles/traffic.json.  Every command reads /dev/null and has its own timeout.
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3final; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 90 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -2 $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/rl -o rl -- python bench.py --no-cpu-baseline --no-workloads --no-latency > $O/rl_bench.json 2> $O/rl.err
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kp -o kp -- python bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline > $O/kp_bench.json 2> $O/kp.err
timeout 150 rocprofv3 --kernel-trace --stats -d $O/pp -o pp -- python bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline > $O/pp_bench.json 2> $O/pp.err
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads --no-latency > $O/pf.json 2> $O/pf.err
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads --no-latency > $O/pw.json 2> $O/pw.err
for op in kp pp; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $c -d $O/${op}_$c -o t -- python tools/roofline_ops.py $op 3 > $O/${op}_$c.log 2>&1
  done
done
ls $O | head -30
