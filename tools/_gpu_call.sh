# The round's final validation call (gpurun --timeout 900 -- 'bash tools/_gpu_call.sh'): GPU tests, smoke, the default bench line and
# the rocprofv3 passes behind profiles/r02_* (summarised afterwards with profiles/summarize_rocpd.py / summarize_pmc.py and
# tools/make_traffic.py).  Every command reads /dev/null and has its own timeout: a command that waits on stdin once cost 30
# GPU-minutes this round.
cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r2final2; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1 < /dev/null; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; tail -c 300 $O/bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/rl -o rl -- python bench.py --no-cpu-baseline --no-workloads > $O/rl_bench.json 2> $O/rl.err < /dev/null
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kp -o kp -- python bench.py --workload kpconv --steps 10 --warmup 3 --no-cpu-baseline > $O/kp_bench.json 2> $O/kp.err < /dev/null
timeout 150 rocprofv3 --kernel-trace --stats -d $O/pp -o pp -- python bench.py --workload pointpillars --steps 10 --warmup 3 --no-cpu-baseline > $O/pp_bench.json 2> $O/pp.err < /dev/null
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads > $O/pf.json 2> $O/pf.err < /dev/null
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads > $O/pw.json 2> $O/pw.err < /dev/null
ls $O
