cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4a/pytest.log; tail -4 gpurun_out/r4a/pytest.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-workloads > $GRAFT_REPO_ROOT/gpurun_out/r4a/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4a/prof.err
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r4a/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
find gpurun_out/r4a/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r4a/prof -name "*.db" -delete
