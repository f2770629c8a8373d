cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2w
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2w/pytest.log 2>&1; tail -3 gpurun_out/r2w/pytest.log
timeout 600 python bench.py > gpurun_out/r2w/bench.json 2> gpurun_out/r2w/bench.err; tail -c 3000 gpurun_out/r2w/bench.json
