cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 120 python tools/knn_only.py 5 2>&1 | grep knn_only > gpurun_out/r2k/abl.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> gpurun_out/r2k/abl.log
timeout 600 python bench.py --no-workloads --no-cpu-baseline > gpurun_out/r2k/bench.log 2>&1
cat gpurun_out/r2k/abl.log; tail -1 gpurun_out/r2k/bench.log | cut -c1-1500
