set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
ML3D_TEST_TILE_ORDER=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2a/pytest.log
timeout 300 tools/ab_knn.sh - ML3D_TILE_ORDER=1 > gpurun_out/r2a/ab_knn.log 2>&1
timeout 300 tools/ab_run.sh base knn_g1 > gpurun_out/r2a/ab_group.log 2>&1
timeout 300 tools/ab_env.sh - ML3D_TILE_ORDER=1 > gpurun_out/r2a/ab_tile.log 2>&1
for e in "" "ML3D_TILE_ORDER=1"; do
  env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_${e:-base}.log 2>&1
done
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r2a/prof -name "*kernel_stats*" | head; find gpurun_out/r2a/prof -name "*.db" | head
for f in $(find gpurun_out/r2a/prof -name "*.db"); do python profiles/summarize_rocpd.py $f gpurun_out/r2a/kernel_stats.csv > /dev/null; done
find gpurun_out/r2a/prof -name "*.db" -size +20M -delete
cat gpurun_out/r2a/*.log | tail -40
