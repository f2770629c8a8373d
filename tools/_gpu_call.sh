cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r3aq
timeout 900 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_pipelines.py -x -q 2>&1 | tail -2
for b in 16 32; do timeout 600 python tools/bench_deformable.py $b 20 2>&1 | tail -1 | tee -a gpurun_out/r3aq/deformable.log | cut -c1-260; done
