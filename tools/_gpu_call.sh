cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out/r3an
for b in 16 32; do timeout 600 python tools/bench_deformable.py $b 20 2>&1 | tail -1 | tee -a gpurun_out/r3an/deformable.log; done
