cd $GRAFT_REPO_ROOT
exec < /dev/null
O=gpurun_out/r3ao; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 bash tools/gpu_ref_pipelines.sh > $O/pipelines.log 2>&1; tail -16 $O/pipelines.log | cut -c1-250
cp gpurun_out/pipelines/r03_pipeline_run.log $O/ 2>/dev/null
