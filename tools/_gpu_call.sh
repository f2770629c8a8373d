cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_pointpillars.py -x -q 2>&1 | tail -3
