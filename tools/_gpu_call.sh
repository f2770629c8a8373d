cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k pointpillars 2>&1 | tail -40 | cut -c1-250
