cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -6 | cut -c1-250
