cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_kpconv.py tests/test_gpu_knn.py -x -q 2>&1 | tail -2
