cd $GRAFT_REPO_ROOT
exec < /dev/null
export TMPDIR=/tmp
timeout 1500 bash tools/gpu_e2e_train.sh 2>&1 | grep -v "it/s\]\|s/it\]" | tail -45 | cut -c1-260
